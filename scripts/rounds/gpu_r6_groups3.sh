#!/bin/bash
# round 6: the candidates of gpu_r6_groups2.sh again, the bench's own batch (256 tiles per step), twice each in alternating order
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$R/gpurun_out/r6groups3; mkdir -p $O
for rep in 1 2; do for gl in "8 2" "4 3" "8 3" "16 2" "16 3" "4 4" "8 4" "12 3"; do set -- $gl
  echo -n "group=$1 lanes=$2 " | tee -a $O/sweep.txt
  IMGFD_SURF_GROUP=$1 IMGFD_SURF_LANES=$2 timeout 300 python bench.py --config 4 --steps 6 --warmup 2 --no-cpu 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); r=d['roofline']; print(d['value'], r['surf']['ms_per_tile'], r['fhog_ms_per_tile'])" | tee -a $O/sweep.txt
done; done
