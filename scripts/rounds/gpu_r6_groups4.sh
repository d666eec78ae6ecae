#!/bin/bash
# round 6: around groups of 4 tiles on three front streams (the bench's config 4, 256 tiles per step)
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$R/gpurun_out/r6groups4; mkdir -p $O
for gl in "4 3" "3 3" "5 3" "6 3" "7 3" "2 2" "3 2" "6 2" "3 1" "4 3" "6 4" "5 4"; do set -- $gl
  echo -n "group=$1 lanes=$2 " | tee -a $O/sweep.txt
  IMGFD_SURF_GROUP=$1 IMGFD_SURF_LANES=$2 timeout 300 python bench.py --config 4 --steps 6 --warmup 2 --no-cpu 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); r=d['roofline']; print(d['value'], r['surf']['ms_per_tile'], r['fhog_ms_per_tile'])" | tee -a $O/sweep.txt
done
