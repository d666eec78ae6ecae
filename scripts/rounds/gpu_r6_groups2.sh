#!/bin/bash
# round 6, after the first-octave change: groups x lanes of imgfd_surf_dev again, sustained (bench.py --config 4, 128 tiles per step)
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$R/gpurun_out/r6groups2; mkdir -p $O
for g in 4 8 16; do for l in 1 2 3; do
  echo -n "group=$g lanes=$l " | tee -a $O/sweep.txt
  IMGFD_SURF_GROUP=$g IMGFD_SURF_LANES=$l timeout 300 python bench.py --config 4 --batch 128 --steps 6 --warmup 2 --no-cpu 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); r=d['roofline']; print(d['value'], r['surf']['ms_per_tile'], r['fhog_ms_per_tile'])" | tee -a $O/sweep.txt
done; done
