#!/bin/bash
# round 6: bands per workgroup of the fused fHOG kernel, sustained (bench.py --config 4): Mpixel/s, SURF and fHOG ms per tile
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$R/gpurun_out/r6fb; mkdir -p $O
for b in 0 1 2 3 4 6 8 16 0; do
  echo -n "IMGFD_FHOG_BANDS=$b " | tee -a $O/bands.txt
  IMGFD_FHOG_BANDS=$b timeout 300 python bench.py --config 4 --steps 6 --warmup 2 --no-cpu 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); r=d['roofline']; print(d['value'], r['surf']['ms_per_tile'], r['fhog_ms_per_tile'])" | tee -a $O/bands.txt
done
