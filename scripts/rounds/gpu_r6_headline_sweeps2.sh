#!/bin/bash
# round 6: rows per segment of the marching Gaussian + gradient kernel at the headline configuration (sustained): Mpixel/s, ms per step
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$R/gpurun_out/r6hs; mkdir -p $O
for s in 0 90 135 180 216 270 360 432 540 720 1080 2160 0 270; do
  echo -n "IMGFD_GAUSS_MARCH_SEG=$s " | tee -a $O/seg.txt
  IMGFD_GAUSS_MARCH_SEG=$s timeout 300 python bench.py --no-extra --no-cpu --steps 12 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])" | tee -a $O/seg.txt
done
