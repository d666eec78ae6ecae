#!/bin/bash
# round 6: the remaining switches that touch the headline configuration (32 x 4K, Harris + FAST-9 + Canny), sustained: Mpixel/s, ms per step
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$R/gpurun_out/r6hs; mkdir -p $O
for v in "X=1" IMGFD_HYST_SWEEPS=6 IMGFD_HYST_SWEEPS=7 IMGFD_HYST_SWEEPS=8 IMGFD_HYST_SWEEPS=10 IMGFD_TENSOR_WORKERS=248 IMGFD_TENSOR_WORKERS=512 IMGFD_MAX_CHUNK_FRAMES=16 IMGFD_MAX_CHUNK_FRAMES=8 IMGFD_GAUSS_MARCH_SEG=270 IMGFD_GAUSS_MARCH_SEG=540 IMGFD_FIR_MODE=0 "X=1"; do
  echo -n "$v " | tee -a $O/sweep.txt
  env $v timeout 300 python bench.py --no-extra --no-cpu --steps 12 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline'].get('in_pipeline',{}).get('avg_launch_us'))" | tee -a $O/sweep.txt
done
