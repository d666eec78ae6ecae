#!/bin/bash
# round 6: rows per segment of canny_blur_march (scripts/variants/lib_blurseg.so reads IMGFD_BLUR_SEG; unset = the launch model) at the
# headline configuration and at config 3 (1024 x 1080p), sustained: Mpixel/s, ms per step
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$R/gpurun_out/r6bs; mkdir -p $O
for s in "" 40 72 104 136 200 264 360 520 1080 2160 ""; do
  echo -n "IMGFD_BLUR_SEG=$s " | tee -a $O/seg.txt
  ( [ -n "$s" ] && export IMGFD_BLUR_SEG=$s; VARIANT_LIB=$R/scripts/variants/lib_blurseg.so timeout 300 python scripts/bench_variant.py --no-extra --no-cpu --steps 12 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])" ) | tee -a $O/seg.txt
done
for s in "" 40 72 136 264 520 1080 ""; do
  echo -n "config 3 IMGFD_BLUR_SEG=$s " | tee -a $O/seg.txt
  ( [ -n "$s" ] && export IMGFD_BLUR_SEG=$s; VARIANT_LIB=$R/scripts/variants/lib_blurseg.so timeout 300 python scripts/bench_variant.py --config 3 --no-cpu --steps 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])" ) | tee -a $O/seg.txt
done
