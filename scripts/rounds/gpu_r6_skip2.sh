#!/bin/bash
# round 6, final schedule: what each SURF kernel costs the sustained batch -- bench.py --config 4 with one kernel left out at a time
# (scripts/variants/lib_skip.so: surf.hip with an IMGFD_SURF_SKIP bit mask: 1 sums, 2 carry, 4 apply, 8 first octave, 16 gather kernel,
# 32 list, 64 screen, 128 finish, 256 ranking, 512 orientation, 1024 descriptor; results are meaningless, the time is the point)
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$R/gpurun_out/r6skip2; mkdir -p $O
for m in ${MASKS:-0 1 2 4 8 16 32 64 128 256 512 1024 1536 2016 24 7 0}; do
  echo -n "skip=$m " | tee -a $O/skip.txt
  IMGFD_SURF_SKIP=$m VARIANT_LIB=$R/scripts/variants/lib_skip.so timeout 200 python scripts/bench_variant.py --config 4 --steps 5 --warmup 2 --no-cpu --max-parity-frames 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); r=d['roofline']; print(d['value'], r['surf']['ms_per_tile'], r['fhog_ms_per_tile'])" | tee -a $O/skip.txt
done
