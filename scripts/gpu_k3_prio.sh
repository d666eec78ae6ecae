#!/bin/bash
# Round 4, structure tensor: progress equalisation by wave priority (FT_PRIO) in the workgroup-marching kernel, A/B in ONE call.
set -u
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/k3p"; mkdir -p "$O"
export TMPDIR=/tmp
{
echo "=== pytest (harris stages + api)"
timeout 600 python -m pytest tests/test_harris_stages.py tests/test_harris_api.py -m gpu -x -q 2>&1 | tail -3
echo "=== doorway (20 B/px), HIP events: product library (FT_PRIO 1), then variants"
for v in "" scripts/variants/lib_p*.so; do
  VARIANT_LIB=$v BATCHES=1,32 timeout 200 python scripts/k3_variants.py 2>&1 | grep structure_tensor
done
echo "=== imgfd_harris_dev, 32 frames (response variant in the pipeline)"
for v in "" scripts/variants/lib_p*.so; do
  VARIANT_LIB=$v timeout 200 python scripts/harris_time.py 2>&1 | tail -1
done
echo "=== FT_PROFILE phase split with FT_PRIO 1"
VARIANT_LIB=scripts/variants/lib_ftprof1.so timeout 200 python scripts/k3_phase.py 2>&1 | tail -9
echo "=== bench (product library)"
timeout 600 python bench.py --no-cpu 2>&1 | tail -1 > "$O/bench.json"; cut -c1-1200 "$O/bench.json"
for v in scripts/variants/lib_p0.so; do
  echo "=== bench with $v in the product library's place"
  cp image_amd/libimgfd.so /tmp/lib_keep.so; cp $v image_amd/libimgfd.so
  timeout 600 python bench.py --no-cpu 2>&1 | tail -1 > "$O/bench_p0.json"; cut -c1-1200 "$O/bench_p0.json"
  cp /tmp/lib_keep.so image_amd/libimgfd.so
done
} > "$O/log.txt" 2>&1
tail -40 "$O/log.txt"
exit 0
