#!/bin/bash
# Structure tensor A/B in ONE GPU call: the product library and every scripts/variants/lib_*.so through the 20 B/px doorway
# (batch 1 and 32, HIP events) and through imgfd_harris_dev (32 frames).  Optional: TESTS="pytest args".
set -u
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/k3ab"; mkdir -p "$O"
export TMPDIR=/tmp
{
if [ -n "${TESTS:-}" ]; then echo "=== pytest $TESTS"; timeout 900 python -m pytest $TESTS -m gpu -x -q 2>&1 | tail -3; fi
echo "=== doorway (20 B/px), HIP events"
for v in "" scripts/variants/lib_*.so; do
  VARIANT_LIB=$v BATCHES=1,32 timeout 200 python scripts/k3_variants.py 2>&1 | grep structure_tensor | cut -c1-260
done
echo "=== imgfd_harris_dev, 32 frames"
for v in "" scripts/variants/lib_*.so; do
  VARIANT_LIB=$v timeout 200 python scripts/harris_time.py 2>&1 | tail -1
done
} > "$O/log.txt" 2>&1
cat "$O/log.txt"
exit 0
