#!/bin/bash
# bench.py's default step with each library under scripts/variants/ put in the product library's place (scratch copy on the GPU box)
cd $GRAFT_REPO_ROOT
cp image_amd/libimgfd.so /tmp/lib_default.so
for v in /tmp/lib_default.so scripts/variants/lib_*.so; do
  cp $v image_amd/libimgfd.so
  echo "--- $v"
  timeout 300 python bench.py --no-cpu --no-extra --no-dist --steps 3 --warmup 1 --inner 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline'].get('avg_launch_us'))"
done
cp /tmp/lib_default.so image_amd/libimgfd.so
for v in "" scripts/variants/lib_*.so; do
  echo "--- ${v:-default}"
  VARIANT_LIB=$v BATCHES=32 timeout 300 python scripts/k3_variants.py 2>&1 | grep us_per
  VARIANT_LIB=$v timeout 300 python scripts/harris_time.py 2>&1 | tail -1
done
