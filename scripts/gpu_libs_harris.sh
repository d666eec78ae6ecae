#!/bin/bash
# imgfd_harris_dev on 32 4K frames with each library under scripts/variants/ (and the product library first)
cd $GRAFT_REPO_ROOT
for v in "" scripts/variants/lib_*.so; do
  VARIANT_LIB=$v timeout 300 python scripts/harris_time.py 2>&1 | tail -1
done
