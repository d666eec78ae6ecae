#!/bin/bash
# imgfd_surf_dev per 4096^2 tile (one lane, default lanes, single tile) with each library under scripts/variants/ (product library first)
cd $GRAFT_REPO_ROOT
for v in "" scripts/variants/lib_*.so; do
  echo "--- ${v:-default}"
  VARIANT_LIB=$v IMGFD_SURF_LANES=1 timeout 300 python scripts/surf_dev_time.py 2>&1 | tail -1
  VARIANT_LIB=$v timeout 300 python scripts/surf_dev_time.py 2>&1 | tail -1
done
