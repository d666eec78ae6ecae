#!/bin/bash
# fHOG per 4096^2 tile with each library under scripts/variants/ (and the product library first)
cd $GRAFT_REPO_ROOT
for v in "" scripts/variants/lib_*.so; do
  echo "--- ${v:-default}"
  VARIANT_LIB=$v TILES=16,1 QUICK=1 timeout 300 python scripts/fhog_variants.py 2>&1 | grep '"fused": 1, "bands": 0, "threads": 256'
done
