#!/bin/bash
# time every experiment build of the library under scripts/variants/ (batch 32 and 1)
cd "$GRAFT_REPO_ROOT"
for f in scripts/variants/*.so; do echo "--- $f"; VARIANT_LIB=$f BATCHES=${BATCHES:-1,32} timeout 300 python scripts/k3_variants.py 2>&1 | grep kernel; done
