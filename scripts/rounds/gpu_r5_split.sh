#!/bin/bash
# round 5, "surf_split": one tile with octaves 1-3 on the companion's stream beside octave 0 (1) or on one stream (0); the batch for reference
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$R/gpurun_out/r5_split; mkdir -p $O
timeout 300 python -m pytest tests/test_surf.py -q -m gpu 2>&1 | grep -E "passed|failed|error|FAILED" | tail -4 | tee $O/pytest_gpu.txt
for i in 1 2 3; do for v in 1 0; do echo "split $v: $(TILES1=1 IMGFD_SURF_SPLIT=$v timeout 120 python scripts/surf_dev_time.py 2>&1 | grep '^{')"; done; done | tee $O/single_tile.txt
echo "batch: $(timeout 120 python scripts/surf_dev_time.py 2>&1 | grep '^{')" | tee -a $O/single_tile.txt
