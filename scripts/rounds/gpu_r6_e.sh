#!/bin/bash
# round 6, call E: two-kernel maximum test (surf_nms_screen + surf_nms_finish): tests, group size x lanes, one tile, timeline of a group
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$R/gpurun_out/r6e; mkdir -p $O
timeout 900 python -m pytest tests/test_surf.py tests/test_full_size.py tests/test_knn.py -q -m gpu -x > $O/pytest_surf.txt 2>&1; grep -E "passed|failed|error" $O/pytest_surf.txt | tail -3
for cfg in "1 1" "8 1" "8 2" "16 2" "4 2"; do set -- $cfg
  echo -n "group $1 lanes $2 " | tee -a $O/surf.txt
  IMGFD_SURF_GROUP=$1 IMGFD_SURF_LANES=$2 timeout 200 python scripts/surf_dev_time.py 2>&1 | grep "^{" | tee -a $O/surf.txt
done
echo -n "single tile " | tee -a $O/surf.txt
TILES1=1 timeout 200 python scripts/surf_dev_time.py 2>&1 | grep "^{" | tee -a $O/surf.txt
TAG=g8 LAST=48 bash scripts/rounds/gpu_r6_tl.sh > /dev/null 2>&1; cp gpurun_out/r6tl/timeline_g8.txt $O/
TILES1=1 TAG=one LAST=14 bash scripts/rounds/gpu_r6_tl.sh > /dev/null 2>&1; cp gpurun_out/r6tl/timeline_one.txt $O/
