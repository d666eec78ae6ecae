#!/bin/bash
# round 6, call D: imgfd_surf_dev in groups of tiles (back stages as one launch per group): tests, group size x front lanes, one tile
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$R/gpurun_out/r6d; mkdir -p $O
timeout 900 python -m pytest tests/test_surf.py tests/test_full_size.py tests/test_knn.py -q -m gpu -x > $O/pytest_surf.txt 2>&1; grep -E "passed|failed|error" $O/pytest_surf.txt | tail -3
for grp in 1 2 4 8 16; do for lanes in 1 2 3; do
  echo -n "group $grp lanes $lanes " | tee -a $O/surf.txt
  IMGFD_SURF_GROUP=$grp IMGFD_SURF_LANES=$lanes timeout 200 python scripts/surf_dev_time.py 2>&1 | grep "^{" | tee -a $O/surf.txt
done; done
echo -n "single tile " | tee -a $O/surf.txt
TILES1=1 timeout 200 python scripts/surf_dev_time.py 2>&1 | grep "^{" | tee -a $O/surf.txt
