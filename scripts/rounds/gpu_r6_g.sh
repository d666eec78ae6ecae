#!/bin/bash
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$R/gpurun_out/r6g; mkdir -p $O
timeout 900 python -m pytest tests/test_surf.py tests/test_full_size.py -q -m gpu -x > $O/pytest_surf.txt 2>&1; grep -E "passed|failed|error" $O/pytest_surf.txt | tail -3
for cfg in "16 2" "8 2"; do set -- $cfg
  echo -n "64 tiles group $1 lanes $2 " | tee -a $O/surf.txt
  TILES=64 IMGFD_SURF_GROUP=$1 IMGFD_SURF_LANES=$2 timeout 200 python scripts/surf_dev_time.py 2>&1 | grep "^{" | tee -a $O/surf.txt
done
echo -n "single tile " | tee -a $O/surf.txt
TILES1=1 timeout 200 python scripts/surf_dev_time.py 2>&1 | grep "^{" | tee -a $O/surf.txt
TILES1=1 TAG=one LAST=14 bash scripts/rounds/gpu_r6_tl.sh > /dev/null 2>&1; cp gpurun_out/r6tl/timeline_one.txt $O/
TILES=32 TAG=g16 LAST=80 bash scripts/rounds/gpu_r6_tl.sh > /dev/null 2>&1; cp gpurun_out/r6tl/timeline_g16.txt $O/
