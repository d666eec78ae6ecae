#!/bin/bash
# round 6, call F: steady state of the grouped imgfd_surf_dev (64 tiles): group size x lanes
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$R/gpurun_out/r6f; mkdir -p $O
for cfg in "8 2" "8 1" "4 2" "16 2" "2 2" "8 3"; do set -- $cfg
  echo -n "64 tiles group $1 lanes $2 " | tee -a $O/surf.txt
  TILES=64 IMGFD_SURF_GROUP=$1 IMGFD_SURF_LANES=$2 timeout 200 python scripts/surf_dev_time.py 2>&1 | grep "^{" | tee -a $O/surf.txt
done
