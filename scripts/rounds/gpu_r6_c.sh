#!/bin/bash
# round 6, call C: priority of the back streams of imgfd_surf_dev (IMGFD_SURF_BACK_PRIO 1 = highest, 0 = default, 2 = lowest) x lanes
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$R/gpurun_out/r6c; mkdir -p $O
for prio in 1 0 2; do for lanes in 1 2 3 4; do
  echo -n "prio $prio lanes $lanes " | tee -a $O/surf.txt
  IMGFD_SURF_BACK_PRIO=$prio IMGFD_SURF_LANES=$lanes timeout 200 python scripts/surf_dev_time.py 2>&1 | grep "^{" | tee -a $O/surf.txt
done; done
for prio in 1 0; do
echo -n "single tile prio $prio " | tee -a $O/surf.txt
IMGFD_SURF_BACK_PRIO=$prio TILES1=1 timeout 200 python scripts/surf_dev_time.py 2>&1 | grep "^{" | tee -a $O/surf.txt
done
