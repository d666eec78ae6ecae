#!/bin/bash
# round 6: what each SURF kernel costs the BATCH (4 lanes, 16 tiles): imgfd_surf_dev with kernels left out
# (scripts/variants/lib_skip.so = surf.hip with an IMGFD_SURF_SKIP bit mask: 1 sums, 2 carry, 4 apply, 8 first octave, 16 octaves 1-3,
# 32 maximum test, 64 ranking, 128 orientation, 256 descriptor; results are meaningless, the time is the point)
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$R/gpurun_out/r6skip; mkdir -p $O
for m in ${MASKS:-0 8 16 32 64 128 256 384 448 480 24 31 0}; do
  for lanes in ${LANES:-4}; do
  echo -n "skip=$m lanes=$lanes " | tee -a $O/skip.txt
  IMGFD_SURF_LANES=$lanes IMGFD_SURF_SKIP=$m VARIANT_LIB=$R/scripts/variants/lib_skip.so timeout 200 python scripts/surf_dev_time.py 2>&1 | grep "^{" | tee -a $O/skip.txt
  done
done
