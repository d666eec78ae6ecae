#!/bin/bash
# round 6, call K: SURF tests on the device, the pyramid kernels alone (two tiles, one lane) and the batch, product library and the
# variants of the first octave's kernel under scripts/variants/ (run length, registers)
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$R/gpurun_out/r6k; mkdir -p $O
timeout 900 python -m pytest tests/test_surf.py -q -m gpu -x --timeout 300 2>&1 | tail -2 | tee $O/pytest_surf.txt
V="default $(ls scripts/variants/ | sed -n 's/^lib_\(.*\)\.so$/\1/p' | tr '\n' ' ')"
TILES=2 IMGFD_SURF_LANES=1 VARIANTS="$V" bash scripts/rounds/gpu_r6_lds_phases.sh 2>&1 | grep -v "^{"
for v in $V; do lib=""; [ $v != default ] && lib=$R/scripts/variants/lib_$v.so
for t in 1 64; do echo -n "$v tiles=$t " | tee -a $O/batch.txt; VARIANT_LIB=$lib TILES=$t timeout 200 python scripts/surf_dev_time.py 2>&1 | grep "^{" | cut -c1-60 | tee -a $O/batch.txt; done; done
