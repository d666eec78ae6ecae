#!/bin/bash
# round 5, integral image rework: SURF tests on the device, per-kernel durations with the residue layout written by the integral
# image's last kernel (1) and by its own kernel (0), whole-call timings, config 4
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$R/gpurun_out/r5_int; mkdir -p $O
timeout 300 python -m pytest tests/test_surf.py tests/test_full_size.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3 | tee $O/pytest_gpu.txt
bash scripts/gpu_surf_kstats.sh IMGFD_SURF_RESIDUE_FUSED 1 0 2>&1 | tee $O/kstats.txt
for f in 1 0; do IMGFD_SURF_RESIDUE_FUSED=$f timeout 120 python scripts/surf_dev_time.py 2>&1 | grep "^{" | tee -a $O/surf.txt; done
TILES1=1 timeout 120 python scripts/surf_dev_time.py 2>&1 | grep "^{" | tee -a $O/surf.txt
timeout 200 python bench.py --config 4 --no-cpu --steps 2 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config4', d['value'], d['ms_per_step'], d['roofline']['surf']['frac'])" | tee $O/config4.txt
