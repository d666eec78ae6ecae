#!/bin/bash
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$R/gpurun_out/r5_check3; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3 | tee $O/pytest_gpu.txt
python scripts/surf_dev_time.py 2>&1 | grep "^{" | tee $O/surf.txt
TILES1=1 python scripts/surf_dev_time.py 2>&1 | grep "^{" | tee -a $O/surf.txt
python bench.py --config 4 --no-cpu --steps 2 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config4', d['value'], d['ms_per_step'], d['roofline']['surf']['frac'])" | tee $O/config4.txt
python bench.py --no-cpu --no-extra --steps 10 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('traffic'))" | tee $O/default.txt
python bench.py --config 3 --no-cpu --steps 3 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config3', d['value'], d['ms_per_step'])" | tee $O/config3.txt
BATCH=32 ITERS=10 python scripts/canny_time.py 2>/dev/null | grep "^{" | tee $O/canny32.txt
