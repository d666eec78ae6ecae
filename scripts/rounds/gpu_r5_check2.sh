#!/bin/bash
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$R/gpurun_out/r5_check2; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3 | tee $O/pytest_gpu.txt
TILES=16,1 timeout 300 python scripts/surf_dev_time.py 2>&1 | grep "^{" | tee $O/surf.txt
python bench.py --config 4 --no-cpu --steps 2 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config4', d['value'], d['ms_per_step'], json.dumps(d['roofline'])[:400])" | tee $O/config4.txt
python bench.py --batch 1 --no-cpu --no-extra --no-dist --steps 10 --warmup 3 --inner 50 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('b1', d['value'], d['ms_per_step']/50)" | tee $O/b1.txt
