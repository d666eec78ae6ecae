#!/bin/bash
# round 6, call I: adaptive sweep count of small Canny batches + ticketed canny_finish: Canny tests, the bench launch tests, single frame
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$R/gpurun_out/r6i; mkdir -p $O
timeout 200 python -m pytest tests/test_canny.py -q -m gpu -x --timeout 60 -k "cu_masked or small_batches" > $O/pytest_new.txt 2>&1; grep -E "passed|failed|error|Timeout" $O/pytest_new.txt | tail -3
for i in 1 2; do timeout 300 python bench.py --batch 1 --inner 50 --steps 20 --no-cpu --no-extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('batch1', d['value'], d['ms_per_step']/50, d['config'].get('shader_clock',{}).get('mean_GHz'))" | tee -a $O/b1.txt; done
timeout 300 bash scripts/gpu_b1_timeline.sh > $O/single_frame_timeline.txt 2>/dev/null
