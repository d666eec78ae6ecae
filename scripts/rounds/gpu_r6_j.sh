#!/bin/bash
# round 6, call J: the whole GPU test-suite after the switches were pruned + smoke
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$R/gpurun_out/r6j; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu --timeout 400 > $O/pytest_gpu.txt 2>&1; grep -E "passed|failed|error" $O/pytest_gpu.txt | tail -3
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep "smoke" ) | tee $O/smoke.txt
