#!/bin/bash
# round 6, call H: whole GPU test-suite (ticketed canny_finish, CU-masked streams, grouped SURF), smoke, function traffic (PMC), short bench
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$R/gpurun_out/r6h; mkdir -p $O
timeout 1200 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.txt 2>&1; grep -E "passed|failed|error" $O/pytest_gpu.txt | tail -3
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep "smoke" ) | tee $O/smoke.txt
bash scripts/gpu_pmc_functions.sh 2>&1 | tail -4 | tee $O/function_traffic.txt
timeout 900 python bench.py --steps 6 2>/dev/null | tail -1 > $O/bench.json
python - <<PY | tee $O/summary.txt
import json
d = json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print(json.dumps(d["summary"], indent=None))
PY
