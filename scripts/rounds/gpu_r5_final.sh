#!/bin/bash
# round 5, final code: the whole GPU test-suite, smoke(), the host-API table of the bench (the single-image SURF call with "surf_split"), one tile of SURF
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$R/gpurun_out/r5_final; mkdir -p $O
timeout 900 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|error|FAILED" | tail -4 | tee $O/pytest_gpu.txt
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 ) | tee $O/smoke.txt
timeout 600 python bench.py --steps 6 2>/dev/null | tail -1 > $O/bench.json
python - <<PY | tee $O/summary.txt
import json
d = json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("default", d["value"], d["ms_per_step"], d["roofline"]["frac"], "cpu", d["cpu_baseline"]["value"])
for k in ("2_batch1", "3", "4", "5", "5_h2d"):
    print(k, d["configs"][k].get("value"), d["configs"][k].get("ms_per_step"))
for k, v in d["configs"]["host_api"]["calls"].items():
    print(k, v["ms_best"], v["floor_over_best"])
PY
TILES1=1 timeout 120 python scripts/surf_dev_time.py 2>&1 | grep "^{" | tee $O/surf_single_tile.txt
