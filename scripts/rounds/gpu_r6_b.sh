#!/bin/bash
# round 6, call B: the two-stage lanes of imgfd_surf_dev -- SURF tests on the device, imgfd_surf_dev per tile for 1..4 lanes
# (16 tiles), a single tile, the round-5 library beside it, a short config-4 line
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$R/gpurun_out/r6b; mkdir -p $O
timeout 900 python -m pytest tests/test_surf.py tests/test_full_size.py tests/test_knn.py -q -m gpu -x > $O/pytest_surf.txt 2>&1; grep -E "passed|failed|error" $O/pytest_surf.txt | tail -3
echo "== lib_r05" | tee -a $O/surf.txt
VARIANT_LIB=$R/scripts/variants/lib_r05.so timeout 200 python scripts/surf_dev_time.py 2>&1 | grep "^{" | tee -a $O/surf.txt
for lanes in 1 2 3 4; do
  echo "== this tree, lanes $lanes" | tee -a $O/surf.txt
  IMGFD_SURF_LANES=$lanes timeout 200 python scripts/surf_dev_time.py 2>&1 | grep "^{" | tee -a $O/surf.txt
done
TILES1=1 timeout 200 python scripts/surf_dev_time.py 2>&1 | grep "^{" | tee -a $O/surf.txt
timeout 600 python bench.py --config 4 --batch 64 --steps 2 --warmup 1 --no-cpu 2>/dev/null | tail -1 > $O/bench_c4.json
python - <<PY | tee $O/c4_summary.txt
import json
d = json.loads(open("$O/bench_c4.json").read().strip().splitlines()[-1])
print("config4 batch64", d["value"], d["ms_per_step"], d["roofline"]["surf"]["ms_per_tile"], d["roofline"]["fhog_ms_per_tile"], d.get("parity"), d["config"].get("shader_clock"))
print(json.dumps(d["summary"]))
PY
