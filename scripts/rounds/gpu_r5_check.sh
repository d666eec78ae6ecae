#!/bin/bash
# full GPU test-suite + single frame / small batches / default bench (short), one call
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$R/gpurun_out/r5_check; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3 | tee $O/pytest_gpu.txt
run() {  local label="$1"; shift
  echo -n "$label  " | tee -a $O/variants.txt
  env "$@" timeout 200 python bench.py --batch ${BATCH:-1} --no-cpu --no-extra --no-dist --steps 10 --warmup 3 --inner ${INNER:-50} 2>/dev/null < /dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('Gpx/s', round(d['value']/1e3,2), ' ms/step', d['ms_per_step'])" | tee -a $O/variants.txt
}
: > $O/variants.txt
run "b1 " ; run "b1 harris_gate 2" IMGFD_HARRIS_GATE=2; run "b1 defer 0" IMGFD_DETECT_DEFER=0
BATCH=2 run "b2"; BATCH=4 run "b4"; BATCH=8 INNER=20 run "b8"; BATCH=8 INNER=20 run "b8 defer1" IMGFD_DETECT_DEFER=1; BATCH=32 INNER=10 run "b32"
python scripts/b1_host_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/host_probe.txt
python scripts/canny_serpentine_time.py 2>&1 | tail -2 | tee $O/serpentine.txt
python bench.py --no-cpu --no-extra --steps 10 2>/dev/null | tail -1 > $O/bench_default.json; python -c "
import json; d=json.load(open('$O/bench_default.json')); print('default', d['value'], d['ms_per_step'], d['roofline']['frac'], d['config'].get('collectives','')[:80])"
python bench.py --config 3 --no-cpu --steps 3 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config3', d['value'], d['ms_per_step'])"
