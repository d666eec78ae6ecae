#!/bin/bash
# round 5, single frame, third pass: taps cache, scan folded into scatter; queue order / gates / sweeps again
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$R/gpurun_out/r5_b1c; mkdir -p $O
timeout 900 python -m pytest tests/test_canny.py tests/test_fast9.py tests/test_harris_api.py tests/test_device_py.py -x -q -m gpu 2>&1 | tail -3 | tee $O/pytest.txt
python scripts/b1_host_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/host_probe.txt
run() {  # label, env...
  local label="$1"; shift
  echo -n "$label  " | tee -a $O/variants.txt
  env "$@" timeout 200 python bench.py --batch ${BATCH:-1} --no-cpu --no-extra --no-dist --steps 10 --warmup 3 --inner 50 2>/dev/null < /dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('Gpx/s', round(d['value']/1e3,2), ' us/pass', round(d['ms_per_step']/50*1000,1))" | tee -a $O/variants.txt
}
: > $O/variants.txt
run "defaults (22 s6, defer, gate 0/1)    "
run "defer 0                              " IMGFD_DETECT_DEFER=0
run "defer 0 harris_gate 0                " IMGFD_DETECT_DEFER=0 IMGFD_HARRIS_GATE=0
run "defer 0 harris_gate 2                " IMGFD_DETECT_DEFER=0 IMGFD_HARRIS_GATE=2
run "harris_gate 0                        " IMGFD_HARRIS_GATE=0
run "harris_gate 2                        " IMGFD_HARRIS_GATE=2
run "s5                                   " IMGFD_HYST_SWEEPS=5
run "s7                                   " IMGFD_HYST_SWEEPS=7
run "graph                                " IMGFD_DETECT_GRAPH=8
run "graph defer 0 harris_gate 0          " IMGFD_DETECT_GRAPH=8 IMGFD_DETECT_DEFER=0 IMGFD_HARRIS_GATE=0
run "42 s5                                " IMGFD_HYST_BLOCK=42 IMGFD_HYST_SWEEPS=5
run "44 s4                                " IMGFD_HYST_BLOCK=44 IMGFD_HYST_SWEEPS=4
BATCH=2 run "b2                                   "
BATCH=4 run "b4                                   "
BATCH=32 run "b32                                  "
BATCH=32 run "b32 hyst 11                          " IMGFD_HYST_BLOCK=11
bash scripts/rounds/gpu_r5_tl.sh > /dev/null 2>&1; cp $R/gpurun_out/r5_tl/timeline.txt $O/timeline_default.txt; cat $O/timeline_default.txt
