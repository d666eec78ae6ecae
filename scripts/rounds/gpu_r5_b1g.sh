#!/bin/bash
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$R/gpurun_out/r5_b1g; mkdir -p $O
IMGFD_DETECT_SWAP=1 timeout 600 python -m pytest tests/test_device_py.py tests/test_frame_stream.py tests/test_sub_batches.py tests/test_bench_launch.py -x -q -m gpu 2>&1 | grep -E "passed|failed" | tee $O/pytest.txt
run() {  local label="$1"; shift
  echo -n "$label  " | tee -a $O/variants.txt
  env "$@" timeout 200 python bench.py --batch ${BATCH:-1} --no-cpu --no-extra --no-dist --steps 10 --warmup 3 --inner ${INNER:-50} 2>/dev/null < /dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('Gpx/s', round(d['value']/1e3,2), ' ms/step', d['ms_per_step'], d['config']['feature_counts'], 'k3', d['roofline'].get('in_pipeline',{}).get('avg_us'))" | tee -a $O/variants.txt
}
: > $O/variants.txt
run "default                 "
run "swap                    " IMGFD_DETECT_SWAP=1
run "swap harris_gate 2      " IMGFD_DETECT_SWAP=1 IMGFD_HARRIS_GATE=2
run "swap canny_gate 2       " IMGFD_DETECT_SWAP=1 IMGFD_CANNY_GATE=2
run "swap graph              " IMGFD_DETECT_SWAP=1 IMGFD_DETECT_GRAPH=8
BATCH=2 run "b2 default              "
BATCH=2 run "b2 swap                 " IMGFD_DETECT_SWAP=1
BATCH=4 run "b4 default              "
BATCH=4 run "b4 swap                 " IMGFD_DETECT_SWAP=1
BATCH=8 INNER=20 run "b8 default              "
BATCH=8 INNER=20 run "b8 swap                 " IMGFD_DETECT_SWAP=1
BATCH=32 INNER=10 run "b32 default             "
BATCH=32 INNER=10 run "b32 swap                " IMGFD_DETECT_SWAP=1
IMGFD_DETECT_SWAP=1 python scripts/b1_host_probe.py 2>&1 | grep -v amdgpu.ids | head -1 | tee $O/host_probe.txt
IMGFD_DETECT_SWAP=1 bash scripts/rounds/gpu_r5_tl.sh > /dev/null 2>&1; cp $R/gpurun_out/r5_tl/timeline.txt $O/timeline_swap.txt; cat $O/timeline_swap.txt
