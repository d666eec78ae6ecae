#!/bin/bash
# round 6: the shortest-segment rule of the marching Gaussian + gradient kernel: Harris tests on the device, headline / single frame / config 5
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$R/gpurun_out/r6seg; mkdir -p $O
timeout 900 python -m pytest tests/test_harris_stages.py tests/test_harris_api.py tests/test_full_size.py tests/test_device_py.py -q -m gpu -x --timeout 300 2>&1 | tail -1
for a in "" "--batch 1 --inner 50 --steps 5" "--batch 8 --steps 8" "--config 5 --steps 2"; do
  for s in 0 1080; do echo -n "args='$a' IMGFD_GAUSS_MARCH_SEG=$s " | tee -a $O/check.txt
  IMGFD_GAUSS_MARCH_SEG=$s timeout 400 python bench.py --no-extra --no-cpu $a 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])" | tee -a $O/check.txt; done
done
