#!/bin/bash
# lab: where FAST-9 and the Harris chain are released relative to Canny's kernels (switches canny_gate / harris_gate), small batches
cd $GRAFT_REPO_ROOT
for b in 1 2 4; do
for env in "" "IMGFD_HARRIS_GATE=0" "IMGFD_HARRIS_GATE=2" "IMGFD_CANNY_GATE=1" "IMGFD_CANNY_GATE=1 IMGFD_HARRIS_GATE=2" "IMGFD_CANNY_GATE=2" ""; do
  echo -n "batch $b  [$env]  "
  env $env timeout 100 python bench.py --batch $b --no-cpu --no-extra --no-dist --steps 10 --warmup 3 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done; done
