#!/bin/bash
# two-stream schedule of imgfd_detect_dev: where the Harris/FAST stream is released (IMGFD_CANNY_GATE 0 | 1 | 2), in which order
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for rep in 1 2; do for v in "IMGFD_CANNY_GATE=2" "IMGFD_CANNY_GATE=0" "IMGFD_CANNY_GATE=0 IMGFD_ORDER=hf" "IMGFD_CANNY_GATE=2 IMGFD_ORDER=hf"; do
  env $v python bench.py --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$v', d['value'], d['ms_per_step'])"
done; done
