#!/bin/bash
# single-frame case (bench.py --batch 1) under each environment setting in VARIANTS
cd $GRAFT_REPO_ROOT
for v in "X=1" $VARIANTS; do
  echo "--- $v"
  env ${v//,/ } timeout 300 python bench.py --no-cpu --no-extra --no-dist --batch ${B:-1} --inner 50 --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step']/50)"
done
