#!/bin/bash
# bench.py's default step under each environment setting in VARIANTS (space-separated K=V[,K=V] groups)
cd $GRAFT_REPO_ROOT
for v in "X=1" $VARIANTS; do
  echo "--- $v"
  env ${v//,/ } timeout 300 python bench.py --no-cpu --no-extra --no-dist --steps 3 --warmup 1 --inner 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline'].get('avg_launch_us'))"
done
