#!/bin/bash
# quick K3 iteration: stage parity tests + standalone K3 timing + in-pipeline bench.  Usage: gpu_k3.sh [pytest -k expr]
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
T="${1:-harris}"
[ "$T" != "none" ] && timeout 600 python -m pytest tests -m gpu -x -q -k "$T" 2>&1 | tail -4
for v in ${VARIANTS:-0}; do  # (variants were experiment builds; the env switch is gone)
  echo "--- IMGFD_K3_VARIANT=$v"
  IMGFD_K3_VARIANT=$v timeout 300 python scripts/k3_time.py 2>&1 | grep kernel
  IMGFD_K3_VARIANT=$v timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench', d['value'], 'Mpix/s; K3 in-pipeline', d['roofline'])"
done
exit 0
