#!/bin/bash
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for thr in 1e30 1e5 3000 130 10; do
cd /tmp; rm -rf /tmp/pn2
PYTHONPATH=$R timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pn2 -o p -- python $R/scripts/harris_thr.py $thr 2>&1 | grep threshold
f=$(find /tmp/pn2 -name "*kernel_stats.csv" | head -1)
python - $f <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'nms_sparse' in r['Name'] or 'fir_tensor' in r['Name']: print(f"   {r['Name'][:40]:40s} avg_us {float(r['AverageNs'])/1e3:9.1f}")
PY
done
