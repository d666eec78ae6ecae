#!/bin/bash
# single 4K frame (bench.py --batch 1): per-kernel stats, one stream
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
cd /tmp; rm -rf /tmp/pb1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb1 -o p -- python $R/bench.py --no-cpu --no-overlap --batch 1 --steps 10 --warmup 2 --inner 5 > /dev/null 2>&1
f=$(find /tmp/pb1 -name "*kernel_stats.csv" | head -1)
python - $f <<'PY'
import csv,sys
tot=0
for r in list(csv.DictReader(open(sys.argv[1]))):
    if 'synth' in r['Name'] or 'at::' in r['Name']: continue
    c=int(r['Calls']); a=float(r['AverageNs'])/1e3
    per=c/60.0*a  # 60 passes traced (12 steps x 5)
    tot+=per
    if per>1: print(f"{r['Name'][:56]:56s} calls/pass {c/60:5.1f} avg_us {a:8.1f} us/pass {per:7.1f}")
print('sum us/pass', round(tot,1))
PY
cd $R; python bench.py --no-cpu --batch 1 --inner 50 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('two-stream batch 1:', d['value'], d['ms_per_step']/50*1000, 'us/frame')"
