#!/bin/bash
# maxima kernel cost against the threshold (mask density)
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for thr in 1e30 3000 300 30 3; do
cd /tmp; rm -rf /tmp/prof5
env IMGFD_SURF_LANES=1 PYTHONPATH=$R timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof5 -o p -- python $R/scripts/surf_thr.py $thr 2>&1 | grep -i "threshold\|error" | head -3
f=$(find /tmp/prof5 -name '*kernel_trace.csv' | head -1)
python - $f <<'PY'
import csv,sys,collections
d=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    k=r['Kernel_Name']
    if 'surf_nms' in k or 'rank' in k or 'surf_pyramid<' in k: d[k[:30]].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in d.items(): print(f"   {k:32s} n {len(v):3d} avg_us {sum(v)/len(v):8.1f}")
PY
done
