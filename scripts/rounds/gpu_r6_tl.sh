#!/bin/bash
# round 6: the launch timeline of imgfd_surf_dev (last call of scripts/surf_dev_time.py): ENV="K=V ..." TILES1=1 for one tile
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$R/gpurun_out/r6tl; mkdir -p $O
cd /tmp; rm -rf /tmp/tl
env $ENV timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o p -- python $R/scripts/surf_dev_time.py > /tmp/tl.log 2>&1
grep "^{" /tmp/tl.log | tee $O/timeline_${TAG:-x}.txt
python - >> $O/timeline_${TAG:-x}.txt <<'PY'
import csv, glob, re
rows = []
for fn in glob.glob('/tmp/tl/**/*kernel_trace.csv', recursive=True):
    rows += list(csv.DictReader(open(fn)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
rows = [r for r in rows if 'surf' in r['Kernel_Name'] or 'fill' in r['Kernel_Name']]
import os
n = int(os.environ.get("LAST", "14"))
last = rows[-n:]
t0 = int(last[0]['Start_Timestamp'])
print("start_us end_us dur_us queue kernel")
for r in last:
    k = re.sub(r'\(.*', '', r['Kernel_Name'].replace('void ', ''))[:40]
    print(f"{(int(r['Start_Timestamp'])-t0)/1e3:9.1f} {(int(r['End_Timestamp'])-t0)/1e3:9.1f} {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:8.1f}  q{r.get('Queue_Id','?')}  {k}")
PY
cat $O/timeline_${TAG:-x}.txt
