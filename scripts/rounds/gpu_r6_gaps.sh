#!/bin/bash
# round 6: gaps between the dependent kernels of one detector's chain on one stream (one 4K frame, nothing beside it)
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$R/gpurun_out/r6gaps; mkdir -p $O; cd /tmp
for w in harris canny; do
rm -rf /tmp/gp; WHICH=$w timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/gp -o p -- python $R/scripts/b1_chain_gaps.py > /tmp/gp.log 2>&1
python - $w <<'PY' | tee -a $O/gaps.txt
import csv, glob, re, sys
rows = []
for fn in glob.glob('/tmp/gp/**/*kernel_trace.csv', recursive=True): rows += list(csv.DictReader(open(fn)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
rows = [r for r in rows if 'synth' not in r['Kernel_Name'] and 'at::' not in r['Kernel_Name']]
last = rows[-int(len(rows) / 20) * 2:]
print("==", sys.argv[1], "(last two calls): gap_before_us dur_us kernel")
prev = None
for r in last:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    k = re.sub(r'\(.*', '', r['Kernel_Name'].replace('void ', ''))[:44]
    print(f"{(s - prev) / 1e3 if prev else 0:8.1f} {(e - s) / 1e3:8.1f}  {k}")
    prev = e
PY
done
