#!/bin/bash
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/co
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/co -o p -- python $R/scripts/k3_corun_probe.py > /dev/null 2>&1
python - <<'PY'
import csv, glob, re
rows = []
for fn in glob.glob('/tmp/co/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(fn)): rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '?')))
rows.sort()
k3 = [r for r in rows if 'fir_tensor' in r[2]][-1]
print("K3 from 0 to %.1f us" % ((k3[1] - k3[0]) / 1e3))
for s, e, k, q in rows:
    if e < k3[0] - 300e3 or s > k3[1] + 100e3: continue
    k = re.sub(r'^void ', '', k); k = re.split(r'\(', k)[0][:40]
    print(f"{(s - k3[0]) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  q{q}  {k}")
PY
