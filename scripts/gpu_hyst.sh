#!/bin/bash
# hysteresis: duration of every canny_hyst_block launch of one imgfd_canny_dev call, in launch order (32 frames 4K)
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
python scripts/canny_time.py 2>/dev/null | grep canny_ms
cd /tmp; rm -rf /tmp/ph
ITERS=2 timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/ph -o p -- python $R/scripts/canny_time.py > /dev/null 2>&1
python - <<'PY'
import csv, glob
rows = []
for fn in glob.glob('/tmp/ph/**/*kernel_trace.csv', recursive=True):
    rows += list(csv.DictReader(open(fn)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
seq = [(r['Kernel_Name'][:22], (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3) for r in rows if 'canny' in r['Kernel_Name']]
# last call only
idx = [i for i, (k, _) in enumerate(seq) if k.startswith('void canny_blur_march') or k.startswith('canny_front')]
last = seq[idx[-1]:]
print(' '.join(f"{k.split('(')[0][6:14]}:{d:.0f}" for k, d in last))
print('hyst total us', sum(d for k, d in last if 'hyst' in k), 'all', sum(d for _, d in last))
PY
