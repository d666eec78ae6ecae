#!/bin/bash
# per-launch timeline of imgfd_canny_dev alone (BATCH frames, one stream): every kernel's duration in launch order for one call
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O="$R/gpurun_out/canny"; mkdir -p "$O"
for B in ${BATCHES:-32 1}; do
cd /tmp; rm -rf /tmp/ct
BATCH=$B ITERS=3 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/ct -o p -- python $R/scripts/canny_time.py 2>&1 | grep canny_ms
python - $B <<'PY' | tee "$O/timeline_b$B.txt"
import csv, glob, re, sys
rows = []
for fn in glob.glob('/tmp/ct/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(fn)): rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
idx = [i for i, r in enumerate(rows) if 'canny_blur_march' in r[2]]
i0, i1 = idx[-2], idx[-1]
t0 = rows[i0][0]
print(f"# one imgfd_canny_dev call, batch {sys.argv[1]}: offset_us dur_us kernel")
for s, e, k in rows[i0:i1]:
    k = re.sub(r'^void ', '', k); k = re.split(r'\(', k)[0][:50]
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f}  {k}")
PY
done
cd $R; echo "=== serpentine"; BATCH=1 timeout 300 python scripts/canny_serpentine_time.py 2>&1 | tail -1; BATCH=8 timeout 300 python scripts/canny_serpentine_time.py 2>&1 | tail -1
