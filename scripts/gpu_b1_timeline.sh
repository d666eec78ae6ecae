#!/bin/bash
# timeline of ONE pass of the single-frame case (bench.py --batch 1): every kernel's start offset and duration
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/b1
IMGFD_DETECT_GRAPH=${GRAPH:-0} timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/b1 -o p -- python $R/bench.py --no-extra --no-cpu --no-dist --batch 1 --inner 4 --steps 3 --warmup 2 > /dev/null 2>&1
python - <<'PY'
import csv, glob, re
rows = []
for fn in glob.glob('/tmp/b1/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(fn)): rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Stream_Id', r.get('Queue_Id', '?'))))
rows.sort()
# last pass: find the last fast9_tile launch and print from there
idx = [i for i, r in enumerate(rows) if 'canny_blur_march' in r[2]]
i0 = idx[-1]
# include kernels that started slightly earlier on the other stream
start = min(rows[i][0] for i in range(max(0, i0 - 2), i0 + 1))
sel = [r for r in rows if r[0] >= start - 2000]
t0 = sel[0][0]
print("offset_us  dur_us  queue  kernel")
for s, e, k, q in sel:
    k = re.sub(r'^void ', '', k); k = re.sub(r'\(anonymous namespace\)::', '', k); k = re.split(r'\(', k)[0][:50]
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f}  {q:>5s}  {k}")
print("pass length us:", (sel[-1][1] - t0) / 1e3)
PY
