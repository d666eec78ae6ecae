#!/bin/bash
# timeline of ONE pass of the default bench step (32 frames, two streams): every kernel's start offset and duration
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/b32
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/b32 -o p -- python $R/bench.py --no-extra --no-cpu --no-dist --inner 4 --steps 2 --warmup 1 > /dev/null 2>&1
python - <<'PY'
import csv, glob, re
rows = []
for fn in glob.glob('/tmp/b32/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(fn)): rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Stream_Id', r.get('Queue_Id', '?'))))
rows.sort()
idx = [i for i, r in enumerate(rows) if 'canny_blur_march' in r[2]]
# the last complete pass inside the timed region: from the second-to-last blur to the last blur
i0, i1 = idx[-2], idx[-1]
t0 = rows[i0][0]
print("offset_us  dur_us  queue  kernel   (one pass of the timed region, 32 frames, two streams)")
hy = []
for s, e, k, q in rows[i0 - 1:i1 + 2]:
    k = re.sub(r'^void ', '', k); k = re.split(r'\(', k)[0][:50]
    if 'canny_hyst_' in k:
        hy.append((s, e)); continue
    if hy:
        print(f"{(hy[0][0] - t0) / 1e3:9.1f} {(hy[-1][1] - hy[0][0]) / 1e3:7.1f}  {'':>5s}  canny_hyst_block x {len(hy)} (sum of durations {sum(b - a for a, b in hy) / 1e3:.1f})"); hy = []
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f}  {q:>5s}  {k}")
PY
