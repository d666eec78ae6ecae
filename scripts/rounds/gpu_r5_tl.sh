#!/bin/bash
# timeline of one single-frame pass under the current defaults (+ env given on the command line)
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$R/gpurun_out/r5_tl; mkdir -p $O
cd /tmp; rm -rf /tmp/b1
env "$@" timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/b1 -o p -- python $R/bench.py --no-extra --no-cpu --no-dist --batch ${BATCH:-1} --inner 4 --steps 3 --warmup 2 > /dev/null 2>&1
python - <<'PY' | tee $O/timeline.txt
import csv, glob, re
rows = []
for fn in glob.glob('/tmp/b1/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(fn)): rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Stream_Id', r.get('Queue_Id', '?'))))
rows.sort()
idx = [i for i, r in enumerate(rows) if 'canny_blur_march' in r[2]]
i0, i1 = idx[-2], idx[-1]
sel = rows[i0:i1 + 1]
t0 = sel[0][0]
print("offset_us  dur_us  queue  kernel   (one pass: from a blur launch to the next)")
for s, e, k, q in sel:
    k = re.sub(r'^void ', '', k); k = re.sub(r'\(anonymous namespace\)::', '', k); k = re.split(r'\(', k)[0][:50]
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f}  {q:>5s}  {k}")
PY
cd $R; python bench.py --batch ${BATCH:-1} --no-cpu --no-extra --no-dist --steps 10 --warmup 3 --inner 50 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('untraced: Gpx/s', round(d['value']/1e3,2), ' us/pass', round(d['ms_per_step']/50*1000,1))" | tee -a $O/timeline.txt
