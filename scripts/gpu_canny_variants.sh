#!/bin/bash
# the first sweeps of imgfd_canny_dev (batch 32 / 1) with every scripts/variants/lib_*.so (timing experiments)
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for v in "" scripts/variants/lib_*.so; do
for B in ${BATCHES:-32}; do
cd /tmp; rm -rf /tmp/ct
echo "=== variant '$v' batch $B"
VARIANT_LIB=${v:+$R/$v} BATCH=$B ITERS=3 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/ct -o p -- python $R/scripts/canny_time.py 2>&1 | grep canny_ms | cut -c1-120
python - <<'PY'
import csv, glob, re
rows = []
for fn in glob.glob('/tmp/ct/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(fn)): rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
idx = [i for i, r in enumerate(rows) if 'canny_blur_march' in r[2]]
i0, i1 = idx[-2], idx[-1]
hy = [(e - s) / 1e3 for s, e, k in rows[i0:i1] if 'canny_hyst_' in k]
oth = [(re.split(r'\(', re.sub(r'^void ', '', k))[0][:30], round((e - s) / 1e3, 1)) for s, e, k in rows[i0:i1] if 'canny_hyst_' not in k]
print("  sweeps:", " ".join(f"{d:.1f}" for d in hy[:12]), f"... sum {sum(hy):.1f} us over {len(hy)}")
print("  others:", oth)
PY
done; done
