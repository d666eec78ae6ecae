#!/bin/bash
# round 5, single 4K frame (BASELINE configs[1]): parity of the block sweeps first, then bench.py --batch 1 and imgfd_canny_dev alone
# under the lab switches (queue order, graph replay, tiles per workgroup of a sweep, sweeps queued); one GPU call, one box
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$R/gpurun_out/r5_b1; mkdir -p $O
timeout 900 python -m pytest tests/test_canny.py -x -q -m gpu 2>&1 | tail -3 | tee $O/pytest_canny.txt
run() {  # label, env...
  local label="$1"; shift
  echo -n "$label  " | tee -a $O/variants.txt
  env "$@" timeout 200 python bench.py --batch 1 --no-cpu --no-extra --no-dist --steps 10 --warmup 3 --inner 50 2>/dev/null < /dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('Gpx/s', round(d['value']/1e3,2), ' us/frame', round(d['ms_per_step']/50*1000,1))" | tee -a $O/variants.txt
}
: > $O/variants.txt
run "r04 order, tile sweeps       " IMGFD_DETECT_DEFER=0 IMGFD_HYST_BLOCK=11
run "deferred queueing            " IMGFD_DETECT_DEFER=1 IMGFD_HYST_BLOCK=11
run "deferred + graph             " IMGFD_DETECT_DEFER=1 IMGFD_HYST_BLOCK=11 IMGFD_DETECT_GRAPH=8
for hb in 22 42 24 44; do for sw in 3 4 6; do
run "deferred, block $hb sweeps $sw   " IMGFD_DETECT_DEFER=1 IMGFD_HYST_BLOCK=$hb IMGFD_HYST_SWEEPS=$sw
done; done
run "deferred, block 44 w4 sweeps 4 " IMGFD_DETECT_DEFER=1 IMGFD_HYST_BLOCK=44 IMGFD_HYST_SWEEPS=4 IMGFD_HYST_WORDS=4
run "deferred, block 22 w4 sweeps 4 " IMGFD_DETECT_DEFER=1 IMGFD_HYST_BLOCK=22 IMGFD_HYST_SWEEPS=4 IMGFD_HYST_WORDS=4
run "deferred+graph, block 44 sw 4  " IMGFD_DETECT_DEFER=1 IMGFD_HYST_BLOCK=44 IMGFD_HYST_SWEEPS=4 IMGFD_DETECT_GRAPH=8
run "deferred+graph, block 22 sw 4  " IMGFD_DETECT_DEFER=1 IMGFD_HYST_BLOCK=22 IMGFD_HYST_SWEEPS=4 IMGFD_DETECT_GRAPH=8
echo "--- imgfd_canny_dev alone, one frame / batch 32" | tee -a $O/variants.txt
for hb in 11 22 42 24 44; do
  echo -n "block $hb b1: " | tee -a $O/variants.txt; BATCH=1 ITERS=50 IMGFD_HYST_BLOCK=$hb timeout 100 python scripts/canny_time.py 2>/dev/null | tee -a $O/variants.txt
  echo -n "block $hb b32: " | tee -a $O/variants.txt; BATCH=32 ITERS=10 IMGFD_HYST_BLOCK=$hb timeout 100 python scripts/canny_time.py 2>/dev/null | tee -a $O/variants.txt
done
# timelines: r04 order, and the block variants
for v in "0 11 0" "1 22 4" "1 44 4"; do set -- $v
cd /tmp; rm -rf /tmp/b1
IMGFD_DETECT_DEFER=$1 IMGFD_HYST_BLOCK=$2 IMGFD_HYST_SWEEPS=$3 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/b1 -o p -- python $R/bench.py --no-extra --no-cpu --no-dist --batch 1 --inner 4 --steps 3 --warmup 2 > /dev/null 2>&1
python - > $O/timeline_defer$1_block$2.txt <<'PY'
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
cd $R
done
tail -40 $O/timeline_defer1_block22.txt
