#!/bin/bash
# Every kernel of the default bench step (32 x 4K frames, one stream): duration (kernel trace), HBM-side traffic and VALU
# instruction counts (PMC passes, separate from the trace and from each other).  -> $1/pmc_all_kernels.txt
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; O="${1:-$R/gpurun_out/pmc_all}"; mkdir -p "$O"; export TMPDIR=/tmp
cd /tmp; rm -rf /tmp/pall
B="python $R/bench.py --no-cpu --no-extra --no-dist --no-overlap --inner 1"
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/pall/t -o p -- $B --steps 8 --warmup 2 > /dev/null 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d /tmp/pall/a -o p -- $B --steps 2 --warmup 1 > /dev/null 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pall/b -o p -- $B --steps 2 --warmup 1 > /dev/null 2>&1
timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES --output-format csv -d /tmp/pall/c -o p -- $B --steps 2 --warmup 1 > /dev/null 2>&1
python - > "$O/pmc_all_kernels.txt" <<'PY'
import csv, glob, collections, re
def short(k):
    k = re.sub(r'^void ', '', k); return re.split(r'\(', k)[0][:44]
dur = collections.defaultdict(list)
for fn in glob.glob('/tmp/pall/t/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(fn)): dur[short(r['Kernel_Name'])].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
for d in 'abc':
    for fn in glob.glob(f'/tmp/pall/{d}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(fn)): cnt[short(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
px = 32 * 3840 * 2160
print("default bench step, 32 frames 3840x2160 per launch, one stream; per launch (median duration; counters: mean over launches)")
print("FETCH = 2 x FETCH_SIZE KiB (gfx950 correction), WRITE = WRITE_SIZE KiB; VALU/px = SQ_INSTS_VALU x 64 / pixels; issue = 4 x SQ_INSTS_VALU / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)")
print(f"{'kernel':46s} {'launches':>8s} {'us':>9s} {'FETCH GB':>9s} {'B/px':>6s} {'WRITE GB':>9s} {'B/px':>6s} {'VALU/px':>8s} {'LDS/px':>7s} {'issue':>6s}")
rows = []
for k, v in dur.items():
    if k.startswith('synth') or 'at::' in k or 'rocclr' in k: continue
    v = sorted(v); c = cnt.get(k, {})
    m = lambda n: (sum(c[n]) / len(c[n])) if n in c and c[n] else float('nan')
    f, w = 2 * 1024 * m('FETCH_SIZE'), 1024 * m('WRITE_SIZE')
    valu, lds, gui = m('SQ_INSTS_VALU'), m('SQ_INSTS_LDS'), m('GRBM_GUI_ACTIVE')
    per = len(v) / 10.0  # launches per step (10 steps traced)
    rows.append((v[len(v)//2] * per, f"{k:46s} {per:8.1f} {v[len(v)//2]:9.1f} {f/1e9:9.3f} {f/px:6.2f} {w/1e9:9.3f} {w/px:6.2f} {valu*64/px:8.1f} {lds*64/px:7.1f} {4*valu/(1024*gui/8):6.2f}"))
for _, line in sorted(rows, reverse=True): print(line)
PY
cat "$O/pmc_all_kernels.txt"
