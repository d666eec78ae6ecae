#!/bin/bash
# Every kernel of config 4 (fHOG + SURF on 4096^2 RGB tiles): duration (kernel trace), HBM-side traffic and VALU / LDS
# instruction counts (separate PMC passes, no tracing beside them).  -> $1/pmc_config4.txt     TILES (default 8)
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; O="${1:-$R/gpurun_out/pmc_c4}"; mkdir -p "$O"; export TMPDIR=/tmp
T="${TILES:-8}"
cd /tmp; rm -rf /tmp/pc4
B="python $R/bench.py --config 4 --no-cpu --no-dist --batch $T"
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/pc4/t -o p -- $B --steps 4 --warmup 1 > /dev/null 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d /tmp/pc4/a -o p -- $B --steps 1 --warmup 1 > /dev/null 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pc4/b -o p -- $B --steps 1 --warmup 1 > /dev/null 2>&1
timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES --output-format csv -d /tmp/pc4/c -o p -- $B --steps 1 --warmup 1 > /dev/null 2>&1
timeout 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY --output-format csv -d /tmp/pc4/d -o p -- $B --steps 1 --warmup 1 > /dev/null 2>&1
TILES_N=$T python - > "$O/pmc_config4.txt" <<'PY'
import csv, glob, collections, re, os
T = int(os.environ["TILES_N"])
def short(k):
    k = re.sub(r'^void ', '', k); k = re.sub(r'\(anonymous namespace\)::', '', k); return re.split(r'\(', k)[0][:46]
dur = collections.defaultdict(list)
for fn in glob.glob('/tmp/pc4/t/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(fn)): dur[short(r['Kernel_Name'])].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
for d in 'abcd':
    for fn in glob.glob(f'/tmp/pc4/{d}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(fn)): cnt[short(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
px = 4096 * 4096
steps = 5.0  # traced steps (4 timed + 1 warm-up)
print(f"config 4 (bench.py --config 4 --batch {T}): one 4096x4096 RGB tile = {px} px; per TILE: us = total kernel time of a step / tiles;")
print("FETCH = 2 x FETCH_SIZE KiB (gfx950 correction), WRITE = WRITE_SIZE KiB, both per tile pixel; VALU/px, LDS/px = wave instructions x 64 / tile pixels;")
print("issue = 4 x SQ_INSTS_VALU / (1024 SIMDs x GRBM_GUI_ACTIVE / 8); conflict = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE")
print(f"{'kernel':48s} {'launch/tile':>11s} {'us/launch':>10s} {'us/tile':>8s} {'FETCH B/px':>10s} {'WRITE B/px':>10s} {'VALU/px':>8s} {'LDS/px':>7s} {'issue':>6s} {'conflict':>8s}")
rows = []
for k, v in dur.items():
    if k.startswith('synth') or 'at::' in k or 'rocclr' in k or 'elementwise' in k: continue
    c = cnt.get(k, {})
    per_tile = len(v) / steps / T
    s = lambda n: sum(c[n]) if n in c and c[n] else float('nan')
    runs = 2.0 * T  # counter passes: 2 steps (1 timed + 1 warm-up) of T tiles
    f, w = 2 * 1024 * s('FETCH_SIZE') / runs, 1024 * s('WRITE_SIZE') / runs
    valu, lds, gui = s('SQ_INSTS_VALU') / runs, s('SQ_INSTS_LDS') / runs, s('GRBM_GUI_ACTIVE')
    issue = 4 * s('SQ_INSTS_VALU') / (1024 * gui / 8) if gui == gui and gui else float('nan')
    conf = s('SQ_LDS_BANK_CONFLICT') / s('SQ_LDS_IDX_ACTIVE') if s('SQ_LDS_IDX_ACTIVE') else float('nan')
    us_tile = sum(v) / steps / T
    rows.append((us_tile, f"{k:48s} {per_tile:11.2f} {sum(v)/len(v):10.1f} {us_tile:8.1f} {f/px:10.2f} {w/px:10.2f} {valu*64/px:8.1f} {lds*64/px:7.1f} {issue:6.2f} {conf:8.3f}"))
for _, line in sorted(rows, reverse=True): print(line)
print(f"sum over kernels: {sum(r[0] for r in rows):.1f} us per tile (kernels of the two SURF lanes overlap in wall time)")
PY
cat "$O/pmc_config4.txt"
