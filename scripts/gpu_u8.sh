#!/bin/bash
# u8 tile kernels (fast9_tile, gauss_grad_tile, Canny front): duration and FETCH_SIZE per launch (32 frames 4K) against the
# run length of the tile runs (IMGFD_TILE_RUN; unset = the library's choice).  usage: gpu_u8.sh "<run lengths>" [kernel filter]
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/u8"; mkdir -p "$O"; export TMPDIR=/tmp
RUNS="${1:-default}"; FILT="${2:-fast9_tile|gauss_grad_tile|canny_blur|canny_grad|canny_front}"
cd /tmp
for run in $RUNS; do
  E=""; [ "$run" != default ] && E="IMGFD_TILE_RUN=${run%f}"; [ "${run%f}" != "$run" ] && E="$E IMGFD_TILE_GRID=full"
  rm -rf /tmp/pu8
  env $E timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/pu8/t -o p -- python $R/bench.py --no-cpu --steps 8 --warmup 2 --inner 1 --no-overlap > /dev/null 2>&1
  env $E timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pu8/c -o p -- python $R/bench.py --no-cpu --steps 1 --warmup 1 --inner 1 --no-overlap > /dev/null 2>&1
  python - "$run" "$FILT" <<'PY'
import csv, sys, glob, collections, re
run, filt = sys.argv[1], re.compile(sys.argv[2])
t = collections.defaultdict(list); f = collections.defaultdict(list)
for fn in glob.glob('/tmp/pu8/t/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(fn)):
        k = r['Kernel_Name']
        if filt.search(k): t[k[:40]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for fn in glob.glob('/tmp/pu8/c/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(fn)):
        k = r['Kernel_Name']
        if filt.search(k) and r['Counter_Name'] == 'FETCH_SIZE': f[k[:40]].append(float(r['Counter_Value']))
alg = 32 * 3840 * 2160
for k in t:
    fs = f.get(k, [0]); fb = 2 * 1024 * sum(fs) / len(fs)
    v = sorted(t[k])
    print(f"run {run:>7s}  {k:42s} n {len(v):3d} min_us {v[0]:8.1f} median_us {v[len(v)//2]:8.1f}   FETCHx2 {fb/1e9:6.3f} GB = {fb/alg:5.2f} B/px")
PY
done | tee -a "$O/u8_runs.txt"
