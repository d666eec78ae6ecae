#!/bin/bash
# Rebuild one translation unit with different -D settings on the GPU box and print the kernel trace of a short bench run
# for each.  Usage: gpu_sweep.sh <file.hip> <kernel-name-filter> "<-DA=1>" "<-DA=2 -DB=3>" ...
set -u
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out"; mkdir -p "$O"
export TMPDIR=/tmp
SRC="$1"; FILTER="$2"; shift 2
for FLAGS in "$@"; do
    echo "=== $SRC $FLAGS"
    touch "$R/image_amd/csrc/$SRC"
    make -s -j16 -C "$R/image_amd/csrc" EXTRA="$FLAGS" 2>&1 | grep -E "error" -A3
    tag=$(echo "$FLAGS" | tr -c 'A-Za-z0-9=' '_')
    (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/sweep_$tag" -o t -- python "$R/bench.py" --steps 5 --warmup 2 --no-cpu ${BENCH_ARGS:---no-overlap} > "$O/sweep_$tag.log" 2>&1)
    tail -1 "$O/sweep_$tag.log" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], d['config']['feature_counts'])" 2>/dev/null
    f=$(find "$O/sweep_$tag" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && python "$R/scripts/kstats.py" "$f" | grep -E "$FILTER"
done
exit 0
