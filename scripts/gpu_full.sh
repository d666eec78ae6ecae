#!/bin/bash
# Full GPU check: all parity tests, smoke, bench (with cpu_baseline), per-kernel trace.  Usage: gpu_full.sh [tag]
set -u
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out"; mkdir -p "$O"
export TMPDIR=/tmp
TAG="${1:-full}"
echo "=== pytest -m gpu"; timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee "$O/pytest_gpu_$TAG.txt"
echo "=== smoke"; timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2 | tee "$O/smoke_$TAG.txt"
echo "=== bench"; timeout 900 python bench.py 2>&1 | tail -1 | tee "$O/bench_$TAG.json"
echo "=== kernel trace"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace_$TAG" -o t -- python "$R/bench.py" --steps 5 --warmup 2 --no-cpu > "$O/trace_$TAG.log" 2>&1
f=$(find "$O/trace_$TAG" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && python "$R/scripts/kstats.py" "$f" | head -24
echo "=== kernel trace, one stream (per-kernel durations without concurrent kernels)"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace1s_$TAG" -o t -- python "$R/bench.py" --steps 5 --warmup 2 --no-cpu --no-overlap > "$O/trace1s_$TAG.log" 2>&1
f=$(find "$O/trace1s_$TAG" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && python "$R/scripts/kstats.py" "$f" | head -16
exit 0
