#!/bin/bash
# One GPU iteration: (optional) parity tests, bench, per-kernel trace.  Usage: gpu_iter.sh [tests-expr|none] [tag]
set -u
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out"; mkdir -p "$O"
export TMPDIR=/tmp
T="${1:-all}"; TAG="${2:-it}"
if [ "$T" != "none" ]; then
  if [ "$T" = "all" ]; then timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
  else timeout 900 python -m pytest tests -m gpu -x -q -k "$T" 2>&1 | tail -8; fi
fi
echo "=== bench"; timeout 600 python bench.py --steps 10 --warmup 2 ${BENCH_ARGS:-} 2>&1 | tail -1 | tee "$O/bench_$TAG.json"
echo "=== kernel trace"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace_$TAG" -o t -- python "$R/bench.py" --steps 3 --warmup 1 --no-cpu > "$O/trace_$TAG.log" 2>&1
f=$(find "$O/trace_$TAG" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cut -d, -f1-4 "$f" | cut -c1-150 | head -24
exit 0
