#!/bin/bash
# whole-step view: GPU parity of the named test files, bench (default + one-stream), rocprofv3 kernel stats of the one-stream run
set -u
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/step"; mkdir -p "$O"
export TMPDIR=/tmp
( timeout 1200 python -m pytest ${TESTS:-tests/test_canny.py tests/test_full_size.py} -m gpu -x -q 2>&1 | tail -6 ) > "$O/pytest.txt" 2>&1
timeout 600 python bench.py --no-cpu > "$O/bench.json" 2> "$O/bench.err"
timeout 600 python bench.py --no-cpu --no-overlap > "$O/bench_one.json" 2> "$O/bench_one.err"
timeout 600 python bench.py --no-cpu --batch 1 --inner 50 > "$O/bench_b1.json" 2> "$O/bench_b1.err"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof" -o p -- python $R/bench.py --no-cpu --no-overlap --steps 3 --warmup 1 --inner 2 > "$O/prof.log" 2>&1
f=$(find "$O/prof" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$O/kernel_stats_one_stream.csv"
rm -rf "$O/prof"
exit 0
