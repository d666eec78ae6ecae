#!/bin/bash
# The reduced end-of-round snapshot (profiles/r04/c_*): what changed after snapshot b is SURF's ranking kernel and the boundary
# checks, so: pytest -m gpu, smoke, the default bench line (all configs), config 4's kernel stats and counter table, SURF timings.
set -u
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; TAG="${1:-c}"; O="$R/gpurun_out/prof_$TAG"; mkdir -p "$O"
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -3 ) > "$O/pytest_gpu.txt" 2>&1 < /dev/null
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 ) > "$O/smoke.txt" 2>&1 < /dev/null
timeout 900 python bench.py > "$O/bench.json" 2> "$O/bench.err" < /dev/null
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_c4" -o p -- python $R/bench.py --config 4 --no-cpu --no-dist --steps 2 --warmup 1 --batch 16 > "$O/prof_c4.log" 2>&1 < /dev/null
f=$(find "$O/prof_c4" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$O/config4_kernel_stats.csv"
rm -rf "$O/prof_c4" "$O/prof_c4.log"
cd "$R"
timeout 600 bash scripts/gpu_pmc_c4.sh "$O" > /dev/null 2>&1 < /dev/null
IMGFD_SURF_LANES=1 timeout 120 python scripts/surf_dev_time.py > "$O/surf_one_lane.txt" 2>/dev/null < /dev/null
timeout 120 python scripts/surf_dev_time.py > "$O/surf_two_lanes.txt" 2>/dev/null < /dev/null
TILES1=1 timeout 120 python scripts/surf_dev_time.py > "$O/surf_single_tile.txt" 2>/dev/null < /dev/null
exit 0
