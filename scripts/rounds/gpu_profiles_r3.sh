#!/bin/bash
# One GPU call that collects what profiles/r03/ holds: pytest -m gpu, smoke, the default bench line (all BASELINE configs and
# the CPU legs inside it), rocprofv3 kernel stats of the default command (two-stream and one-stream) and of configs 3 / 4,
# the structure-tensor PMC passes, the all-kernel counter tables of the default step and of config 4, the single-frame timeline.
# Usage on the box: bash scripts/rounds/gpu_profiles_r3.sh [tag]   -> gpurun_out/prof_<tag>/
set -u
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; TAG="${1:-r3}"; O="$R/gpurun_out/prof_$TAG"; mkdir -p "$O"
export TMPDIR=/tmp
( timeout 1800 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -3 ) > "$O/pytest_gpu.txt" 2>&1
( timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 ) > "$O/smoke.txt" 2>&1
timeout 1500 python bench.py > "$O/bench.json" 2> "$O/bench.err"
timeout 900 python bench.py --no-overlap --no-cpu --no-extra > "$O/bench_one_stream.json" 2>> "$O/bench.err"
timeout 900 python bench.py --gpus 2 --share-device --no-cpu --batch 16 --steps 5 --inner 4 > "$O/bench_2ranks_one_device.json" 2>> "$O/bench.err"
cd /tmp
for mode in two_stream one_stream; do
  extra=""; [ $mode = one_stream ] && extra="--no-overlap"
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_$mode" -o p -- python $R/bench.py --no-cpu --no-extra --no-dist --steps 5 --warmup 2 --inner 2 $extra > "$O/prof_$mode.log" 2>&1
  f=$(find "$O/prof_$mode" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$O/bench_kernel_stats_$mode.csv"
  rm -rf "$O/prof_$mode" "$O/prof_$mode.log"
done
for c in 3 4; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_c$c" -o p -- python $R/bench.py --config $c --no-cpu --no-dist --steps 2 --warmup 1 --batch $([ $c = 3 ] && echo 256 || echo 16) > "$O/prof_c$c.log" 2>&1
  f=$(find "$O/prof_c$c" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$O/config${c}_kernel_stats.csv"
  rm -rf "$O/prof_c$c" "$O/prof_c$c.log"
done
cd "$R"
bash scripts/gpu_pmc_k3.sh > /dev/null 2>&1
cp gpurun_out/k3/k3_pmc.txt "$O/k3_pmc_summary.txt"; cp gpurun_out/k3/k3_traffic.json "$O/k3_traffic.json"
BATCHES=1,8,32 python scripts/k3_variants.py 2>/dev/null | grep kernel > "$O/k3_doorway.txt"
bash scripts/gpu_pmc_all.sh "$O" > /dev/null 2>&1
bash scripts/gpu_pmc_c4.sh "$O" > /dev/null 2>&1
bash scripts/gpu_b1_timeline.sh > "$O/single_frame_timeline.txt" 2>&1
bash scripts/gpu_b32_timeline.sh > "$O/two_stream_timeline.txt" 2>&1
IMGFD_SURF_LANES=1 python scripts/surf_dev_time.py > "$O/surf_one_lane.txt" 2>/dev/null
python scripts/surf_dev_time.py > "$O/surf_two_lanes.txt" 2>/dev/null
TILES1=1 python scripts/surf_dev_time.py > "$O/surf_single_tile.txt" 2>/dev/null
TILES=16,1 python scripts/fhog_variants.py > "$O/fhog_variants.txt" 2>/dev/null
NOISE=1 TILES=16 python scripts/fhog_variants.py >> "$O/fhog_variants.txt" 2>/dev/null
exit 0
