#!/bin/bash
# One GPU call that collects what profiles/r02/ holds: pytest -m gpu, smoke, the bench lines of every config (CPU legs
# included), rocprofv3 kernel stats of the default command (two-stream and one-stream), the structure-tensor PMC passes.
# Usage on the box: bash scripts/gpu_profiles.sh [tag]   -> gpurun_out/prof_<tag>/
set -u
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; TAG="${1:-final}"; O="$R/gpurun_out/prof_$TAG"; mkdir -p "$O"
export TMPDIR=/tmp
( timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -6 ) > "$O/pytest_gpu.txt" 2>&1
( timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 ) > "$O/smoke.txt" 2>&1
timeout 900 python bench.py > "$O/bench.json" 2> "$O/bench.err"
timeout 900 python bench.py --batch 1 --inner 50 > "$O/bench_config2_batch1.json" 2>> "$O/bench.err"
timeout 900 python bench.py --no-overlap --no-cpu > "$O/bench_one_stream.json" 2>> "$O/bench.err"
timeout 900 python bench.py --config 3 > "$O/bench_config3.json" 2>> "$O/bench.err"
timeout 1200 python bench.py --config 4 --steps 5 --warmup 1 > "$O/bench_config4.json" 2>> "$O/bench.err"
timeout 1200 python bench.py --config 5 > "$O/bench_config5.json" 2>> "$O/bench.err"
timeout 900 python bench.py --gpus 2 --share-device --no-cpu --batch 16 --steps 5 > "$O/bench_2ranks_one_device.json" 2>> "$O/bench.err"
cd /tmp
for mode in two_stream one_stream; do
  extra=""; [ $mode = one_stream ] && extra="--no-overlap"
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_$mode" -o p -- python $R/bench.py --no-cpu --steps 5 --warmup 2 --inner 2 $extra > "$O/prof_$mode.log" 2>&1
  f=$(find "$O/prof_$mode" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$O/bench_kernel_stats_$mode.csv"
  rm -rf "$O/prof_$mode"
done
for c in 3 4; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_c$c" -o p -- python $R/bench.py --config $c --no-cpu --steps 2 --warmup 1 --batch $([ $c = 3 ] && echo 256 || echo 16) > "$O/prof_c$c.log" 2>&1
  f=$(find "$O/prof_c$c" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$O/config${c}_kernel_stats.csv"
  rm -rf "$O/prof_c$c"
done
cd "$R"
bash scripts/gpu_pmc_k3.sh > /dev/null 2>&1
cp gpurun_out/k3/k3_pmc.txt "$O/k3_pmc_summary.txt"; cp gpurun_out/k3/k3_traffic.json "$O/k3_traffic.json"
BATCHES=1,8,32 python scripts/k3_variants.py 2>/dev/null | grep kernel > "$O/k3_doorway.txt"
bash scripts/gpu_pmc_all.sh "$O" > /dev/null 2>&1
bash scripts/gpu_surf.sh > "$O/surf_kernels.txt" 2>&1
bash scripts/gpu_hyst.sh > "$O/canny_launches.txt" 2>&1
exit 0
