#!/bin/bash
# First GPU visit: instruction-rate microbench, parity tests, smoke, bench, per-kernel profile.
set -u
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
echo "=== rocminfo" ; /opt/rocm/bin/rocminfo 2>/dev/null | grep -E "Marketing Name|gfx9|Compute Unit" | head -6
echo "=== ubench" ; timeout 120 scripts/ubench/ubench.bin 2>&1 | tee gpurun_out/ubench.txt
echo "=== pytest -m gpu" ; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/pytest_gpu.txt
echo "=== smoke" ; timeout 300 python __graft_entry__.py smoke 2>&1 | tail -5 | tee gpurun_out/smoke.txt
echo "=== k3 variants"
for remap in 1 0; do IMGFD_XCD_REMAP=$remap timeout 300 python scripts/k3_time.py 2>&1 | tail -4; done | tee gpurun_out/k3_time.txt
echo "=== bench" ; timeout 600 python bench.py --steps 10 --warmup 2 2>&1 | tail -3 | tee gpurun_out/bench_fma.json
timeout 600 python bench.py --steps 10 --warmup 2 --fir-mode 0 --no-cpu 2>&1 | tail -1 | tee gpurun_out/bench_strict.json
echo "=== rocprofv3 kernel stats"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof_r1" -o r1 -- python "$GRAFT_REPO_ROOT/bench.py" --steps 5 --warmup 1 --no-cpu > "$GRAFT_REPO_ROOT/gpurun_out/prof_bench.log" 2>&1
cd "$GRAFT_REPO_ROOT"; ls -R gpurun_out/prof_r1 | head -20
f=$(find gpurun_out/prof_r1 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -25 "$f"
