#!/bin/bash
# Round 4, structure tensor: the wave-autonomous kernel (fir_tensor_wave.hip) against the workgroup-marching one, in ONE GPU
# call (boxes differ by +-3 %): parity tests, doorway / in-pipeline timings per variant library, the FT_PROFILE phase split of
# the workgroup kernel, PMC passes of the wave kernel.  Output: gpurun_out/k3w/
set -u
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/k3w"; mkdir -p "$O"
export TMPDIR=/tmp
{
echo "=== pytest (wave kernel + harris stages)"
timeout 600 python -m pytest tests/test_tensor_wave.py tests/test_harris_stages.py -m gpu -x -q 2>&1 | tail -4
echo "=== doorway (20 B/px), HIP events"
IMGFD_TENSOR_WAVE=0 BATCHES=1,32 timeout 200 python scripts/k3_variants.py 2>&1 | grep structure_tensor
IMGFD_TENSOR_WAVE=1 BATCHES=1,32 timeout 200 python scripts/k3_variants.py 2>&1 | grep structure_tensor
for v in scripts/variants/lib_w_*.so; do
  VARIANT_LIB=$v IMGFD_TENSOR_WAVE=1 BATCHES=1,32 timeout 200 python scripts/k3_variants.py 2>&1 | grep structure_tensor
done
echo "=== imgfd_harris_dev, 32 frames (response variant in the pipeline)"
IMGFD_TENSOR_WAVE=0 timeout 200 python scripts/harris_time.py 2>&1 | tail -1
IMGFD_TENSOR_WAVE=1 timeout 200 python scripts/harris_time.py 2>&1 | tail -1
for v in scripts/variants/lib_w_*.so; do
  VARIANT_LIB=$v IMGFD_TENSOR_WAVE=1 timeout 200 python scripts/harris_time.py 2>&1 | tail -1
done
echo "=== FT_PROFILE phase split of the workgroup kernel (fir_tensor<7,256,fma,vec,0>)"
VARIANT_LIB=scripts/variants/lib_ftprof.so IMGFD_TENSOR_WAVE=0 timeout 200 python scripts/k3_phase.py 2>&1 | tail -10
echo "=== bench, tensor_wave 0 / 1"
IMGFD_TENSOR_WAVE=0 timeout 400 python bench.py --no-cpu 2>&1 | tail -1 | cut -c1-1500
IMGFD_TENSOR_WAVE=1 timeout 400 python bench.py --no-cpu 2>&1 | tail -1 | cut -c1-1500
} > "$O/log.txt" 2>&1
echo "=== PMC, wave kernel" >> "$O/log.txt"
IMGFD_TENSOR_WAVE=1 bash scripts/gpu_pmc_k3.sh fir_tensor_wave >> "$O/log.txt" 2>&1
cp gpurun_out/k3/k3_pmc.txt "$O/pmc_wave.txt" 2>/dev/null
cp gpurun_out/k3/k3_traffic.json "$O/traffic_wave.json" 2>/dev/null
tail -60 "$O/log.txt"
exit 0
