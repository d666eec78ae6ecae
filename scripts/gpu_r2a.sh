#!/bin/bash
# round 2, GPU call A: new structure-tensor kernel -- parity on the device, variant timings, bench lines, 2-rank check
set -u
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r2a"; mkdir -p "$O"
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_harris_stages.py tests/test_harris_api.py tests/test_full_size.py -m gpu -x -q 2>&1 | tail -15 ) > "$O/pytest_harris.txt" 2>&1
{
  echo "--- default (new kernel)";           timeout 300 python scripts/k3_variants.py
  echo "--- old (round-1 kernel)";           IMGFD_TENSOR_IMPL=old timeout 300 python scripts/k3_variants.py
  echo "--- wide (float4 rows through LDS)"; IMGFD_TENSOR_IMPL=wide timeout 300 python scripts/k3_variants.py
  echo "--- no xcd remap";                   IMGFD_XCD_REMAP=0 timeout 300 python scripts/k3_variants.py
  echo "--- strict";                         FIR_MODE=0 BATCHES=32 timeout 300 python scripts/k3_variants.py
  echo "--- seg 242";                        IMGFD_TENSOR_SEG=242 BATCHES=1,32 timeout 300 python scripts/k3_variants.py
} > "$O/k3_variants.txt" 2>&1
timeout 600 python bench.py --no-cpu > "$O/bench_default.json" 2> "$O/bench_default.err"
IMGFD_NO_FUSED_RESPONSE=1 timeout 600 python bench.py --no-cpu > "$O/bench_nofuse.json" 2> "$O/bench_nofuse.err"
timeout 600 python bench.py --no-cpu --batch 1 --inner 50 > "$O/bench_b1.json" 2> "$O/bench_b1.err"
timeout 600 python bench.py --gpus 2 --share-device --no-cpu --batch 8 --steps 3 --inner 2 > "$O/bench_2ranks.json" 2> "$O/bench_2ranks.err"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_onestream" -o p -- python $R/bench.py --no-cpu --no-overlap --steps 3 --warmup 1 --inner 2 > "$O/prof_onestream.log" 2>&1
f=$(find "$O/prof_onestream" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$O/bench_kernel_stats_one_stream.csv"
rm -rf "$O/prof_onestream"
# counters of the new kernel (separate passes, no tracing)
pmc() {
  name=$1; shift
  BATCHES=32 ITERS=4 timeout 600 rocprofv3 --pmc "$@" --output-format csv -d "$O/pmc_$name" -o p -- python $R/scripts/k3_variants.py > "$O/pmc_$name.log" 2>&1
  f=$(find "$O/pmc_$name" -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" "fir_tensor" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r.get("Kernel_Name", "")
    if sys.argv[2] not in k: continue
    agg[k[:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    for c, v in d.items():
        print(f"   {c:28s} mean {sum(v)/len(v):16.1f}  n={len(v)}")
PY
  rm -rf "$O/pmc_$name"
}
{
echo "=== PMC pass 1"; pmc a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS
echo "=== PMC pass 2"; pmc b SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY
} > "$O/k3_pmc.txt" 2>&1
exit 0
