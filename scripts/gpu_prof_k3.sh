#!/bin/bash
# Per-kernel trace of the bench pipeline + PMC passes on the structure-tensor kernel.
set -u
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out"; mkdir -p "$O"
export TMPDIR=/tmp
cd /tmp
echo "=== kernel trace (bench, 3 steps)"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace_bench" -o t -- python "$R/bench.py" --steps 3 --warmup 1 --no-cpu > "$O/trace_bench.log" 2>&1
f=$(find "$O/trace_bench" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cat "$f" | cut -c1-200 | head -30
pmc() { # name, counters...
  name=$1; shift
  timeout 600 rocprofv3 --pmc "$@" --output-format csv -d "$O/pmc_$name" -o p -- python "$R/scripts/k3_time.py" > "$O/pmc_$name.log" 2>&1
  f=$(find "$O/pmc_$name" -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r.get("Kernel_Name", "")
    if "fir_march" not in k: continue
    agg[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    for c, v in d.items():
        print(f"   {c:28s} mean {sum(v)/len(v):16.1f}  n={len(v)}")
PY
}
echo "=== PMC pass 1"; pmc a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS
echo "=== PMC pass 2"; pmc b SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY
echo "=== PMC pass 3"; pmc c FETCH_SIZE GRBM_GUI_ACTIVE
echo "=== PMC pass 4"; pmc d WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
exit 0
