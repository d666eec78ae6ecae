#!/bin/bash
# PMC passes on the standalone K3 timing loop (scripts/k3_time.py).  Separate passes, no tracing domains.
set -u
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out"; mkdir -p "$O"
export TMPDIR=/tmp
TAG="${1:-k3}"; KERN="${2:-fir_march<7}"; SCRIPT="${3:-scripts/k3_time.py}"
cd /tmp
pmc() {
  name=$1; shift
  timeout 600 rocprofv3 --pmc "$@" --output-format csv -d "$O/pmc_${TAG}_$name" -o p -- python $R/$SCRIPT > "$O/pmc_${TAG}_$name.log" 2>&1
  f=$(find "$O/pmc_${TAG}_$name" -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" "$KERN" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r.get("Kernel_Name", "")
    if sys.argv[2] not in k: continue
    agg[k[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
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
