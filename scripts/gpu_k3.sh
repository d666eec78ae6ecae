#!/bin/bash
# structure-tensor kernel: device parity of the stage tests, variant timings, PMC passes (separate, no tracing)
set -u
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/k3"; mkdir -p "$O"
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_harris_stages.py -m gpu -x -q 2>&1 | tail -5 ) > "$O/pytest.txt" 2>&1
{
  echo "--- default"; timeout 300 python scripts/k3_variants.py
  for v in ${VARIANTS:-}; do echo "--- $v"; env $v timeout 300 python scripts/k3_variants.py; done
} > "$O/k3_variants.txt" 2>&1
[ "${PMC:-1}" = "0" ] && exit 0
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
cd /tmp
{
echo "=== PMC pass 1"; pmc a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS
echo "=== PMC pass 2"; pmc b SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY
} > "$O/k3_pmc.txt" 2>&1
exit 0
