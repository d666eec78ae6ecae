#!/bin/bash
# PMC counters (separate passes, no tracing) of the kernels matching $1 while running $2.. (default: scripts/harris_time.py)
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
pat=$1; shift
cmd=${*:-python $R/scripts/harris_time.py}
cd /tmp
pmc() {
  name=$1; shift
  rm -rf /tmp/pk_$name
  ITERS=3 timeout 150 rocprofv3 --pmc "$@" --output-format csv -d /tmp/pk_$name -o p -- $cmd > /tmp/pk_$name.log 2>&1
  f=$(find /tmp/pk_$name -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" "$pat" <<'PY'
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
pmc a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE
pmc b SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU
pmc c FETCH_SIZE
pmc d WRITE_SIZE   # (FETCH_SIZE and WRITE_SIZE in ONE pass hung rocprofv3 until the timeout)
