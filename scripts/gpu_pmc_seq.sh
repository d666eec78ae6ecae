#!/bin/bash
# PMC counters per launch POSITION: the kernels matching $1 are launched in runs of $2 per call (e.g. 24 hysteresis sweeps);
# prints the mean counter values of the 1st, 2nd, ... launch of a run.  Usage: gpu_pmc_seq.sh pattern period cmd...
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
pat=$1; period=$2; shift 2
cmd=${*:-python $R/scripts/canny_time.py}
cd /tmp
pmc() {
  name=$1; shift
  rm -rf /tmp/ps_$name
  ITERS=2 timeout 200 rocprofv3 --pmc "$@" --output-format csv -d /tmp/ps_$name -o p -- $cmd > /tmp/ps_$name.log 2>&1
  f=$(find /tmp/ps_$name -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" "$pat" "$period" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
period = int(sys.argv[3])
disp = collections.OrderedDict()
for r in rows:
    if sys.argv[2] not in r.get("Kernel_Name", ""): continue
    disp.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(disp)
names = sorted({c for d in disp.values() for c in d})
print("pos  " + "  ".join(f"{n:>20s}" for n in names))
for pos in range(min(period, 8)):
    sel = [disp[i] for k, i in enumerate(ids) if k % period == pos]
    print(f"{pos:3d}  " + "  ".join(f"{sum(d.get(n, 0) for d in sel) / max(1, len(sel)):20.0f}" for n in names))
PY
}
pmc a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU
pmc b SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_SCA
pmc c FETCH_SIZE TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum
