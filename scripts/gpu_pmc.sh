#!/bin/bash
# gpu_pmc.sh "<kernel name regex>" <python script + args...>: PMC passes (separate, no tracing) over the kernels a script launches
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/pmc"; mkdir -p "$O"; export TMPDIR=/tmp
FILT="$1"; shift
cd /tmp
pass() {
  rm -rf /tmp/pmcx
  ITERS=2 BATCHES=32 timeout 600 rocprofv3 --pmc "$@" --output-format csv -d /tmp/pmcx -o p -- python $R/$CMD > /dev/null 2>&1
  python - "$FILT" <<'PY'
import csv, sys, glob, collections, re
filt = re.compile(sys.argv[1])
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob('/tmp/pmcx/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(fn)):
        k = r['Kernel_Name']
        if filt.search(k): agg[k[:48]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    print(k, ' '.join(f"{c}={sum(v)/len(v):.4g}" for c, v in d.items()))
PY
}
CMD="$*"
pass SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS
pass SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY
pass FETCH_SIZE GRBM_GUI_ACTIVE
pass WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
