#!/bin/bash
# round 6: the fused fHOG histogram kernel (fhog_hist8) with a phase compiled out (scripts/variants/lib_fh_*.so,
# -DFH_EXPERIMENT_NO_PHASE1 / _NO_PHASE2; results are garbage, the time is the point): 16 tiles 4096x4096, kernel averages
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$R/gpurun_out/r6fh; mkdir -p $O; cd /tmp
for v in default fh_nop1 fh_nop2 fh_none; do
  lib=""; [ $v != default ] && lib=$R/scripts/variants/lib_$v.so
  rm -rf /tmp/fh_$v
  echo "== $v" | tee -a $O/phases.txt
  VARIANT_LIB=$lib TILES=16 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fh_$v -o p -- python $R/scripts/fhog_time.py 2>&1 | grep "^{" | tee -a $O/phases.txt
  f=$(find /tmp/fh_$v -name "*kernel_stats.csv" | head -1)
  python - "$f" <<'PY' | tee -a $O/phases.txt
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "fhog" in r["Name"]: print("%-40s calls %s avg_us per 16 tiles %.1f" % (r["Name"][:40], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
