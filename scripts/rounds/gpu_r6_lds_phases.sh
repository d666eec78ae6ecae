#!/bin/bash
# round 6: the first octave's kernel (surf_pyramid_lds<0>) split into its phases: the product library against
# scripts/variants/lib_noload.so (-DSURF_LDS_EXPERIMENT_NO_LOAD: no window load) and lib_nomath.so (-DSURF_LDS_EXPERIMENT_NO_MATH:
# window load + stores, no filter arithmetic), one tile per call, kernel times from rocprofv3 --kernel-trace --stats
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$R/gpurun_out/r6lds; mkdir -p $O; cd /tmp
for v in ${VARIANTS:-default noload nomath}; do
  lib=""; [ $v != default ] && lib=$R/scripts/variants/lib_$v.so
  rm -rf /tmp/lp_$v
  VARIANT_LIB=$lib TILES=${TILES:-1} timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lp_$v -o p -- python $R/scripts/surf_dev_time.py 2>&1 | grep "^{" | cut -c1-100
  f=$(find /tmp/lp_$v -name "*kernel_stats.csv" | head -1)
  echo "== $v" | tee -a $O/phases.txt
  python - "$f" <<'PY' | tee -a $O/phases.txt
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "surf_pyramid" in r["Name"]: print("%-40s calls %s avg_us %.1f" % (r["Name"][:40], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
