#!/bin/bash
# surf_rank_select A/B: imgfd_surf_dev per 4096^2 tile -- single tile, default lanes -- and the kernels' own times (single tile)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for v in "" scripts/variants/lib_*.so; do
  echo "--- ${v:-default}"
  VARIANT_LIB=$v TILES1=1 timeout 120 python scripts/surf_dev_time.py 2>&1 | tail -1
  VARIANT_LIB=$v timeout 120 python scripts/surf_dev_time.py 2>&1 | tail -1
  d=/tmp/prof_$(basename ${v:-default} .so)
  (cd /tmp && VARIANT_LIB=${v:+$GRAFT_REPO_ROOT/$v} TILES1=1 timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o t -- python $GRAFT_REPO_ROOT/scripts/surf_dev_time.py >/dev/null 2>&1 </dev/null)
  f=$(find $d -name "*kernel_stats.csv" 2>/dev/null | head -1)
  [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "surf_" in r["Name"]:
        print(f'{r["Name"][:44]:<44} calls {r["Calls"]:>3} avg_us {float(r["AverageNs"])/1e3:8.1f} min_us {float(r["MinNs"])/1e3:8.1f}')
PY
done
