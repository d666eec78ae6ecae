#!/bin/bash
# imgfd_surf_dev on 16 tiles, ONE lane (every kernel alone on the device): average duration per kernel, for each value of a switch
# usage: gpu_surf_kstats.sh IMGFD_SURF_ENDS 1 0
cd /tmp; export TMPDIR=/tmp; R="$GRAFT_REPO_ROOT"; VAR="$1"; shift
for v in "$@"; do
  rm -rf /tmp/sp
  env $VAR=$v IMGFD_SURF_LANES=1 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sp -o p -- python $R/scripts/surf_dev_time.py > /dev/null 2>&1
  f=$(find /tmp/sp -name "*kernel_stats.csv" | head -1)
  echo "$VAR=$v (one lane, us per launch)"
  python $R/scripts/kstats.py $f 2>/dev/null | head -14 || python - $f <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:14]:
    print("  %-46s calls %4s avg_us %8.1f" % (r["Name"][:46], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
