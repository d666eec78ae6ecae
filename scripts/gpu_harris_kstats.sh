#!/bin/bash
# per-kernel times of imgfd_harris_dev on 32 4K frames (environment = variant)
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/hk
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/hk -o p -- python $R/scripts/harris_time.py > /dev/null 2>&1
python $R/scripts/kstats.py $(find /tmp/hk -name "*kernel_stats.csv" | head -1) | grep -v "synth\|at::" | head -8
