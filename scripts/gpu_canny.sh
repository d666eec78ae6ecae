#!/bin/bash
# Canny alone: whole-function time per variant library, and the per-kernel split (rocprofv3 stats) of the default build
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/canny"; mkdir -p "$O"; export TMPDIR=/tmp
python scripts/canny_time.py 2>/dev/null | grep canny_ms
for f in scripts/variants/*.so; do [ -f "$f" ] && VARIANT_LIB=$f python scripts/canny_time.py 2>/dev/null | grep canny_ms; done
cd /tmp
ITERS=3 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof" -o p -- python $R/scripts/canny_time.py > "$O/prof.log" 2>&1
f=$(find "$O/prof" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$O/kernel_stats.csv"; rm -rf "$O/prof"
python - "$O/kernel_stats.csv" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:8]:
    print(f"{r['Name'][:60]:60s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:10.1f}")
PY
