#!/bin/bash
# Harris batch path: parity on the device, NMS from threshold bits vs the tiled kernel (IMGFD_NMS_TILED=1), per-kernel times
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
python -m pytest tests/test_harris_stages.py tests/test_harris_api.py tests/test_full_size.py tests/test_sub_batches.py tests/test_fuzz_sizes.py -m gpu -x -q 2>&1 | tail -2
for v in "" "IMGFD_NMS_TILED=1"; do
  env $v python bench.py --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$v', d['value'], d['ms_per_step'], d['parity'] if 'parity' in d else '')" | cut -c1-300
done
cd /tmp; rm -rf /tmp/pn
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pn -o p -- python $R/bench.py --no-cpu --no-overlap --steps 4 --warmup 1 --inner 1 > /dev/null 2>&1
f=$(find /tmp/pn -name "*kernel_stats.csv" | head -1)
python - $f <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    print(f"{r['Name'][:60]:60s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:10.1f}")
PY
