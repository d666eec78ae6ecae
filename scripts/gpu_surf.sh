#!/bin/bash
# SURF batch path: parity on the device, config-4 line, single-lane per-kernel durations by grid (octave)
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/surf"; mkdir -p "$O"; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_surf.py -m gpu -x -q 2>&1 | tail -2
timeout 900 python bench.py --config 4 --no-cpu --steps 3 --warmup 1 --batch 64 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('2 lanes:', d['value'], 'fhog ms/tile', d['roofline']['fhog_ms_per_tile'], 'surf ms/tile', d['roofline']['surf']['ms_per_tile'], d['parity']['parity_sample'])"
for env in "" "IMGFD_SURF_RESIDUE=0"; do
cd /tmp; rm -rf /tmp/prof4
env IMGFD_SURF_LANES=1 $env timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof4 -o p -- python $R/bench.py --config 4 --batch 8 --no-cpu --steps 2 --warmup 1 > /dev/null 2>&1
f=$(find /tmp/prof4 -name '*kernel_trace.csv' | head -1)
echo "--- single lane $env"
python - $f <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
d=collections.defaultdict(list)
for r in rows:
    k=r['Kernel_Name']
    if 'surf' in k:
        d[k[:44]+' grid '+r['Grid_Size_X']+'x'+r['Grid_Size_Y']+'x'+r['Grid_Size_Z']].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
tot=0
for k,v in d.items():
    print(f"{k:80s} n {len(v):4d} avg_us {sum(v)/len(v):8.1f}"); tot+=sum(v)/24
print("sum per tile (24 tiles) us:", round(tot,1))
PY
done
