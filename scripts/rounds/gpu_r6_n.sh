#!/bin/bash
# round 6, call N: the first octave stores only what the maximum test can read: SURF tests on the device, the pyramid kernels alone,
# short probes, the bench's config 4 (sustained, with parity), written bytes of the kernel (PMC WRITE_SIZE, its own pass)
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$R/gpurun_out/r6n; mkdir -p $O
timeout 900 python -m pytest tests/test_surf.py tests/test_bench_line.py -q -m gpu -x --timeout 300 2>&1 | tail -2 | tee $O/pytest_surf.txt
TILES=2 IMGFD_SURF_LANES=1 IMGFD_SURF_GROUP=2 VARIANTS="default" bash scripts/rounds/gpu_r6_lds_phases.sh 2>&1 | grep -v "^{" | tee $O/alone.txt
for t in 1 16 64; do echo -n "tiles=$t " | tee -a $O/batch.txt; TILES=$t timeout 200 python scripts/surf_dev_time.py 2>&1 | grep "^{" | cut -c1-60 | tee -a $O/batch.txt; done
cd $R; timeout 900 python bench.py --config 4 --no-cpu 2>/dev/null | tail -1 > $O/bench_config4.json
python -c "import json; d=json.loads(open('$O/bench_config4.json').read()); print(d['value'], d['roofline']['surf']['ms_per_tile'], d['roofline']['fhog_ms_per_tile'], json.dumps(d['summary'].get('parity')))" | cut -c1-400 | tee $O/config4.txt
cd /tmp; rm -rf /tmp/wr; TILES=8 timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/wr -o p -- python $R/scripts/surf_dev_time.py > /dev/null 2>&1
python - <<'PY' | tee $O/write_size.txt
import csv, glob, collections
f = glob.glob("/tmp/wr/**/*counter_collection.csv", recursive=True)
tot = collections.defaultdict(float); n = collections.defaultdict(int)
for r in csv.DictReader(open(f[0])):
    if "surf_pyramid" in r["Kernel_Name"]: tot[r["Kernel_Name"][:30]] += float(r["Counter_Value"]); n[r["Kernel_Name"][:30]] += 1
for k in tot: print(k, "WRITE_SIZE KiB per launch", round(tot[k] / n[k], 1), "B/px", round(tot[k] / n[k] * 1024 / (4096 * 4096), 3))
PY
