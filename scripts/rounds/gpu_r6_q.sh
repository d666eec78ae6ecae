#!/bin/bash
# round 6, call Q: SURF tests on the device, the pyramid kernels alone, short probes, the bench's config 4 (sustained) after a kernel change
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$R/gpurun_out/r6q; mkdir -p $O
timeout 900 python -m pytest tests/test_surf.py tests/test_bench_line.py -q -m gpu -x --timeout 300 2>&1 | tail -2 | tee $O/pytest_surf.txt
TILES=2 IMGFD_SURF_LANES=1 IMGFD_SURF_GROUP=2 VARIANTS="default" bash scripts/rounds/gpu_r6_lds_phases.sh 2>&1 | grep -v "^{" | tee $O/alone.txt
for t in 1 64; do echo -n "tiles=$t " | tee -a $O/batch.txt; TILES=$t timeout 200 python scripts/surf_dev_time.py 2>&1 | grep "^{" | cut -c1-60 | tee -a $O/batch.txt; done
cd $R; for i in 1 2; do timeout 300 python bench.py --config 4 --steps 6 --warmup 2 --no-cpu 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); r=d['roofline']; print(d['value'], r['surf']['ms_per_tile'], r['fhog_ms_per_tile'])" | tee -a $O/config4.txt; done
