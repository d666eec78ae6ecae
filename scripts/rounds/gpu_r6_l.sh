#!/bin/bash
# round 6, call L: new defaults of imgfd_surf_dev (groups of 4 on three streams): SURF tests, short probes, the bench's config 4 (with its CPU leg and parity)
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$R/gpurun_out/r6l; mkdir -p $O
timeout 900 python -m pytest tests/test_surf.py tests/test_bench_line.py -q -m gpu -x --timeout 300 2>&1 | tail -2 | tee $O/pytest_surf.txt
for t in 1 4 16 64; do echo -n "tiles=$t " | tee -a $O/batch.txt; TILES=$t timeout 200 python scripts/surf_dev_time.py 2>&1 | grep "^{" | cut -c1-60 | tee -a $O/batch.txt; done
timeout 900 python bench.py --config 4 2>/dev/null | tail -1 > $O/bench_config4.json
python -c "import json; d=json.loads(open('$O/bench_config4.json').read()); print(d['value'], d['roofline']['surf']['ms_per_tile'], d['roofline']['fhog_ms_per_tile'], d.get('parity'))" | cut -c1-400
