#!/bin/bash
# Round 6's snapshot in one GPU call -> gpurun_out/prof_<tag>/ (copy what should be judged to profiles/r06/<tag>_*):
# pytest -m gpu, smoke(), the default bench line (all configs + CPU legs), rocprofv3 kernel stats of the bench and of config 4,
# K3 counters + traffic, function traffic (PMC), the all-kernel counter table of config 4, timelines (one pass at batch 1, one group of
# imgfd_surf_dev, one tile), the imgfd_surf_i32 host probe.
# Usage on the box: bash scripts/rounds/gpu_profiles_r6.sh [tag]
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; TAG="${1:-a}"; O=$R/gpurun_out/prof_$TAG; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu --timeout 400 > $O/pytest_gpu_full.txt 2>&1; grep -E "passed|failed|error" $O/pytest_gpu_full.txt | tail -2 | tee $O/pytest_gpu.txt
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep "smoke" ) | tee $O/smoke.txt
timeout 1200 python bench.py 2>/dev/null | tail -1 > $O/bench.json
python - <<PY | tee $O/summary.txt
import json
d = json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print(json.dumps(d["summary"]))
PY
( cd /tmp; rm -rf /tmp/kb; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kb -o p -- python $R/bench.py --steps 4 --no-cpu --no-extra > /dev/null 2>&1; cp $(find /tmp/kb -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
  rm -rf /tmp/kc; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kc -o p -- python $R/bench.py --config 4 --batch 32 --steps 3 --warmup 1 --no-cpu > /dev/null 2>&1; cp $(find /tmp/kc -name "*kernel_stats.csv" | head -1) $O/config4_kernel_stats.csv )
bash scripts/gpu_pmc_k3.sh > /dev/null 2>&1; cp gpurun_out/k3/k3_traffic.json $O/ 2>/dev/null; cp gpurun_out/k3/k3_pmc.txt $O/k3_pmc_summary.txt 2>/dev/null
bash scripts/gpu_pmc_functions.sh 2>&1 | tail -3 > $O/function_traffic.txt; cp gpurun_out/traffic/function_traffic.json $O/
TILES=8 timeout 900 bash scripts/gpu_pmc_c4.sh $O > /dev/null 2>&1
timeout 300 bash scripts/gpu_b1_timeline.sh 2>/dev/null | grep -v "fir_tensor<7, 256, true, true, 0>\|copyBuffer\|at::native\|fir_march\|gradient_kernel\|clock_probe" > $O/single_frame_timeline.txt
TILES=32 TAG=g LAST=60 bash scripts/rounds/gpu_r6_tl.sh > /dev/null 2>&1; cp gpurun_out/r6tl/timeline_g.txt $O/surf_group_timeline.txt
TILES1=1 TAG=one LAST=14 bash scripts/rounds/gpu_r6_tl.sh > /dev/null 2>&1; cp gpurun_out/r6tl/timeline_one.txt $O/surf_single_tile_timeline.txt
for t in 64 16; do echo -n "tiles $t " >> $O/surf_per_tile.txt; TILES=$t timeout 200 python scripts/surf_dev_time.py 2>&1 | grep "^{" >> $O/surf_per_tile.txt; done
echo -n "single tile " >> $O/surf_per_tile.txt; TILES1=1 timeout 200 python scripts/surf_dev_time.py 2>&1 | grep "^{" >> $O/surf_per_tile.txt
( rm -rf /tmp/sp; cd /tmp; IMGFD_SURF_GROUP=1 IMGFD_SURF_LANES=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sp -o p -- python $R/scripts/surf_dev_time.py > /dev/null 2>&1; python $R/scripts/kstats.py $(find /tmp/sp -name "*kernel_stats.csv" | head -1) | head -16 > $O/surf_one_tile_at_a_time_kernel_stats.txt )
timeout 200 python scripts/surf_host_probe.py 2>&1 | tail -1 > $O/surf_host_probe.txt
TILES=4 timeout 200 python scripts/fhog_time.py 2>&1 | tail -1 > $O/fhog_time.txt
ls $O
