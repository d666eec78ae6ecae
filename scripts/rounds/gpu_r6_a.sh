#!/bin/bash
# round 6, call A: GPU test-suite on the new SURF table layout (residue only), per-kernel SURF times one lane (round-5 library
# vs this tree), batch / single-tile timings of both, a short config-4 line
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$R/gpurun_out/r6a; mkdir -p $O
timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -5 | tee $O/pytest_gpu.txt
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 ) | tee $O/smoke.txt
for lib in scripts/variants/lib_r05.so image_amd/libimgfd.so; do
  tag=$(basename $lib .so)
  echo "== $tag" | tee -a $O/surf.txt
  VARIANT_LIB=$R/$lib timeout 200 python scripts/surf_dev_time.py 2>&1 | grep "^{" | tee -a $O/surf.txt
  VARIANT_LIB=$R/$lib timeout 200 python scripts/surf_dev_time.py 2>&1 | grep "^{" | tee -a $O/surf.txt
  VARIANT_LIB=$R/$lib TILES1=1 timeout 200 python scripts/surf_dev_time.py 2>&1 | grep "^{" | tee -a $O/surf.txt
  rm -rf /tmp/sp
  ( cd /tmp; VARIANT_LIB=$R/$lib IMGFD_SURF_LANES=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sp -o p -- python $R/scripts/surf_dev_time.py > /dev/null 2>&1 )
  f=$(find /tmp/sp -name "*kernel_stats.csv" | head -1)
  echo "-- one lane, per kernel ($tag)" | tee -a $O/surf.txt
  python scripts/kstats.py $f 2>/dev/null | head -14 | tee -a $O/surf.txt
  cp $f $O/kernel_stats_one_lane_$tag.csv 2>/dev/null
done
timeout 600 python bench.py --config 4 --batch 32 --steps 2 --warmup 1 --no-cpu 2>/dev/null | tail -1 > $O/bench_c4.json
python - <<PY | tee $O/c4_summary.txt
import json
d = json.loads(open("$O/bench_c4.json").read().strip().splitlines()[-1])
print("config4 batch32", d["value"], d["ms_per_step"], d["roofline"]["surf"]["ms_per_tile"], d["roofline"]["fhog_ms_per_tile"], d.get("parity"))
PY
