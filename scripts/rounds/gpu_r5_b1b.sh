#!/bin/bash
# round 5, single 4K frame, second pass: fused finish kernel, block sweeps with shifted groupings, sweep priority, K3 worker count, gates
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$R/gpurun_out/r5_b1b; mkdir -p $O
for hb in 11 22 44; do
  echo "pytest canny, hyst_block $hb" | tee -a $O/pytest_canny.txt
  IMGFD_HYST_BLOCK=$hb timeout 600 python -m pytest tests/test_canny.py tests/test_full_size.py -x -q -m gpu -k "canny or Canny" 2>&1 | tail -3 | tee -a $O/pytest_canny.txt
done
run() {  # label, env...
  local label="$1"; shift
  echo -n "$label  " | tee -a $O/variants.txt
  env "$@" timeout 200 python bench.py --batch ${BATCH:-1} --no-cpu --no-extra --no-dist --steps 10 --warmup 3 --inner 50 2>/dev/null < /dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('Gpx/s', round(d['value']/1e3,2), ' us/frame', round(d['ms_per_step']/50*1000/d['config'].get('batch',1) if False else d['ms_per_step']/50*1000,1))" | tee -a $O/variants.txt
}
: > $O/variants.txt
run "r04: 11, 10 sweeps, 3-kernel finish, no defer " IMGFD_DETECT_DEFER=0 IMGFD_HYST_BLOCK=11 IMGFD_CANNY_FINISH=0
run "11, fused finish                              " IMGFD_DETECT_DEFER=0 IMGFD_HYST_BLOCK=11
run "11, fused finish, defer                       " IMGFD_HYST_BLOCK=11
run "22 s6                                         " IMGFD_HYST_BLOCK=22 IMGFD_HYST_SWEEPS=6
run "22 s6 prio0                                   " IMGFD_HYST_BLOCK=22 IMGFD_HYST_SWEEPS=6 IMGFD_HYST_PRIO=0
run "22 s6 no defer                                " IMGFD_HYST_BLOCK=22 IMGFD_HYST_SWEEPS=6 IMGFD_DETECT_DEFER=0
run "44 s4                                         " IMGFD_HYST_BLOCK=44 IMGFD_HYST_SWEEPS=4
run "24 s6                                         " IMGFD_HYST_BLOCK=24 IMGFD_HYST_SWEEPS=6
for tw in 224 192 160 128; do
run "22 s6 K3 workers $tw                          " IMGFD_HYST_BLOCK=22 IMGFD_HYST_SWEEPS=6 IMGFD_TENSOR_WORKERS=$tw
done
run "44 s4 K3 workers 192                          " IMGFD_HYST_BLOCK=44 IMGFD_HYST_SWEEPS=4 IMGFD_TENSOR_WORKERS=192
for g in "0 0" "0 1" "0 2" "1 1" "1 2" "2 1"; do set -- $g
run "22 s6 canny_gate $1 harris_gate $2             " IMGFD_HYST_BLOCK=22 IMGFD_HYST_SWEEPS=6 IMGFD_CANNY_GATE=$1 IMGFD_HARRIS_GATE=$2
run "22 s6 canny_gate $1 harris_gate $2 K3w 192     " IMGFD_HYST_BLOCK=22 IMGFD_HYST_SWEEPS=6 IMGFD_CANNY_GATE=$1 IMGFD_HARRIS_GATE=$2 IMGFD_TENSOR_WORKERS=192
done
run "22 s6 graph                                   " IMGFD_HYST_BLOCK=22 IMGFD_HYST_SWEEPS=6 IMGFD_DETECT_GRAPH=8
echo "--- batch 2 and 4" | tee -a $O/variants.txt
for b in 2 4; do
BATCH=$b run "b$b r04 path                                  " IMGFD_DETECT_DEFER=0 IMGFD_HYST_BLOCK=11 IMGFD_CANNY_FINISH=0
BATCH=$b run "b$b 11 fused defer                            " IMGFD_HYST_BLOCK=11
BATCH=$b run "b$b 22 s6                                     " IMGFD_HYST_BLOCK=22 IMGFD_HYST_SWEEPS=6
BATCH=$b run "b$b 22 s7 w4                                  " IMGFD_HYST_BLOCK=22 IMGFD_HYST_SWEEPS=7 IMGFD_HYST_WORDS=4
done
echo "--- batch 32 default bench" | tee -a $O/variants.txt
BATCH=32 run "b32 r04 path                                 " IMGFD_HYST_BLOCK=11 IMGFD_CANNY_FINISH=0
BATCH=32 run "b32 11 fused                                 " IMGFD_HYST_BLOCK=11
BATCH=32 run "b32 22 w4 s8                                 " IMGFD_HYST_BLOCK=22 IMGFD_HYST_SWEEPS=8 IMGFD_HYST_WORDS=4
BATCH=32 run "b32 24 w2 s7                                 " IMGFD_HYST_BLOCK=24 IMGFD_HYST_SWEEPS=7 IMGFD_HYST_WORDS=2
python scripts/canny_serpentine_time.py 2>&1 | tail -5 | tee $O/serpentine.txt
