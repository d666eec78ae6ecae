#!/bin/bash
# round 3, GPU call A: the fused fHOG kernel on the device (parity, sqrt, timing of the lab switches) and the counter table
# of config 4
set -u
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r3a"; mkdir -p "$O"; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_fhog.py tests/test_full_size.py -m gpu -q -x 2>&1 | tail -8 ) > "$O/pytest_fhog.txt" 2>&1
timeout 600 python scripts/fhog_variants.py > "$O/fhog_variants.txt" 2>&1
NOISE=1 TILES=16 timeout 600 python scripts/fhog_variants.py >> "$O/fhog_variants.txt" 2>&1
bash scripts/gpu_pmc_c4.sh "$O" > /dev/null 2>&1
IMGFD_FHOG_FUSED=0 bash scripts/gpu_pmc_c4.sh "$O/stage" > /dev/null 2>&1
cat "$O/pytest_fhog.txt" "$O/fhog_variants.txt" "$O/pmc_config4.txt"
exit 0
