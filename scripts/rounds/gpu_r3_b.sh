#!/bin/bash
# round 3, GPU call B: fused fHOG kernel (parity + timing of the lab switches) and its counters
set -u
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r3b"; mkdir -p "$O"; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_fhog.py tests/test_full_size.py -m gpu -q -x -k fhog 2>&1 | tail -4 ) > "$O/pytest_fhog.txt" 2>&1
TILES=${TILES:-16,1} timeout 600 python scripts/fhog_variants.py > "$O/fhog_variants.txt" 2>&1
NOISE=1 TILES=16 timeout 600 python scripts/fhog_variants.py >> "$O/fhog_variants.txt" 2>&1
[ -n "${PMC:-}" ] && bash scripts/gpu_pmc_c4.sh "$O" > /dev/null 2>&1
cat "$O/pytest_fhog.txt" "$O/fhog_variants.txt"; [ -n "${PMC:-}" ] && grep "fhog\|kernel " "$O/pmc_config4.txt"
exit 0
