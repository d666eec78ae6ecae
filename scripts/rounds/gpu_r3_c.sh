#!/bin/bash
# round 3, GPU call C: SURF (band / strip integral image, radix select with shrinking lists, overflow redo): parity, batch and
# single-tile timing, per-kernel counters of config 4
set -u
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/r3c"; mkdir -p "$O"; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_surf.py tests/test_full_size.py -m gpu -q -x -k "surf or integral" 2>&1 | tail -4 ) > "$O/pytest_surf.txt" 2>&1
timeout 600 python scripts/surf_dev_time.py > "$O/surf_time.txt" 2>&1
TILES1=1 timeout 600 python scripts/surf_dev_time.py >> "$O/surf_time.txt" 2>&1
bash scripts/gpu_pmc_c4.sh "$O" > /dev/null 2>&1
cat "$O/pytest_surf.txt" "$O/surf_time.txt" "$O/pmc_config4.txt"
exit 0
