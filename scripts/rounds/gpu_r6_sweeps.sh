#!/bin/bash
# round 6: one 4K frame per pass (bench.py --batch 1) with fewer sweep launches queued before the union-find step
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$R/gpurun_out/r6sweeps; mkdir -p $O
for s in 0 1 2 3 4 5 8; do
  echo -n "IMGFD_HYST_SWEEPS=$s " | tee -a $O/sweeps.txt
  IMGFD_HYST_SWEEPS=$s timeout 200 python bench.py --batch 1 --inner 50 --steps 5 --no-cpu 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step']/50, d.get('parity'))" | cut -c1-150 | tee -a $O/sweeps.txt
done
