#!/bin/bash
# round 6: compile-time variants of surf.hip (scripts/variants/lib_*.so) under the bench's sustained config 4, product library first and last
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$R/gpurun_out/r6vs; mkdir -p $O
V="default $(ls scripts/variants/ | sed -n 's/^lib_\(.*\)\.so$/\1/p' | tr '\n' ' ') default"
for v in $V; do lib=""; [ $v != default ] && lib=$R/scripts/variants/lib_$v.so
  echo -n "$v " | tee -a $O/sustained.txt
  VARIANT_LIB=$lib timeout 300 python scripts/bench_variant.py --config 4 --steps 6 --warmup 2 --no-cpu 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); r=d['roofline']; print(d['value'], r['surf']['ms_per_tile'], r['fhog_ms_per_tile'])" | tee -a $O/sustained.txt
done
