#!/bin/bash
# round 6: compile-time variants (scripts/variants/lib_*.so) under the sustained headline configuration, product library first and last
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$R/gpurun_out/r6vh; mkdir -p $O
V="default $(ls scripts/variants/ | sed -n 's/^lib_\(.*\)\.so$/\1/p' | tr '\n' ' ') default"
for v in $V; do lib=""; [ $v != default ] && lib=$R/scripts/variants/lib_$v.so
  echo -n "$v " | tee -a $O/sustained.txt
  VARIANT_LIB=$lib timeout 300 python scripts/bench_variant.py --no-extra --no-cpu --steps 12 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); r=d['roofline']; print(d['value'], d['ms_per_step'], r['frac'], r.get('in_pipeline',{}).get('avg_launch_us'))" | tee -a $O/sustained.txt
done
