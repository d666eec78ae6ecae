#!/bin/bash
# lab: streaming (non-temporal) stores for the big output planes (IMGFD_NT_OUT, one translation unit per variant library)
cd $GRAFT_REPO_ROOT
for v in "" scripts/variants/lib_nt_*.so ""; do
  echo "--- ${v:-default}"
  VARIANT_LIB=$v timeout 200 python bench.py --no-cpu --no-extra --no-dist --steps 10 --warmup 3 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('step', d['value'], d['ms_per_step'], 'k3 doorway us', d['roofline']['avg_launch_us'], 'in pipeline', d['roofline']['in_pipeline']['avg_launch_us'])"
  VARIANT_LIB=$v timeout 100 python scripts/canny_time.py 2>/dev/null < /dev/null | tail -1
  VARIANT_LIB=$v timeout 100 python scripts/surf_dev_time.py 2>/dev/null < /dev/null | tail -1
done
