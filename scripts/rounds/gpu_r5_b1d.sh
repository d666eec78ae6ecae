#!/bin/bash
cd "$GRAFT_REPO_ROOT"; R="$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$R/gpurun_out/r5_b1d; mkdir -p $O
IMGFD_HYST_WORDS=1 timeout 900 python -m pytest tests/test_canny.py -x -q -m gpu 2>&1 | tail -2 | tee $O/pytest.txt
run() {  # label, env...
  local label="$1"; shift
  echo -n "$label  " | tee -a $O/variants.txt
  env "$@" timeout 200 python bench.py --batch ${BATCH:-1} --no-cpu --no-extra --no-dist --steps 10 --warmup 3 --inner 50 2>/dev/null < /dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('Gpx/s', round(d['value']/1e3,2), ' us/pass', round(d['ms_per_step']/50*1000,1))" | tee -a $O/variants.txt
}
: > $O/variants.txt
run "defaults (w2 22 s6)   "
for hb in 22 42 24 44; do for sw in 10; do
run "w1 $hb s$sw " IMGFD_HYST_WORDS=1 IMGFD_HYST_BLOCK=$hb IMGFD_HYST_SWEEPS=$sw
done; done
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee $O/probe.txt
import json, os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import ctypes as C
import torch
from image_amd.device import DeviceDetector
det = DeviceDetector(0)
def counter(name):
    v = C.c_int64(0); det.lib.imgfd_get_counter(det.ctx.handle, name.encode(), C.byref(v)); return int(v.value)
def tune(name, v): det.ctx.check(det.lib.imgfd_set_tuning(det.ctx.handle, name.encode(), int(v)), name)
frames = det.synth_frames(1, 3840, 2160, seed0=50000)
edges = torch.empty_like(frames); counts = torch.zeros(1, dtype=torch.int64, device="cuda")
for hb, hw in ((22, 2), (22, 1), (42, 1), (24, 1), (44, 1)):
    tune("hyst_block", hb); tune("hyst_words", hw); tune("hyst_sweeps", 14)
    det.canny(frames, out=(edges, counts)); torch.cuda.synchronize()
    need = counter("canny_sweeps_working")
    res = {}
    for q in (need + 1, need + 2):
        tune("hyst_sweeps", q)
        for _ in range(3): det.canny(frames, out=(edges, counts))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): det.canny(frames, out=(edges, counts))
        e1.record(); e1.synchronize()
        res[q] = round(e0.elapsed_time(e1) / 50 * 1000, 1)
    print(json.dumps({"block": hb, "words": hw, "working_sweeps": need, "canny_us_by_queued_sweeps": res, "edge_pixels": int(counts.sum())}), flush=True)
PY
