"""How many sweeps does the hysteresis of a frame need under each grouping of the tiles?  (diagnostic counters
"canny_sweeps_working" / "canny_frames_unconverged"; 14 sweeps queued; imgfd_canny_dev alone, HIP events)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
from image_amd.device import DeviceDetector
NX, NY = int(os.environ.get("NX", 3840)), int(os.environ.get("NY", 2160))
det = DeviceDetector(0)
def counter(name):
    v = C.c_int64(0); det.lib.imgfd_get_counter(det.ctx.handle, name.encode(), C.byref(v)); return int(v.value)
def tune(name, v): det.ctx.check(det.lib.imgfd_set_tuning(det.ctx.handle, name.encode(), int(v)), name)
for B in (1, 32):
    frames = det.synth_frames(B, NX, NY, seed0=50000)
    edges = torch.empty_like(frames); counts = torch.zeros(B, dtype=torch.int64, device="cuda")
    for hb, hw, shift in ((11, 2, 0), (11, 4, 0), (22, 2, 0), (22, 2, 1), (42, 2, 1), (24, 2, 1), (44, 2, 0), (44, 2, 1), (22, 4, 1), (44, 4, 1), (42, 4, 1)):
        tune("hyst_block", hb); tune("hyst_words", hw); tune("hyst_shift", shift); tune("hyst_sweeps", 14)
        det.canny(frames, out=(edges, counts)); torch.cuda.synchronize()
        need = counter("canny_sweeps_working")
        res = {}
        for q in sorted({need + 1, need + 2, 14}):
            if q > 32: continue
            tune("hyst_sweeps", q)
            for _ in range(3): det.canny(frames, out=(edges, counts))
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            it = 30 if B == 1 else 8
            e0.record()
            for _ in range(it): det.canny(frames, out=(edges, counts))
            e1.record(); e1.synchronize()
            res[q] = round(e0.elapsed_time(e1) / it * 1000, 1)
        print(json.dumps({"batch": B, "block": hb, "words": hw, "shift": shift, "working_sweeps": need, "unconverged_at_14": counter("canny_frames_unconverged"),
                          "canny_us_by_queued_sweeps": res, "edge_pixels": int(counts.sum())}), flush=True)
