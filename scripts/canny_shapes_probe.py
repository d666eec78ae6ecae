"""imgfd_canny_dev alone for every (tile words, tiles per workgroup) of the block sweeps, batches 1..32: working sweeps (diagnostic counter) and time with
working + 3 sweeps queued"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
from image_amd.device import DeviceDetector
det = DeviceDetector(0)
def counter(name):
    v = C.c_int64(0); det.lib.imgfd_get_counter(det.ctx.handle, name.encode(), C.byref(v)); return int(v.value)
def tune(name, v): det.ctx.check(det.lib.imgfd_set_tuning(det.ctx.handle, name.encode(), int(v)), name)
for B in (1, 2, 4, 8, 32):
    frames = det.synth_frames(B, 3840, 2160, seed0=50000)
    edges = torch.empty_like(frames); counts = torch.zeros(B, dtype=torch.int64, device="cuda")
    for hw in (1, 2, 4):
        for hb in (22, 42, 24, 44):
            tune("hyst_block", hb); tune("hyst_words", hw); tune("hyst_sweeps", 20)
            det.canny(frames, out=(edges, counts)); torch.cuda.synchronize()
            need = counter("canny_sweeps_working")
            tune("hyst_sweeps", need + 3)
            for _ in range(3): det.canny(frames, out=(edges, counts))
            torch.cuda.synchronize()
            it = max(4, 60 // B)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(it): det.canny(frames, out=(edges, counts))
            e1.record(); e1.synchronize()
            print(json.dumps({"batch": B, "words": hw, "block": hb, "working_sweeps": need, "canny_us_per_frame": round(e0.elapsed_time(e1) / it * 1000 / B, 1)}), flush=True)
