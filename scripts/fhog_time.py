"""Time imgfd_fhog_dev on 4096x4096 RGB tiles (BASELINE config 4) with HIP events; prints one JSON line."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_amd import _lib
if os.environ.get("VARIANT_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["VARIANT_LIB"])
import numpy as np, torch
from image_amd import synth
from image_amd.device import DeviceDetector
N, S = int(os.environ.get("TILES", 4)), int(os.environ.get("SIZE", 4096))
det = DeviceDetector(0); lib, ctx = det.lib, det.ctx.handle
tile = torch.from_numpy(synth.frame_rgb(3, S, S)).cuda()
frames = tile.unsqueeze(0).repeat(N, 1, 1, 1).contiguous()
nr, nc = C.c_int(), C.c_int()
lib.imgfd_fhog_size(S, S, 8, 1, 1, C.byref(nr), C.byref(nc))
out = torch.empty((N, 31, nc.value, nr.value), dtype=torch.float32, device="cuda")
def run():
    det.ctx.check(lib.imgfd_fhog_dev(ctx, frames.data_ptr(), N, S, S, S * S * 3, 8, 1, 1, out.data_ptr()), "fhog_dev")
for _ in range(2): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
h = lib.imgfd_ctx_stream(ctx)
st = torch.cuda.ExternalStream(h) if h else torch.cuda.current_stream()  # DeviceDetector adopts torch's current stream
with torch.cuda.stream(st):
    e0.record(); 
    for _ in range(5): run()
    e1.record()
e1.synchronize()
ms = e0.elapsed_time(e1) / 5
alg = (3 * S * S + 31 * 4 * nr.value * nc.value) * N
print(json.dumps({"kernel": "fhog (K13-K15)", "tiles": N, "size": S, "ms_per_batch": round(ms, 3), "Mpix_s": round(N * S * S / ms / 1e3, 1),
                  "algorithmic_GBps": round(alg / ms / 1e6, 1)}))
