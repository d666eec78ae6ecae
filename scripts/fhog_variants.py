"""Time imgfd_fhog_dev on 4096x4096 RGB tiles for the lab switches of the fused kernel (imgfd_set_tuning); JSON lines.
TILES (default 16), SIZE (4096), NOISE=1: uniform-noise tile (every gradient, worst case for the orientation table)."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from image_amd import synth, _lib
if os.environ.get("VARIANT_LIB"):   # experiment builds of the library (scripts/variants/*.so, not tracked)
    _lib.LIB_PATH = os.path.abspath(os.environ["VARIANT_LIB"])
from image_amd.device import DeviceDetector
S = int(os.environ.get("SIZE", 4096))
det = DeviceDetector(0); lib, ctx = det.lib, det.ctx.handle
if os.environ.get("NOISE"):
    tile = torch.randint(0, 256, (S, S, 3), dtype=torch.uint8, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
else:
    tile = torch.from_numpy(synth.frame_rgb(3, S, S)).cuda()
nr, nc = C.c_int(), C.c_int()
lib.imgfd_fhog_size(S, S, 8, 1, 1, C.byref(nr), C.byref(nc))
ref = None
for N in [int(x) for x in os.environ.get("TILES", "16,1").split(",")]:
    frames = tile.unsqueeze(0).repeat(N, 1, 1, 1).contiguous()
    out = torch.empty((N, 31, nc.value, nr.value), dtype=torch.float32, device="cuda")
    def run():
        det.ctx.check(lib.imgfd_fhog_dev(ctx, frames.data_ptr(), N, S, S, S * S * 3, 8, 1, 1, out.data_ptr()), "fhog_dev")
    for fused, bands, sq in [(0, 0, 256), (1, 0, 256), (1, 1, 256), (1, 2, 256), (1, 4, 256), (1, 8, 256), (1, 16, 256), (1, 0, 512)]:
        for k, v in (("fhog_fused", fused), ("fhog_bands", bands), ("fhog_threads", sq)):
            det.ctx.check(lib.imgfd_set_tuning(ctx, k.encode(), v), k)
        for _ in range(2): run()
        torch.cuda.synchronize()
        if ref is None: ref = out[0].clone()
        same = bool(torch.equal(out[0].view(torch.int32), ref.view(torch.int32)))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = max(3, 64 // N)
        e0.record()
        for _ in range(reps): run()
        e1.record(); e1.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print(json.dumps({"tiles": N, "fused": fused, "bands": bands, "threads": sq, "us_per_tile": round(1e3 * ms / N, 1),
                          "bits_equal_to_stage_kernels": same, "noise": bool(os.environ.get("NOISE"))}), flush=True)
