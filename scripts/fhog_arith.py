"""Time imgfd_fhog_dev on 4096x4096 RGB tiles for values of the lab switch "fhog_arith" (a wave with at least that many lanes
outside the gradient table's LDS centre computes their words instead of gathering them); synthetic tile, uniform noise, and a
tile of natural-image statistics stand-in (smooth + edges); JSON lines."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from image_amd import synth
from image_amd.device import DeviceDetector
S, N = 4096, int(os.environ.get("TILES", 16))
det = DeviceDetector(0); lib, ctx = det.lib, det.ctx.handle
nr, nc = C.c_int(), C.c_int()
lib.imgfd_fhog_size(S, S, 8, 1, 1, C.byref(nr), C.byref(nc))
g = torch.Generator(device="cuda").manual_seed(1)
tiles = {"synthetic": torch.from_numpy(synth.frame_rgb(3, S, S)).cuda(),
         "noise": torch.randint(0, 256, (S, S, 3), dtype=torch.uint8, device="cuda", generator=g)}
smooth = torch.nn.functional.avg_pool2d(tiles["noise"].permute(2, 0, 1).float()[None], 9, 1, 4)[0].permute(1, 2, 0)
tiles["noise_smoothed_9x9"] = smooth.round().to(torch.uint8).contiguous()
for name, tile in tiles.items():
    frames = tile.unsqueeze(0).repeat(N, 1, 1, 1).contiguous()
    out = torch.empty((N, 31, nc.value, nr.value), dtype=torch.float32, device="cuda")
    def run():
        det.ctx.check(lib.imgfd_fhog_dev(ctx, frames.data_ptr(), N, S, S, S * S * 3, 8, 1, 1, out.data_ptr()), "fhog_dev")
    ref = None
    for lanes in (0, 1, 8, 16, 24, 32, 48, 64, 0):
        det.ctx.check(lib.imgfd_set_tuning(ctx, b"fhog_arith", lanes), "fhog_arith")
        for _ in range(2): run()
        torch.cuda.synchronize()
        if ref is None: ref = out[0].clone()
        same = bool(torch.equal(out[0].view(torch.int32), ref.view(torch.int32)))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(6): run()
        e1.record(); e1.synchronize()
        print(json.dumps({"tile": name, "tiles": N, "fhog_arith": lanes, "us_per_tile": round(1e3 * e0.elapsed_time(e1) / 6 / N, 1), "bits_equal": same}), flush=True)
