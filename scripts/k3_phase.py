"""Experiment: per-phase shader-clock attribution of fir_march<7,tensor> (library built with EXTRA=-DFIR_PROFILE)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from image_amd import synth
from image_amd.device import DeviceDetector
NX, NY = 3840, 2160
det = DeviceDetector(0); lib, ctx = det.lib, det.ctx.handle
raw = C.CDLL(lib._name)
img = torch.from_numpy(synth.frame(2, NX, NY).astype(np.float32)).cuda()
Is, Ix, Iy = (torch.empty_like(img) for _ in range(3))
det.ctx.check(lib.imgfd_k_gaussian(ctx, img.data_ptr(), Is.data_ptr(), NX, NY, C.c_float(1.0), 0), "gauss")
det.ctx.check(lib.imgfd_k_gradient(ctx, Is.data_ptr(), Ix.data_ptr(), Iy.data_ptr(), NX, NY, 0), "grad")
torch.cuda.synchronize()
det.ctx.set_fir_mode(1)
out = (C.c_ulonglong * 8)()
us = det.time_structure_tensor(Ix, Iy, 2.5, 0, warmup=3, iters=3)
raw.imgfd_debug_fir_profile(out, 1)
iters = 20
us = det.time_structure_tensor(Ix, Iy, 2.5, 0, warmup=0, iters=iters)
raw.imgfd_debug_fir_profile(out, 0)
names = ["prefetch0", "commit(+vmcnt wait)", "barrier1", "prefetch issue", "row pass", "barrier2", "col pass", "-"]
tot = sum(out)
print(f"avg {us:.1f} us/launch (instrumented)")
for n, v in zip(names, out):
    print(f"  {n:22s} {100.0 * v / tot:5.1f}%   {v / iters / 2040:10.0f} ticks/wave/launch")
