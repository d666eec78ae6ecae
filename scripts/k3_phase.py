"""Experiment: per-phase shader-clock attribution of fir_tensor (library built with `make EXTRA=-DFT_PROFILE`)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from image_amd import _lib
if os.environ.get("VARIANT_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["VARIANT_LIB"])
from image_amd.device import DeviceDetector
NX, NY, B = 3840, 2160, int(os.environ.get("BATCH", "32"))
det = DeviceDetector(0); lib = det.lib
raw = C.CDLL(lib._name)
frames = det.synth_frames(B, NX, NY, seed0=50000)
ix = torch.empty((B, NY, NX), dtype=torch.float32, device="cuda"); iy = torch.empty_like(ix)
for f in range(B): det.gradients_of(frames[f], ix[f], iy[f])
det.ctx.set_fir_mode(1)
out = (C.c_ulonglong * 8)()
us = det.time_structure_tensor_batch(ix, iy, warmup=3, iters=3)
raw.imgfd_debug_ft_profile(out, 1)
iters = 10
us = det.time_structure_tensor_batch(ix, iy, warmup=0, iters=iters)
raw.imgfd_debug_ft_profile(out, 0)
names = ["loop bookkeeping -> barrier 1 arrive", "barrier 1 wait", "small phase (ring reads, commit)", "barrier 2 wait", "prefetch issue", "column pass", "row pass", "-"]
tot = sum(out)
print(f"avg {us:.1f} us/launch (instrumented), batch {B}")
for n, v in zip(names, out):
    print(f"  {n:40s} {100.0 * v / tot:5.1f}%")
