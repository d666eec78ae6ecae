"""Time the Harris structure-tensor kernel (and its neighbours) on a 4K frame: HIP events around
back-to-back launches through imgfd_time_structure_tensor.  Prints one JSON line per configuration."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from image_amd import synth
from image_amd.device import DeviceDetector

NX, NY = 3840, 2160
det = DeviceDetector(0)
lib, ctx = det.lib, det.ctx.handle
img = torch.from_numpy(synth.frame(2, NX, NY).astype(np.float32)).cuda()
Is, Ix, Iy = (torch.empty_like(img) for _ in range(3))
det.ctx.check(lib.imgfd_k_gaussian(ctx, img.data_ptr(), Is.data_ptr(), NX, NY, 1.0, 0), "gauss")
det.ctx.check(lib.imgfd_k_gradient(ctx, Is.data_ptr(), Ix.data_ptr(), Iy.data_ptr(), NX, NY, 0), "grad")
torch.cuda.synchronize()
for mode in (1, 0):
    det.ctx.set_fir_mode(mode)
    us = det.time_structure_tensor(Ix, Iy, 2.5, 0, warmup=10, iters=100)
    gbs = 20 * NX * NY / (us * 1e-6) / 1e9
    print(json.dumps({"kernel": "structure_tensor", "fir_mode": mode, "xcd_remap": os.environ.get("IMGFD_XCD_REMAP", "1"),
                      "us": round(us, 2), "algorithmic_GBps": round(gbs, 1), "frac_of_8TBps": round(gbs / 8000, 4)}))
