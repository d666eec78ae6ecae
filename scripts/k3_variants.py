"""Time the 20 B/px structure-tensor kernel on 4K frames for the batch sizes in BATCHES (default 1,8,32): HIP events
around back-to-back launches (imgfd_time_structure_tensor_batch).  The variant under test is chosen by the
environment (IMGFD_XCD_REMAP, IMGFD_TENSOR_SEG, IMGFD_TENSOR_PER_CU), read once per process.
Prints one JSON line per batch size."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from image_amd import _lib
if os.environ.get("VARIANT_LIB"):   # experiment builds of the library (scripts/variants/*.so, not tracked)
    _lib.LIB_PATH = os.path.abspath(os.environ["VARIANT_LIB"])
from image_amd.device import DeviceDetector

NX, NY = 3840, 2160
det = DeviceDetector(0)
det.ctx.set_fir_mode(int(os.environ.get("FIR_MODE", "1")))
batches = [int(b) for b in os.environ.get("BATCHES", "1,8,32").split(",")]
bmax = max(batches)
frames = det.synth_frames(bmax, NX, NY, seed0=50000)
ix = torch.empty((bmax, NY, NX), dtype=torch.float32, device="cuda")
iy = torch.empty_like(ix)
for f in range(bmax):
    det.gradients_of(frames[f], ix[f], iy[f])
tag = {k: os.environ[k] for k in ("VARIANT_LIB", "IMGFD_XCD_REMAP", "IMGFD_TENSOR_SEG", "IMGFD_TENSOR_PER_CU", "IMGFD_TENSOR_WAVE", "FIR_MODE") if k in os.environ}
for b in batches:
    us = det.time_structure_tensor_batch(ix[:b], iy[:b], warmup=int(os.environ.get("WARMUP", "3")), iters=int(os.environ.get("ITERS", "30")))
    gbs = 20 * NX * NY * b / (us * 1e-6) / 1e9
    print(json.dumps({"kernel": "structure_tensor", "variant": tag or "default", "batch": b, "us_per_launch": round(us, 2),
                      "us_per_frame": round(us / b, 2), "algorithmic_GBps": round(gbs, 1), "frac_of_8TBps": round(gbs / 8000, 4)}), flush=True)
