"""Time imgfd_fast9_dev on a batch of 4K frames resident in HBM (VARIANT_LIB = an alternative build of the library)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_amd import _lib
if os.environ.get("VARIANT_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["VARIANT_LIB"])
import torch
from image_amd.device import DeviceDetector
NX, NY, B = int(os.environ.get("NX", 3840)), int(os.environ.get("NY", 2160)), int(os.environ.get("BATCH", 32))
det = DeviceDetector(0)
frames = det.synth_frames(B, NX, NY, seed0=2)
KW = dict(threshold=20, suppress_non_max=True)   # the bench's parameters
out = det.fast9(frames, **KW)
for _ in range(3): det.fast9(frames, out=out, **KW)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
it = int(os.environ.get("ITERS", 20))
e0.record()
for _ in range(it): det.fast9(frames, out=out, **KW)
e1.record(); e1.synchronize()
ms = e0.elapsed_time(e1) / it
print(json.dumps({"fast9_ms_per_batch": round(ms, 4), "batch": B, "points": int(out[1].sum()), "variant": os.environ.get("VARIANT_LIB", "default")}))
