"""Time imgfd_harris_dev on 32 4K frames resident in HBM (VARIANT_LIB = an alternative build of the library)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_amd import _lib
if os.environ.get("VARIANT_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["VARIANT_LIB"])
import torch
from image_amd.device import DeviceDetector
det = DeviceDetector(0)
frames = det.synth_frames(32, 3840, 2160, seed0=2)
out = det.harris(frames)
for _ in range(3): det.harris(frames, out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
it = int(os.environ.get("ITERS", 10))
e0.record()
for _ in range(it): det.harris(frames, out=out)
e1.record(); e1.synchronize()
print(json.dumps({"harris_ms_per_batch": round(e0.elapsed_time(e1) / it, 4), "corners": int(out[1].sum()), "variant": os.environ.get("VARIANT_LIB", "default")}))
