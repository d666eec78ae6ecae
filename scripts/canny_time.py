"""Time imgfd_canny_dev on a batch of 4K (or WxH) frames resident in HBM; prints one JSON line.  Under
`rocprofv3 --kernel-trace --stats` the per-kernel split is in the stats file."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_amd import _lib
if os.environ.get("VARIANT_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["VARIANT_LIB"])
import torch
from image_amd.device import DeviceDetector
NX, NY, B = int(os.environ.get("NX", 3840)), int(os.environ.get("NY", 2160)), int(os.environ.get("BATCH", 32))
det = DeviceDetector(0)
frames = det.synth_frames(B, NX, NY, seed0=50000)
edges = torch.empty_like(frames); counts = torch.zeros(B, dtype=torch.int64, device="cuda")
for _ in range(3): det.canny(frames, out=(edges, counts))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
it = int(os.environ.get("ITERS", 10))
e0.record()
for _ in range(it): det.canny(frames, out=(edges, counts))
e1.record(); e1.synchronize()
ms = e0.elapsed_time(e1) / it
print(json.dumps({"canny_ms_per_batch": round(ms, 3), "batch": B, "size": [NX, NY], "Mpix_s": round(B * NX * NY / ms / 1e3, 1),
                  "edge_pixels": int(counts.sum()), "variant": os.environ.get("VARIANT_LIB", "default")}))
