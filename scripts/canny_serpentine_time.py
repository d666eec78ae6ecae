"""Worst case for the queued hysteresis sweeps: ONE weak edge that snakes through a whole 4K frame and is lit from its far
end only (tests/test_canny.py::_serpentine at 3840x2160).  Times imgfd_canny_dev on BATCH such frames and reports how many of
the queued sweeps still changed something (the finishing kernel, one workgroup per frame, completes the rest)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from image_amd.device import DeviceDetector
from test_canny import _serpentine, SERP_KW
B = int(os.environ.get("BATCH", 8))
img = _serpentine(3840, 2160)
det = DeviceDetector(0)
frames = torch.from_numpy(np.stack([img] * B)).cuda()
edges = torch.empty_like(frames); counts = torch.zeros(B, dtype=torch.int64, device="cuda")
kw = dict(s=SERP_KW["s"], low_thr=SERP_KW["low_thr"], high_thr=SERP_KW["high_thr"])
for _ in range(2): det.canny(frames, out=(edges, counts), **kw)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3): det.canny(frames, out=(edges, counts), **kw)
e1.record(); e1.synchronize()
print(json.dumps({"serpentine_4k_canny_ms_per_batch": round(e0.elapsed_time(e1) / 3, 3), "batch": B, "edge_pixels_per_frame": int(counts[0])}))
