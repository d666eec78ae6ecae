"""Time imgfd_surf_dev on 16 bench tiles (4096^2) resident in HBM (VARIANT_LIB = an alternative build of the library)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_amd import _lib
if os.environ.get("VARIANT_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["VARIANT_LIB"])
import torch
from image_amd.device import DeviceDetector
T, S = (1 if os.environ.get("TILES1") else int(os.environ.get("TILES", "16"))), 4096
det = DeviceDetector(0)
tiles = torch.empty((T, S, S, 3), dtype=torch.uint8, device="cuda")
for t in range(T):
    tiles[t] = det.synth_frames(3, S, S, seed0=3 * (3 + t)).permute(1, 2, 0)
feat = torch.zeros((T, 1000, 70), dtype=torch.float64, device="cuda")
counts = torch.zeros((T,), dtype=torch.int64, device="cuda")
for _ in range(2): det.surf(tiles, feat, counts, max_points=1000, threshold=30.0)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(4): det.surf(tiles, feat, counts, max_points=1000, threshold=30.0)
e1.record(); e1.synchronize()
print(json.dumps({"surf_ms_per_tile": round(e0.elapsed_time(e1) / 4 / T, 4), "points": int(counts.sum()), "variant": os.environ.get("VARIANT_LIB", "default")}))
