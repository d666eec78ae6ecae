"""SURF batch path on a few 4096^2 bench tiles at a given threshold (argv[1]); prints the counts.  Used under rocprofv3 to
separate the fixed cost of the maxima kernel (threshold 1e30: empty masks) from the per-candidate cost."""
import sys

import torch

from image_amd.device import DeviceDetector

thr = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
T, S = 4, 4096
det = DeviceDetector(0)
tiles = torch.empty((T, S, S, 3), dtype=torch.uint8, device="cuda")
for t in range(T):
    tiles[t] = det.synth_frames(3, S, S, seed0=3 * (3 + t)).permute(1, 2, 0)
feat = torch.zeros((T, 1000, 70), dtype=torch.float64, device="cuda")
counts = torch.zeros((T,), dtype=torch.int64, device="cuda")
for _ in range(3):
    det.surf(tiles, feat, counts, max_points=1000, threshold=thr)
torch.cuda.synchronize()
print("threshold", thr, "counts", counts.tolist())
