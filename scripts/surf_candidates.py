import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from image_amd.device import DeviceDetector
det = DeviceDetector(0)
S = 4096
tiles = torch.empty((1, S, S, 3), dtype=torch.uint8, device="cuda")
tiles[0] = det.synth_frames(3, S, S, seed0=9).permute(1, 2, 0)
feat = torch.zeros((1, 400000, 70), dtype=torch.float64, device="cuda")
counts = torch.zeros((1,), dtype=torch.int64, device="cuda")
det.surf(tiles, feat, counts, max_points=10**7, threshold=30.0)
torch.cuda.synchronize()
print("candidates kept by the box test:", int(counts[0]))
