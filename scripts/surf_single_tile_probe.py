import json, os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from image_amd.device import DeviceDetector
S = 4096
det = DeviceDetector(0)
tiles = torch.empty((1, S, S, 3), dtype=torch.uint8, device="cuda")
tiles[0] = det.synth_frames(3, S, S, seed0=9).permute(1, 2, 0)
feat = torch.zeros((1, 1000, 70), dtype=torch.float64, device="cuda"); counts = torch.zeros((1,), dtype=torch.int64, device="cuda")
for redo in (True, False):
    for _ in range(3): det.surf(tiles, feat, counts, redo=redo); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(20):
        det.surf(tiles, feat, counts, redo=redo)
        if not redo: torch.cuda.synchronize()
    torch.cuda.synchronize()
    print("redo", redo, "ms per call", round((time.perf_counter() - t) / 20 * 1e3, 4))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): det.surf(tiles, feat, counts, redo=False)
e1.record(); e1.synchronize()
print("back to back, no wait between calls: ms per call", round(e0.elapsed_time(e1) / 20, 4))
