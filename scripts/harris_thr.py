"""imgfd_harris_dev on 32 4K frames at a given threshold (argv[1]): separates the fixed cost of harris_nms_sparse (threshold
1e30: no candidate) from its per-candidate cost.  Run under rocprofv3 --kernel-trace --stats."""
import sys

import torch

from image_amd.device import DeviceDetector

thr = float(sys.argv[1]) if len(sys.argv) > 1 else 130.0
det = DeviceDetector(0)
frames = det.synth_frames(32, 3840, 2160, seed0=2)
out = det.harris(frames, threshold=thr)
for _ in range(3):
    det.harris(frames, threshold=thr, out=out)
torch.cuda.synchronize()
print("threshold", thr, "corners per frame", float(out[1].float().mean()))
