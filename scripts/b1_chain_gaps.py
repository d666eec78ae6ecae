"""One 4K frame, each detector ALONE on the context's stream, 20 calls back to back: for the kernel trace (rocprofv3 --kernel-trace) that
shows the gaps between the dependent kernels of one chain."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from image_amd.device import DeviceDetector
det = DeviceDetector(0)
NX, NY = 3840, 2160
frames = det.synth_frames(1, NY, NX, seed0=50000)
out_h = (torch.zeros((1, 4096, 3), dtype=torch.float32, device="cuda"), torch.zeros((1,), dtype=torch.int64, device="cuda"))
out_f = (torch.zeros((1, 65536, 2), dtype=torch.int32, device="cuda"), torch.zeros((1,), dtype=torch.int64, device="cuda"))
out_c = (torch.zeros((1, NY, NX), dtype=torch.uint8, device="cuda"), torch.zeros((1,), dtype=torch.int64, device="cuda"))
which = os.environ.get("WHICH", "harris")
for _ in range(20):
    if which == "harris": det.harris(frames, out=out_h)
    elif which == "canny": det.canny(frames, out=out_c)
    else: det.fast9(frames, threshold=20, suppress_non_max=True, out=out_f)
torch.cuda.synchronize()
