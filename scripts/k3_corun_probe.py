"""Does a small-register kernel run on CUs that a structure-tensor workgroup occupies?  Stream A: imgfd_harris_dev on 32 4K
frames (K3: one 12-wave workgroup per CU, 156 registers per lane); stream B: a chain of tiny elementwise kernels (torch, a few
registers, no LDS).  Run under rocprofv3 --kernel-trace and look at where the stream-B kernels land."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from image_amd.device import DeviceDetector
det = DeviceDetector(0)
frames = det.synth_frames(32, 3840, 2160, seed0=2)
out = det.harris(frames)
x = torch.zeros(1 << 22, device="cuda", dtype=torch.int32)   # 4 M elements: 16 K workgroups of 256
sb = torch.cuda.Stream()
torch.cuda.synchronize()
for rep in range(3):
    det.harris(frames, out=out)
    with torch.cuda.stream(sb):
        for _ in range(60): x.add_(1)
    torch.cuda.synchronize()
print("done", int(x[0]))
