"""Where the time of imgfd_surf_i32 on one 4096^2 RGB tile goes (host vectors, PCIe included): the call on INTEGER(x) (201 MB up), on
bytes (50 MB up), the detection stages alone, and plain copies of the same sizes from the same (pageable) memory."""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from image_amd import _binding, _lib, synth
ctx = _lib.Context(0); lib, h = ctx.lib, ctx.handle
S = 4096
rgb8 = np.ascontiguousarray(synth.frame_rgb(3, S, S)); rgb32 = np.ascontiguousarray(rgb8.astype(np.int32))
def best(fn, n=7):
    fn(); fn()
    ts = []
    for _ in range(n):
        t = time.perf_counter(); fn(); ts.append(time.perf_counter() - t)
    return round(1e3 * min(ts), 3)
def surf_i32():
    o = _binding.SurfOut(); ctx.check(lib.imgfd_surf_i32(h, rgb32.ctypes.data_as(C.c_void_p), S, S, 1000, 30.0, C.byref(o)), "surf_i32")
    if o.n: lib.imgfd_free(o.data)
def surf_u8():
    o = _binding.SurfOut(); ctx.check(lib.imgfd_surf(h, rgb8.ctypes.data_as(C.c_void_p), S, S, 1000, 30.0, C.byref(o)), "surf")
    if o.n: lib.imgfd_free(o.data)
pts = np.zeros((65536, 5)); n = C.c_int64()
def points_u8():
    ctx.check(lib.imgfd_surf_interest_points(h, rgb8.ctypes.data_as(C.c_void_p), S, S, 30.0, pts.ctypes.data_as(C.c_void_p), 65536, C.byref(n)), "points")
d = torch.empty(rgb32.nbytes, dtype=torch.uint8, device="cuda")
h32, h8 = torch.from_numpy(rgb32.view(np.uint8).reshape(-1)), torch.from_numpy(rgb8.reshape(-1))
def copy32(): d.copy_(h32); torch.cuda.synchronize()
def copy8(): d[:h8.numel()].copy_(h8); torch.cuda.synchronize()
out = {"imgfd_surf_i32_ms": best(surf_i32), "imgfd_surf_u8_ms": best(surf_u8), "imgfd_surf_interest_points_u8_ms": best(points_u8),
       "plain_copy_201MB_pageable_ms": best(copy32), "plain_copy_50MB_pageable_ms": best(copy8)}
print(json.dumps(out))
