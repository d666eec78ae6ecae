"""Run imgfd_surf (host-pointer API) on one SIZE x SIZE RGB tile (BASELINE config 4) and print wall times."""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from image_amd import _lib, _binding, synth
S = int(os.environ.get("SIZE", 4096))
ctx = _lib.Context(0)
rng = np.random.default_rng(1)
rgb = synth.frame_rgb(3, S, S).astype(np.float64) * 0.4
yy, xx = np.mgrid[0:S, 0:S]
for _ in range(400):   # blobs so that the Hessian pyramid has maxima
    cx, cy, s = rng.uniform(0, S), rng.uniform(0, S), rng.uniform(3, 20)
    x0, x1, y0, y1 = int(max(0, cx - 4 * s)), int(min(S, cx + 4 * s)), int(max(0, cy - 4 * s)), int(min(S, cy + 4 * s))
    rgb[y0:y1, x0:x1] += rng.uniform(-100, 140) * np.exp(-((xx[y0:y1, x0:x1] - cx) ** 2 + (yy[y0:y1, x0:x1] - cy) ** 2) / (2 * s * s))[..., None]
rgb = np.ascontiguousarray(np.clip(rgb, 0, 255).astype(np.uint8))
out = _binding.SurfOut()
ts = []
for it in range(3):
    t = time.perf_counter()
    ctx.check(ctx.lib.imgfd_surf(ctx.handle, rgb.ctypes.data_as(C.c_void_p), S, S, 1000, 30.0, C.byref(out)), "imgfd_surf")
    ts.append(time.perf_counter() - t)
    n = out.n
    if n: ctx.lib.imgfd_free(out.data)
pts = np.zeros((400000, 5)); cnt = C.c_int64(0)
t = time.perf_counter()
ctx.check(ctx.lib.imgfd_surf_interest_points(ctx.handle, rgb.ctypes.data_as(C.c_void_p), S, S, 30.0, pts.ctypes.data_as(C.c_void_p), 400000, C.byref(cnt)), "ip")
t_ip = time.perf_counter() - t
print(json.dumps({"imgfd_surf_ms": [round(1e3 * x, 2) for x in ts], "points": int(n), "interest_points": int(cnt.value),
                  "interest_points_ms": round(1e3 * t_ip, 2), "size": S}))
# batch path with K19 on the device: 4 tiles already in HBM -> features in HBM
import torch
NT = 4
d_rgb = torch.from_numpy(rgb).to("cuda:0").unsqueeze(0).repeat(NT, 1, 1, 1).contiguous()
feat = torch.zeros((NT, 1000, 70), dtype=torch.float64, device="cuda:0"); cnt_d = torch.zeros(NT, dtype=torch.int64, device="cuda:0")
torch.cuda.synchronize()
ts = []
for it in range(3):
    t = time.perf_counter()
    ctx.check(ctx.lib.imgfd_surf_dev(ctx.handle, C.c_void_p(d_rgb.data_ptr()), NT, S, S, 3 * S * S, 1000, 30.0,
                                     C.c_void_p(feat.data_ptr()), 1000, C.c_void_p(cnt_d.data_ptr())), "imgfd_surf_dev")
    ctx.sync(); ts.append((time.perf_counter() - t) / NT)
print(json.dumps({"imgfd_surf_dev_ms_per_tile": [round(1e3 * x, 2) for x in ts], "features": cnt_d.cpu().tolist()}))
# descriptor matching: n x n unit vectors of 64 doubles, k = 2, device-resident
for n_pts in (1000, 10000):
    g = torch.Generator(device="cpu").manual_seed(1)
    a = torch.nn.functional.normalize(torch.randn((n_pts, 64), dtype=torch.float64, generator=g), dim=1).to("cuda:0")
    b = torch.nn.functional.normalize(torch.randn((n_pts, 64), dtype=torch.float64, generator=g), dim=1).to("cuda:0")
    idx = torch.zeros((n_pts, 2), dtype=torch.int32, device="cuda:0"); dist = torch.zeros((n_pts, 2), dtype=torch.float64, device="cuda:0")
    torch.cuda.synchronize()
    ts = []
    for it in range(3):
        t = time.perf_counter()
        ctx.check(ctx.lib.imgfd_knn_dev(ctx.handle, C.c_void_p(a.data_ptr()), n_pts, 64, 1, C.c_void_p(b.data_ptr()), n_pts, 64, 1, 64, 2,
                                        C.c_void_p(idx.data_ptr()), C.c_void_p(dist.data_ptr())), "knn")
        ctx.sync(); ts.append(time.perf_counter() - t)
    ref = torch.cdist(b, a).topk(2, dim=1, largest=False)
    print(json.dumps({"knn_n": n_pts, "knn_ms": [round(1e3 * x, 3) for x in ts], "index_agrees_with_torch_cdist": bool((ref.indices.int() == idx).all()),
                      "Gpair_s": round(n_pts * n_pts / min(ts) / 1e9, 2)}))
