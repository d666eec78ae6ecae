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
