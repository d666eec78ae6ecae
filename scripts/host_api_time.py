"""PCIe-inclusive rate of the host-pointer (drop-in) entry points on one 3840x2160 frame: what an R caller sees."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from image_amd import _lib, api, synth
NX, NY = 3840, 2160
ctx = _lib.Context(0)
img = synth.frame(2, NX, NY)
x_num = img.T.astype(np.float64)       # what image_harris() receives: W x H numeric matrix
x_int = img.T.astype(np.int32)         # as.integer(x) for F9 / Canny
def best(fn, reps=5):
    fn(); ts = []
    for _ in range(reps):
        t = time.perf_counter(); fn(); ts.append(time.perf_counter() - t)
    return min(ts)
t_h = best(lambda: api.image_harris(x_num, ctx=ctx))
t_f = best(lambda: api.image_detect_corners(x_int, threshold=20, suppress_non_max=True, ctx=ctx))
t_c = best(lambda: api.image_canny_edge_detector(x_int, ctx=ctx))
# the C ABI alone on marshalled vectors (what the .Call glue does after INTEGER(x) / REAL(x)): PCIe + device + PCIe
import ctypes as C
from image_amd import _binding
lib, h = ctx.lib, ctx.handle
d64 = np.ascontiguousarray(img.astype(np.float64)); i32 = np.ascontiguousarray(img.astype(np.int32))
edges = np.zeros((NY, NX), np.uint8); nz = C.c_int64(0)
def c_harris():
    out = _binding.Corners()
    ctx.check(lib.imgfd_harris_f64(h, d64.ctypes.data_as(C.c_void_p), NX, NY, 0.06, 1.0, 2.5, 130.0, 0, 0, 0, 1, 0, 1, 0, 10, 0, C.byref(out)), "h")
    if out.n: lib.imgfd_free(out.corners)
def c_fast9():
    out = _binding.Points()
    ctx.check(lib.imgfd_fast9_i32(h, i32.ctypes.data_as(C.c_void_p), NX, NY, NX, 20, 1, C.byref(out)), "f")
    if out.n: lib.imgfd_free(out.points)
def c_canny():
    ctx.check(lib.imgfd_canny_i32(h, i32.ctypes.data_as(C.c_void_p), NX, NY, 2.0, 3.0, 10.0, 1, edges.ctypes.data_as(C.c_void_p), C.byref(nz)), "c")
c_h, c_f, c_c = best(c_harris), best(c_fast9), best(c_canny)
px = NX * NY
print(json.dumps({"c_abi_only": True, "imgfd_harris_f64_ms": round(1e3 * c_h, 2), "imgfd_fast9_i32_ms": round(1e3 * c_f, 2),
                  "imgfd_canny_i32_ms": round(1e3 * c_c, 2), "sum_Mpix_s": round(px / (c_h + c_f + c_c) / 1e6, 1),
                  "note": "pageable host vectors: 8 B/px (Harris doubles) and 4 B/px (ints) up, corner lists / 1 B/px edge map down"}))
print(json.dumps({"frame": f"{NX}x{NY}", "image_harris_ms": round(1e3 * t_h, 2), "image_detect_corners_ms": round(1e3 * t_f, 2),
                  "image_canny_edge_detector_ms": round(1e3 * t_c, 2),
                  "sum_Mpix_s": round(px / (t_h + t_f + t_c) / 1e6, 1),
                  "note": "Python mirror of the R wrappers incl. host-side array marshalling, PCIe upload (8 / 4 / 4 B per pixel), download of results"}))
