"""ctypes doorway to the CPU oracle -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

``oracle/liboracle.so`` holds the C restatements of the reference algorithms
(harris_oracle.c, fast9_oracle.c, canny_oracle.c, ...); ``oracle/_ref/*.so``
hold the reference's own sources compiled in place (only buildable where
/root/reference exists; the built files travel to the GPU box).

Only tests/, ``__graft_entry__.smoke()`` and bench.py's ``cpu_baseline`` leg may
import this package.  ``image_amd`` never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")


def build(ref: bool = True) -> None:
    """Compile liboracle.so and, when /root/reference is present, oracle/_ref/*.so."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
    if ref and os.path.isdir("/root/reference"):
        subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build(ref=False)
        _lib = C.CDLL(path)
    return _lib


def ref_path(name: str) -> str:
    return os.path.join(_HERE, "_ref", f"libref_{name}.so")


def have_ref(name: str) -> bool:
    return os.path.exists(ref_path(name))


_refs: dict = {}


def ref(name: str) -> C.CDLL:
    if name not in _refs:
        _refs[name] = C.CDLL(ref_path(name))
    return _refs[name]


# --------------------------------------------------------------------- Harris
HARRIS_DEFAULTS = dict(k=0.06, sigma_d=1.0, sigma_i=2.5, threshold=130.0, gaussian=0, gradient=0,
                       strategy=0, Nselect=1, measure=0, Nscales=1, precision=0, cells=10)


def _harris_args(img, kw):
    p = dict(HARRIS_DEFAULTS); p.update(kw)
    img = np.ascontiguousarray(img, dtype=np.float32)
    ny, nx = img.shape
    return img, nx, ny, p


def harris(img, planes: bool = False, **kw):
    """Restated harris_scale().  img: (ny, nx) array.  Returns (xyR[n,3] float32[, planes dict])."""
    img, nx, ny, p = _harris_args(img, kw)
    fn = lib().orc_harris
    fn.restype = C.c_long
    cap = nx * ny // 4 + 16
    out = np.zeros((cap, 3), np.float32)
    pl = np.zeros((6, ny, nx), np.float32) if planes else None
    n = fn(img.ctypes.data_as(C.c_void_p), nx, ny, C.c_float(p["k"]), C.c_float(p["sigma_d"]),
           C.c_float(p["sigma_i"]), C.c_float(p["threshold"]), p["gaussian"], p["gradient"],
           p["strategy"], p["Nselect"], p["measure"], p["Nscales"], p["precision"], p["cells"],
           out.ctypes.data_as(C.c_void_p), C.c_long(cap),
           pl.ctypes.data_as(C.c_void_p) if planes else None)
    res = out[:n].copy()
    if planes:
        return res, dict(zip(("Ix", "Iy", "A", "B", "C", "R"), pl))
    return res


def ref_harris(img, threads: int | None = None, **kw):
    """The reference's own harris_scale() (oracle/_ref/libref_harris.so)."""
    img, nx, ny, p = _harris_args(img, kw)
    L = ref("harris")
    if threads is not None:
        L.ref_set_threads(int(threads))
    fn = L.ref_harris
    fn.restype = C.c_long
    cap = nx * ny // 4 + 16
    out = np.zeros((cap, 3), np.float32)
    n = fn(img.ctypes.data_as(C.c_void_p), nx, ny, C.c_float(p["k"]), C.c_float(p["sigma_d"]),
           C.c_float(p["sigma_i"]), C.c_float(p["threshold"]), p["gaussian"], p["gradient"],
           p["strategy"], p["Nselect"], p["measure"], p["Nscales"], p["precision"], p["cells"],
           out.ctypes.data_as(C.c_void_p), C.c_long(cap))
    return out[:n].copy()


def harris_stage(which: str, *arrays, use_ref: bool = False, **kw):
    """Single reference stages for plane-level parity: 'gaussian', 'gradient',
    'autocorrelation', 'response', 'nms'."""
    L = ref("harris") if use_ref else lib()
    pre = "ref_" if use_ref else "orc_"
    a = [np.ascontiguousarray(x, dtype=np.float32) for x in arrays]
    ny, nx = a[0].shape
    vp = lambda x: x.ctypes.data_as(C.c_void_p)
    if which == "gaussian":
        out = np.empty_like(a[0])
        getattr(L, pre + "gaussian")(vp(a[0]), vp(out), nx, ny, C.c_float(kw["sigma"]), kw.get("type", 0))
        return out
    if which == "gradient":
        ix, iy = np.zeros_like(a[0]), np.zeros_like(a[0])
        getattr(L, pre + "gradient")(vp(a[0]), vp(ix), vp(iy), nx, ny, kw.get("type", 0))
        return ix, iy
    if which == "autocorrelation":
        A, B, Cc = (np.empty_like(a[0]) for _ in range(3))
        getattr(L, pre + "autocorrelation")(vp(a[0]), vp(a[1]), vp(A), vp(B), vp(Cc),
                                            C.c_float(kw["sigma"]), nx, ny, kw.get("gauss", 0))
        return A, B, Cc
    if which == "response":
        R = np.empty_like(a[0])
        getattr(L, pre + "response")(vp(a[0]), vp(a[1]), vp(a[2]), vp(R), kw.get("measure", 0), nx, ny,
                                     C.c_float(kw.get("k", 0.06)))
        return R
    if which == "nms":
        # rows_independent: the OpenMP schedule in which no row sees another row's skip marks (restatement only)
        fn = lib().orc_nms_rows_independent if kw.get("rows_independent") else getattr(L, pre + "nms"); fn.restype = C.c_long
        cap = nx * ny // 4 + 16
        out = np.zeros((cap, 3), np.float32)
        n = fn(vp(a[0]), nx, ny, C.c_float(kw["Th"]), int(kw["radius"]), vp(out), C.c_long(cap))
        return out[:n].copy()
    raise ValueError(which)


# --------------------------------------------------------------------- FAST-9
def fast9(img, threshold: int, suppress_non_max: bool = False, stride: int | None = None,
          width: int | None = None):
    """Restated F9::detectCorners.  img: (h, stride) uint8.  Returns int32 (n,2) of (x,y)."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, s = img.shape
    w = s if width is None else width
    fn = lib().orc_fast9; fn.restype = C.c_long
    cap = w * h // 2 + 16
    out = np.zeros((cap, 2), np.int32)
    n = fn(img.ctypes.data_as(C.c_void_p), w, h, s if stride is None else stride,
           int(threshold) & 0xFF, int(bool(suppress_non_max)), out.ctypes.data_as(C.c_void_p),
           C.c_long(cap), None)
    return out[:n].copy()


def ref_fast9(img, threshold: int, suppress_non_max: bool = False, width: int | None = None):
    """The reference's own C API f9_detect_corners (f9.h:77-86)."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, s = img.shape
    w = s if width is None else width
    L = ref("f9")
    L.f9_alloc.restype = C.c_void_p
    L.f9_detect_corners.restype = C.c_void_p
    ctx = C.c_void_p(L.f9_alloc())
    n = C.c_int(0)
    p = L.f9_detect_corners(ctx, img.ctypes.data_as(C.c_void_p), w, h, s, C.c_ubyte(int(threshold) & 0xFF),
                            C.c_bool(bool(suppress_non_max)), C.byref(n))
    if n.value:
        out = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_int)), shape=(n.value, 2)).copy()
    else:
        out = np.zeros((0, 2), np.int32)
    L.f9_dealloc(ctx)
    return out.astype(np.int32)


# ---------------------------------------------------------------------- Canny
def canny(img, s: float = 2.0, low_thr: float = 3.0, high_thr: float = 10.0, accGrad: bool = True,
          debug: bool = False):
    """Restated canny_edge_detector().  img: (ny, nx) uint8 with C index x + nx*y.
    Returns (edges uint8 (ny,nx) of 0/255, pixels_nonzero[, dict(blur, grad, nms)])."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    ny, nx = img.shape
    edges = np.zeros((ny, nx), np.uint8)
    fn = lib().orc_canny; fn.restype = C.c_long
    vp = lambda x: x.ctypes.data_as(C.c_void_p)
    if debug:
        blur = np.zeros((ny, nx)); grad = np.zeros((ny, nx)); nms = np.zeros((ny, nx), np.uint8)
        n = fn(vp(img), nx, ny, C.c_double(s), C.c_double(low_thr), C.c_double(high_thr),
               int(bool(accGrad)), vp(edges), vp(blur), vp(grad), vp(nms))
        return edges, int(n), dict(blur=blur, grad=grad, nms=nms)
    n = fn(vp(img), nx, ny, C.c_double(s), C.c_double(low_thr), C.c_double(high_thr),
           int(bool(accGrad)), vp(edges), None, None, None)
    return edges, int(n)


def fft_gblur(img, s: float = 2.0):
    """gblur() as the reference computes it (tools.c:146-185): y = float(ifft2(fft2(x) * fft2(g)) / (w h)) with the wrapped
    Gaussian -- three 2-D FFTs, here through numpy's pocketfft (the reference calls FFTW3, which this image does not have)."""
    img = np.asarray(img)
    h, w = img.shape
    xs = np.where(np.arange(w) < w // 2, np.arange(w), np.arange(w) - w).astype(np.float64)
    ys = np.where(np.arange(h) < h // 2, np.arange(h), np.arange(h) - h).astype(np.float64)
    g = np.exp(-(xs[None, :] ** 2 + ys[:, None] ** 2) / (s * s))
    g /= g.sum()
    y = np.fft.ifft2(np.fft.fft2(img.astype(np.float64)) * np.fft.fft2(g))
    return y.real.astype(np.float32)


def canny_fft(img, s: float = 2.0, low_thr: float = 3.0, high_thr: float = 10.0, accGrad: bool = True):
    """canny_edge_detector() with the reference's ALGORITHM for the blur (FFT product) instead of the restatement's direct
    circular sums: the CPU leg a timing baseline should quote (bench.py).  Same edge map (tests/test_oracle.py)."""
    return canny_from_blur(fft_gblur(img, s), low_thr, high_thr, accGrad)


def canny_from_blur(blur, low_thr: float = 3.0, high_thr: float = 10.0, accGrad: bool = True):
    """The stages behind the blur (gradient, maxima, hysteresis: rcpp_canny.cpp:153-215) on a given blurred plane (ny, nx) of
    float values -- lets a test swap the blur's FFT for another implementation (FFTW3 itself is not in this image).
    Returns (edges uint8 (ny,nx) of 0/255, pixels_nonzero)."""
    data = np.ascontiguousarray(np.asarray(blur, dtype=np.float32), dtype=np.float64)   # data[] holds floats (crealf, tools.c:126)
    ny, nx = data.shape
    L = lib()
    vp = lambda x: x.ctypes.data_as(C.c_void_p)
    grad, theta, edges = np.zeros((ny, nx)), np.zeros((ny, nx)), np.zeros((ny, nx), np.uint8)
    L.orc_canny_gradient(vp(data), vp(grad), vp(theta), nx, ny, int(bool(accGrad)))
    L.orc_canny_maxima(vp(grad), vp(theta), vp(edges), nx, ny, int(low_thr), int(high_thr))
    L.orc_canny_hysteresis.restype = C.c_long
    n = L.orc_canny_hysteresis(vp(edges), nx, ny)
    return edges, int(n)


def ref_canny(img, s: float = 2.0, low_thr: float = 3.0, high_thr: float = 10.0, accGrad: bool = True):
    """The reference's own canny_edge_detector() (rcpp_canny.cpp:122-244 + tools.c + adsf.c compiled in place into
    oracle/_ref/libref_canny.so; FFTW3 replaced by the plain DFT of oracle/fftw_stub.c).  Slow: O(n^3) transforms.
    Returns (edges uint8 (ny,nx) of 0/255, pixels_nonzero)."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    ny, nx = img.shape
    as_int = np.ascontiguousarray(img, dtype=np.int32)         # as.integer(x), canny_edges_detector.R
    edges = np.zeros((ny, nx), np.uint8)
    fn = ref("canny").ref_canny
    fn.restype = C.c_long
    n = fn(as_int.ctypes.data_as(C.c_void_p), nx, ny, C.c_double(s), C.c_double(low_thr), C.c_double(high_thr),
           int(bool(accGrad)), edges.ctypes.data_as(C.c_void_p))
    return edges, int(n)


# ----------------------------------------------------------------------- fHOG
def _rgb(img):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    assert img.ndim == 3 and img.shape[2] == 3, "rgb image (rows, cols, 3)"
    return img


def fhog(rgb, cell_size: int = 8, pad_r: int = 1, pad_c: int = 1, debug: bool = False):
    """Restated extract_fhog_features (oracle/fhog_oracle.c).  rgb: (rows, cols, 3) uint8.
    Returns hog (hog_nr, hog_nc, 31) float32 in dlib's array2d order [, dict(hist, norm)]."""
    rgb = _rgb(rgb)
    rows, cols, _ = rgb.shape
    nr, nc = C.c_int(), C.c_int()
    if lib().orc_fhog_dims(rows, cols, cell_size, pad_r, pad_c, C.byref(nr), C.byref(nc)):
        raise ValueError("fhog oracle: unsupported arguments (cell_size < 2?)")
    out = np.zeros((nr.value, nc.value, 31), np.float32)
    vp = lambda x: x.ctypes.data_as(C.c_void_p)
    cells_nr = int(np.float32(rows) / np.float32(cell_size) + 0.5)
    cells_nc = int(np.float32(cols) / np.float32(cell_size) + 0.5)
    hist = np.zeros((cells_nr + 2, cells_nc + 2, 18), np.float32) if debug else None
    norm = np.zeros((cells_nr, cells_nc), np.float32) if debug else None
    if out.size:
        lib().orc_fhog(vp(rgb), rows, cols, cell_size, pad_r, pad_c, vp(out), vp(hist) if debug else None,
                       vp(norm) if debug else None)
    return (out, dict(hist=hist, norm=norm)) if debug else out


def ref_fhog(rgb, cell_size: int = 8, pad_r: int = 1, pad_c: int = 1):
    """dlib's own extract_fhog_features (oracle/_ref/libref_dlib.so), array2d order (hog_nr, hog_nc, 31)."""
    rgb = _rgb(rgb)
    rows, cols, _ = rgb.shape
    L = ref("dlib")
    nr, nc = C.c_int(), C.c_int()
    L.ref_fhog(rgb.ctypes.data_as(C.c_void_p), rows, cols, cell_size, pad_r, pad_c, None, C.byref(nr), C.byref(nc))
    out = np.zeros((nr.value, nc.value, 31), np.float32)
    if out.size:
        L.ref_fhog(rgb.ctypes.data_as(C.c_void_p), rows, cols, cell_size, pad_r, pad_c,
                   out.ctypes.data_as(C.c_void_p), C.byref(nr), C.byref(nc))
    return out


# ----------------------------------------------------------------------- SURF
def _surf_call(L, name, rgb, *args, rec, cap=400000):
    rgb = _rgb(rgb)
    rows, cols, _ = rgb.shape
    fn = getattr(L, name); fn.restype = C.c_long
    out = np.zeros((cap, rec), np.float64)
    n = fn(rgb.ctypes.data_as(C.c_void_p), rows, cols, *args, out.ctypes.data_as(C.c_void_p), C.c_long(cap))
    assert n <= cap, "raise cap"
    return out[:n].copy()


def surf_integral(rgb, use_ref: bool = False):
    rgb = _rgb(rgb)
    rows, cols, _ = rgb.shape
    out = np.zeros((rows, cols), np.int32)
    (ref("dlib").ref_integral if use_ref else lib().orc_surf_integral)(rgb.ctypes.data_as(C.c_void_p), rows, cols,
                                                                      out.ctypes.data_as(C.c_void_p))
    return out


def surf_interest_points(rgb, threshold: float = 30.0, use_ref: bool = False):
    """(n, 5): x, y, scale, score, laplacian in get_interest_points' emission order."""
    L, name = (ref("dlib"), "ref_surf_interest_points") if use_ref else (lib(), "orc_surf_interest_points")
    return _surf_call(L, name, rgb, C.c_double(threshold), rec=6)[:, :5]


def surf(rgb, max_points: int = 1000, threshold: float = 30.0, use_ref: bool = False):
    """dict of x, y, angle, pyramid_scale, score, laplacian (n,) and surf (n, 64), like rcpp_surf.cpp:45-52."""
    L, name = (ref("dlib"), "ref_surf") if use_ref else (lib(), "orc_surf")
    r = _surf_call(L, name, rgb, C.c_long(max_points), C.c_double(threshold), rec=71, cap=max(16, min(int(max_points), 400000)))
    return dict(x=r[:, 0], y=r[:, 1], angle=r[:, 2], pyramid_scale=r[:, 3], score=r[:, 4], laplacian=r[:, 5], surf=r[:, 7:])


# ----------------------------------------------------------------------- kNN (descriptor matching)
def knn(data, query, k: int = 1):
    """Exact k nearest rows of `data` for every row of `query` (Euclidean).  Returns (index int32 (nq,k) 0-based,
    dist float64 (nq,k)); restates FNN::get.knnx as image.dlib/README.md:19-37 uses it (parity unpinned: FNN absent)."""
    data = np.ascontiguousarray(data, np.float64); query = np.ascontiguousarray(query, np.float64)
    nd, dim = data.shape
    nq = query.shape[0]
    assert query.shape[1] == dim
    idx = np.zeros((nq, k), np.int32); dist = np.zeros((nq, k), np.float64)
    vp = lambda x: x.ctypes.data_as(C.c_void_p)
    lib().orc_knn(vp(data), C.c_long(nd), vp(query), C.c_long(nq), dim, k, vp(idx), vp(dist))
    return idx, dist
