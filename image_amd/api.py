"""Python mirror of the reference's R-level functions for the hot path.

R is not available in the build image, so the host side above the C ABI is written in Python with the
same names, arguments, defaults, argument handling and return structure as the R wrappers:

  image_harris()               image.CornerDetectionHarris/R/pkg.R:56-106
  detect_corners()             image.CornerDetectionHarris/R/RcppExports.R:4-6 (numeric enum codes)
  image_detect_corners()       image.CornerDetectionF9/R/image_detect_corners.R:48-60
  image_canny_edge_detector()  image.CannyEdges/R/canny_edges_detector.R:63-67
  image_fhog()                 image.dlib/R/image_fhog.R:35-48
  image_surf()                 image.dlib/R/image_surf.R:83-90

An R matrix ``x`` is mirrored by a 2-D numpy array with the same ``[row, col]`` indexing.  R hands the
matrix's column-major memory to C, so the C image (index ``i + nrow*j``) is ``x.T`` in numpy terms.
Every function calls libimgfd.so through ctypes; nothing here computes on the CPU.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _binding, _lib

_GAUSSIAN = ("fast Gaussian", "precise Gaussian", "no Gaussian")
_GRADIENT = ("central differences", "Sobel operator")
_STRATEGY = ("all corners", "sort all corners", "N corners", "distributed N corners")
_MEASURE = ("Harris", "Shi-Tomasi", "Harmonic Mean")
_PRECISION = ("quadratic approximation", "quartic interpolation", "no subpixel")


class RList(dict):
    """An R list with a class attribute: dict access plus ``.r_class``."""

    r_class: str = ""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:  # pragma: no cover
            raise AttributeError(k) from e


def _match_arg(arg, choices, name):
    """R's match.arg(arg) for a formal whose default is the choices vector."""
    if isinstance(arg, (tuple, list)):
        if tuple(arg) == tuple(choices):
            return choices[0]
        if len(arg) != 1:
            raise ValueError(f"'{name}' must be of length 1")
        arg = arg[0]
    hits = [c for c in choices if c.startswith(arg)]
    if arg in choices:
        hits = [arg]
    if len(hits) != 1:
        raise ValueError(f"'{name}' should be one of " + ", ".join(f"'{c}'" for c in choices))
    return hits[0]


def _r_option_code(arg, choices, name):
    """``which(arg %in% match.arg(arg)) - 1L`` exactly as pkg.R:70-74 evaluates it.

    For the untouched default (the whole choices vector) and for any exact single string this is 0;
    a partially matched string gives integer(0), which Rcpp rejects."""
    matched = _match_arg(arg, choices, name)
    vec = list(arg) if isinstance(arg, (tuple, list)) else [arg]
    which = [i + 1 for i, v in enumerate(vec) if v == matched]
    if len(which) != 1:
        raise ValueError("Expecting a single value: [extent=%d]." % len(which))
    return which[0] - 1


def _ctx(ctx):
    return ctx if ctx is not None else _lib.default_context()


def detect_corners(x, nx, ny, k=0.060000, sigma_d=1.000000, sigma_i=2.500000, threshold=130, gaussian=1,
                   gradient=0, strategy=0, Nselect=1, measure=0, Nscales=1, precision=1, cells=10, verbose=1,
                   ctx=None):
    """Rcpp-level detect_corners(): rcpp_harris.cpp:19-60 (numeric enum codes, Rcpp defaults).

    ``x`` is the NumericVector: any array with nx*ny elements in C-image order (x fastest)."""
    ctx = _ctx(ctx)
    # the NumericVector as R holds it (REAL(x)); the (float) x[i] narrowing of rcpp_harris.cpp:34-35 happens on the device
    img = np.ascontiguousarray(np.asarray(x, dtype=np.float64).ravel(order="K"))
    if img.size != int(nx) * int(ny):
        raise ValueError("x must hold nx*ny values")
    out = _binding.Corners()
    st = ctx.lib.imgfd_harris_f64(ctx.handle, img.ctypes.data_as(C.c_void_p), int(nx), int(ny), float(k),
                              float(sigma_d), float(sigma_i), float(threshold), int(gaussian), int(gradient),
                              int(strategy), int(Nselect), int(measure), int(Nscales), int(precision),
                              int(cells), int(bool(verbose)), C.byref(out))
    ctx.check(st, "imgfd_harris_f64")
    n = out.n
    if n:
        arr = np.ctypeslib.as_array(C.cast(out.corners, C.POINTER(C.c_float)), shape=(n, 3)).copy()
        ctx.lib.imgfd_free(out.corners)
    else:
        arr = np.zeros((0, 3), np.float32)
    if verbose:
        names = (" 1.Smoothing the image: \t \t", " 2.Computing the gradient: \t \t",
                 " 3.Computing the autocorrelation: \t", " 4.Computing corner strength function: \t",
                 " 5.Non-maximum suppression:  \t\t", " 6.Selecting output corners:  \t\t",
                 " 7.Calculating subpixel accuracy: \t")
        print("\nHarris corner detection:")
        print("[nx=%d, ny=%d, sigma_i=%f]" % (nx, ny, sigma_i))
        for i, nm in enumerate(names):
            if i == 6 and precision not in (1, 2):
                continue
            print("%sTime: %fs" % (nm, out.stage_seconds[i]))
        print(" * Number of corners detected: %d" % n)
    res = RList(x=arr[:, 0].astype(np.float64), y=arr[:, 1].astype(np.float64),
                strength=arr[:, 2].astype(np.float64))
    return res


def image_harris(x, k=0.060000, sigma_d=1.000000, sigma_i=2.500000, threshold=130,
                 gaussian=_GAUSSIAN, gradient=_GRADIENT, strategy=_STRATEGY, Nselect=1, measure=_MEASURE,
                 Nscales=1, precision=_PRECISION, cells=10, verbose=False, ctx=None):
    """image_harris(): pkg.R:56-106.  ``x``: matrix (W x H, like the R call) of grey values."""
    gaussian = _r_option_code(gaussian, _GAUSSIAN, "gaussian")
    gradient = _r_option_code(gradient, _GRADIENT, "gradient")
    strategy = _r_option_code(strategy, _STRATEGY, "strategy")
    measure = _r_option_code(measure, _MEASURE, "measure")
    precision = _r_option_code(precision, _PRECISION, "precision")
    x = np.asarray(x)
    if x.ndim != 2:
        raise ValueError("x is not a matrix nor a magick-image")
    w, h = x.shape  # w <- nrow(x); h <- ncol(x), pkg.R:94-95
    corners = detect_corners(x.T, w, h, k=k, sigma_d=sigma_d, sigma_i=sigma_i, threshold=threshold,
                             gaussian=gaussian, gradient=gradient, strategy=strategy, Nselect=Nselect,
                             measure=measure, Nscales=Nscales, precision=precision, cells=cells,
                             verbose=verbose, ctx=ctx)
    corners.r_class = "image.harris"
    return corners


def _as_integer(x):
    """R's as.integer() on a numeric matrix: truncation toward zero."""
    x = np.asarray(x)
    if np.issubdtype(x.dtype, np.integer):
        return x.astype(np.int64)
    return np.trunc(x).astype(np.int64)


def image_detect_corners(x, threshold=50, suppress_non_max=False, ctx=None):
    """image_detect_corners(): image_detect_corners.R:48-60 over f9_rcpp.cpp:8-35."""
    x = np.asarray(x)
    if x.ndim != 2:
        raise ValueError("is.matrix(x) is not TRUE")
    ctx = _ctx(ctx)
    width, height = x.shape  # width = nrow(x), height = ncol(x), bytes_per_row = nrow(x)
    # as.integer(x): the IntegerVector as R holds it (column-major == x.T raster); (unsigned char) x[i] of
    # f9_rcpp.cpp:11 happens on the device
    img = np.ascontiguousarray(_as_integer(x).T.astype(np.int32))
    out = _binding.Points()
    st = ctx.lib.imgfd_fast9_i32(ctx.handle, img.ctypes.data_as(C.c_void_p), int(width), int(height), int(width),
                                 int(_as_integer(threshold)) & 0xFF, int(bool(suppress_non_max)), C.byref(out))
    ctx.check(st, "imgfd_fast9_i32")
    n = out.n
    if n:
        pts = np.ctypeslib.as_array(C.cast(out.points, C.POINTER(C.c_int)), shape=(n, 2)).copy()
        ctx.lib.imgfd_free(out.points)
    else:
        pts = np.zeros((0, 2), np.int32)
    # corners_x = out.y ; corners_y = width - out.x, f9_rcpp.cpp:29-30
    res = RList(x=pts[:, 1].astype(np.float64), y=(width - pts[:, 0]).astype(np.float64))
    res.r_class = "image.corners"
    return res


def image_canny_edge_detector(x, s=2, low_thr=3, high_thr=10, accGrad=True, ctx=None):
    """image_canny_edge_detector(): canny_edges_detector.R:63-67 over rcpp_canny.cpp:122-244."""
    x = np.asarray(x)
    if x.ndim != 2:
        raise ValueError("x must be a matrix")
    ctx = _ctx(ctx)
    nx, ny = x.shape  # X = nrow(x), Y = ncol(x)
    img = np.ascontiguousarray(_as_integer(x).T.astype(np.int32))  # as.integer(x); rcpp_canny.cpp:135-136 on the device
    # NumericMatrix(nx, ny), rcpp_canny.cpp:226-233: allocated here (the glue: Rf_allocMatrix), filled by the library -- the
    # edge map is widened to doubles on the device (the R matrix is column-major: element (x, y) at x + nx*y)
    edges = np.zeros((ny, nx), np.float64)
    nonzero = C.c_int64(0)
    st = ctx.lib.imgfd_canny_f64out(ctx.handle, img.ctypes.data_as(C.c_void_p), int(nx), int(ny), float(s), float(low_thr),
                                float(high_thr), int(bool(accGrad)), edges.ctypes.data_as(C.c_void_p), C.byref(nonzero))
    ctx.check(st, "imgfd_canny_f64out")
    res = RList(edges=edges.T,
                pixels_nonzero=int(nonzero.value), nx=float(nx), ny=float(ny), s=float(s),
                low_thr=float(low_thr), high_thr=float(high_thr), accGrad=bool(accGrad))
    res.r_class = "image_canny"
    return res


def _rgb_bytes(x):
    """The glue's std::vector<int> -> rgb_pixel narrowing (rcpp_fhog.cpp:17-24) of an R integer array with
    dim (3, width, height): element [ch, c, r] sits at ch + 3*c + 3*cols*r, i.e. interlaced RGB rows."""
    x = np.asarray(x)
    if x.ndim != 3 or x.shape[0] != 3:
        raise ValueError("x must be a 3-dimensional array: RGB x width x height")
    _, width, height = x.shape
    # R's column-major (3, W, H) array read as a flat IntegerVector == C-order (H, W, 3); the narrowing to bytes
    # (rgb_pixel(unsigned char, ..)) happens on the device
    rgb = np.ascontiguousarray(_as_integer(x).astype(np.int32).transpose(2, 1, 0))  # (height, width, 3) int32
    return rgb, width, height


def image_fhog(x, cell_size=8, filter_rows_padding=1, filter_cols_padding=1, ctx=None):
    """image_fhog(): image_fhog.R:35-48 over dlib_fhog(), rcpp_fhog.cpp:10-46."""
    ctx = _ctx(ctx)
    rgb, width, height = _rgb_bytes(x)
    nr, nc = C.c_int(0), C.c_int(0)
    # (a context-free call: it leaves no message in the context for ctx.check to report)
    if ctx.lib.imgfd_fhog_size(int(height), int(width), int(cell_size), int(filter_rows_padding), int(filter_cols_padding),
                               C.byref(nr), C.byref(nc)) != 0:
        raise ValueError("image_fhog: cell_size, filter_rows_padding and filter_cols_padding must be >= 1 (DLIB_ASSERT of fhog.h:712-720)")
    n = 31 * nr.value * nc.value
    flat = np.zeros((n,), np.float64)  # the glue: Rf_allocVector(REALSXP, n), filled by the library (widened on the device)
    if n:
        st = ctx.lib.imgfd_fhog_f64out(ctx.handle, rgb.ctypes.data_as(C.c_void_p), int(height), int(width), int(cell_size),
                                       int(filter_rows_padding), int(filter_cols_padding), flat.ctypes.data_as(C.c_void_p), n,
                                       C.byref(nr), C.byref(nc))
        ctx.check(st, "imgfd_fhog_f64out")
    # out$fhog <- array(out$fhog, dim = c(hog_height, hog_width, 31)): column-major fill
    res = RList(hog_height=nr.value, hog_width=nc.value,
                fhog=flat.reshape((nr.value, nc.value, 31), order="F"),
                hog_cell_size=int(cell_size), filter_rows_padding=int(filter_rows_padding),
                filter_cols_padding=int(filter_cols_padding))
    return res


def image_surf(x, max_points=1000, detection_threshold=30, ctx=None):
    """image_surf(): image_surf.R:83-90 over dlib_surf_points(), rcpp_surf.cpp:10-53."""
    ctx = _ctx(ctx)
    rgb, width, height = _rgb_bytes(x)
    out = _binding.SurfOut()
    st = ctx.lib.imgfd_surf_i32(ctx.handle, rgb.ctypes.data_as(C.c_void_p), int(height), int(width), int(max_points),
                                float(detection_threshold), C.byref(out))
    ctx.check(st, "imgfd_surf_i32")
    n = int(out.n)

    def vec(p, m):
        return np.ctypeslib.as_array(p, shape=(m,)).copy() if m else np.zeros((0,), np.float64)
    res = RList(points=n, x=vec(out.x, n), y=vec(out.y, n), angle=vec(out.angle, n), pyramid_scale=vec(out.pyramid_scale, n),
                score=vec(out.score, n), laplacian=vec(out.laplacian, n), surf=vec(out.surf, n * 64).reshape(n, 64))
    if n:
        ctx.lib.imgfd_free(out.data)
    res["surf"][np.isnan(res["surf"])] = 0  # out$surf[is.nan(out$surf)] <- 0, image_surf.R:88
    return res


def get_knnx(data, query, k=1, ctx=None):
    """The matching step of the reference's README (image.dlib/README.md:19-37): ``FNN::get.knnx(sp1$surf, sp2$surf, k)``.
    Returns, like FNN, ``nn.index`` (1-based rows of ``data``, shape (n_query, k)) and ``nn.dist`` (Euclidean, ascending)."""
    ctx = _ctx(ctx)
    data = np.ascontiguousarray(data, np.float64)
    query = np.ascontiguousarray(query, np.float64)
    if data.ndim != 2 or query.ndim != 2 or data.shape[1] != query.shape[1]:
        raise ValueError("data and query must be matrices with the same number of columns")
    if k > data.shape[0]:
        raise ValueError("k must not exceed the number of data points")  # FNN: "ANN: ERROR------->" / stop()
    nq = query.shape[0]
    idx = np.zeros((nq, k), np.int32)
    dist = np.zeros((nq, k), np.float64)
    st = ctx.lib.imgfd_knn(ctx.handle, data.ctypes.data_as(C.c_void_p), data.shape[0], query.ctypes.data_as(C.c_void_p), nq,
                           data.shape[1], int(k), 0, idx.ctypes.data_as(C.c_void_p), dist.ctypes.data_as(C.c_void_p))
    ctx.check(st, "imgfd_knn")
    return RList({"nn.index": idx + 1, "nn.dist": dist})
