"""The R glue EXECUTED (not only compiled): r/<pkg>/src/glue.c runs on a miniature R runtime (tests/rmini/rmini.c --
vectors, as.integer coercion, named lists, column-major matrices, PROTECT balance, Rf_error, the .Call registration
table) against the library under test, called the way each package's R wrapper calls it, and its R-level results are
compared with the oracle.  Reference: the Rcpp bodies the glue replaces --
image.CannyEdges/src/rcpp_canny.cpp:122-244 (+ R/canny_edges_detector.R:63-67), image.CornerDetectionF9/src/f9_rcpp.cpp:8-35
(+ R/image_detect_corners.R:48-60), image.CornerDetectionHarris/src/rcpp_harris.cpp:9-60 (+ R/pkg.R:70-104),
image.dlib/src/rcpp_fhog.cpp:10-46 and rcpp_surf.cpp:10-53 (+ R/image_fhog.R:35-48, R/image_surf.R:83-90).
R itself is not in this image: this is the closest executable check of the boundary R users hit."""
import os

import numpy as np
import pytest

import oracle
from image_amd import synth
from rmini_harness import ROOT, RError, RPackage

# (a sanitizer build of the emulator library, scripts/asan_emu.sh: the glue objects link the plain build and are loaded
# RTLD_DEEPBIND, which the sanitizer runtime does not support)
pytestmark = pytest.mark.skipif(bool(os.environ.get("IMGFD_EMU_LIB")), reason="glue objects link the plain emulator build")

_pkgs: dict = {}


def package(be, pkg):
    key = (be.name, pkg)
    if key not in _pkgs:
        lib = os.path.join(ROOT, "tests", "hipemu", "libimgfd_emu.so") if be.name == "emu" else os.path.join(ROOT, "image_amd", "libimgfd.so")
        _pkgs[key] = RPackage(pkg, lib, be.name)
    return _pkgs[key]


def r_matrix(p, m, integer=True):
    """as.integer(x) / as.numeric(x) of an R matrix given as numpy [row, col]: the column-major vector"""
    flat = np.asarray(m).flatten(order="F")
    return p.integer(flat) if integer else p.numeric(flat)


def test_canny_list_as_the_r_wrapper_gets_it(be):
    """image_canny_edge_detector(x): canny_edge_detector(as.integer(x), nrow(x), ncol(x), s, low_thr, high_thr, accGrad)"""
    p = package(be, "image.CannyEdges")
    nx, ny = 72, 50
    frame = synth.frame(21, nx, ny)                  # [y, x]
    x = frame.T                                      # the R matrix: nrow = nx, ncol = ny, x[i, j] = pixel (i, j)
    out = p.call("_image_CannyEdges_canny_edge_detector", r_matrix(p, x), p.integer(nx), p.integer(ny), p.numeric(2), p.numeric(3),
                 p.numeric(10), p.logical(True))
    assert list(out) == ["edges", "pixels_nonzero", "nx", "ny", "s", "low_thr", "high_thr", "accGrad"]   # rcpp_canny.cpp:236-243
    ref, n = oracle.canny(frame)
    assert out["edges"].shape == (nx, ny) and out["edges"].dtype == np.float64                            # NumericMatrix(nx, ny)
    assert np.array_equal(out["edges"], ref.T.astype(np.float64))
    assert out["pixels_nonzero"].dtype == np.int32 and int(out["pixels_nonzero"][0]) == n
    assert (float(out["nx"][0]), float(out["ny"][0]), float(out["s"][0]), float(out["low_thr"][0]), float(out["high_thr"][0])) == (nx, ny, 2.0, 3.0, 10.0)
    assert out["accGrad"].dtype == bool and bool(out["accGrad"][0])
    # a numeric (double) matrix goes through as.integer's truncation toward zero inside coerceVector
    out2 = p.call("_image_CannyEdges_canny_edge_detector", r_matrix(p, x + 0.75, integer=False), p.numeric(nx), p.numeric(ny), p.numeric(2),
                  p.numeric(3), p.numeric(10), p.logical(True))
    assert np.array_equal(out2["edges"], out["edges"])


def test_canny_r_errors(be):
    p = package(be, "image.CannyEdges")
    x = synth.frame(22, 16, 16).T
    with pytest.raises(RError, match="image must hold X\\*Y values"):
        p.call("_image_CannyEdges_canny_edge_detector", r_matrix(p, x), p.integer(32), p.integer(16), p.numeric(2), p.numeric(3), p.numeric(10), p.logical(True))
    with pytest.raises(RError, match="imgfd: .*positive"):      # the library's message reaches the R condition
        p.call("_image_CannyEdges_canny_edge_detector", r_matrix(p, x), p.integer(16), p.integer(16), p.numeric(0), p.numeric(3), p.numeric(10), p.logical(True))
    out = p.call("_image_CannyEdges_canny_edge_detector", r_matrix(p, x), p.integer(16), p.integer(16), p.numeric(2), p.numeric(3), p.numeric(10), p.logical(False))
    assert int(out["pixels_nonzero"][0]) == oracle.canny(x.T.copy(), accGrad=False)[1]   # the package is usable after an error
    with pytest.raises(AssertionError, match="registered with 7 arguments"):             # .Call checks the arity
        p.call("_image_CannyEdges_canny_edge_detector", r_matrix(p, x), p.integer(16))
    assert p.dll.rmini_dynamic_symbols() == 0                                            # R_useDynamicSymbols(dll, FALSE)


@pytest.mark.parametrize("nonmax", [False, True])
def test_fast9_list_as_the_r_wrapper_gets_it(be, nonmax):
    """image_detect_corners(x): detect_corners(as.integer(x), width = nrow(x), height = ncol(x), bytes_per_row = nrow(x), ...)"""
    p = package(be, "image.CornerDetectionF9")
    w, h = 96, 64
    frame = synth.frame(23, w, h)
    out = p.call("_image_CornerDetectionF9_detect_corners", r_matrix(p, frame.T), p.integer(w), p.integer(h), p.integer(w),
                 p.logical(nonmax), p.integer(30))
    ref = oracle.fast9(frame, 30, nonmax)
    assert isinstance(out, list) and len(out) == 2 and len(ref) > 5      # an unnamed list: the R wrapper names it (:57-58)
    assert out[0].dtype == np.float64
    assert np.array_equal(out[0], ref[:, 1].astype(np.float64))         # corners_x = out.y, f9_rcpp.cpp:29
    assert np.array_equal(out[1], (w - ref[:, 0]).astype(np.float64))   # corners_y = width - out.x, :30
    # values beyond a byte: (unsigned char) x[i] keeps the low 8 bits (f9_rcpp.cpp:10-11)
    wide = frame.T.astype(np.int64) + 256 * (np.arange(w * h).reshape(w, h) % 3)
    out2 = p.call("_image_CornerDetectionF9_detect_corners", r_matrix(p, wide), p.integer(w), p.integer(h), p.integer(w),
                  p.logical(nonmax), p.integer(30))
    assert np.array_equal(out2[0], out[0]) and np.array_equal(out2[1], out[1])


def _harris_args(p, x, w, h, **kw):
    d = dict(k=0.06, sigma_d=1.0, sigma_i=2.5, threshold=130.0, gaussian=1, gradient=0, strategy=0, Nselect=1, measure=0, Nscales=1,
             precision=1, cells=10, verbose=False)
    d.update(kw)
    return [r_matrix(p, x, integer=False), p.integer(w), p.integer(h), p.numeric(d["k"]), p.numeric(d["sigma_d"]), p.numeric(d["sigma_i"]),
            p.numeric(d["threshold"]), p.integer(d["gaussian"]), p.integer(d["gradient"]), p.integer(d["strategy"]), p.integer(d["Nselect"]),
            p.integer(d["measure"]), p.integer(d["Nscales"]), p.integer(d["precision"]), p.integer(d["cells"]), p.logical(d["verbose"])]


def test_harris_list_as_the_r_wrapper_gets_it(be):
    """detect_corners(x, nx = nrow(x), ny = ncol(x), ...) with the codes of the Rcpp-level defaults (gaussian 1 = fast Gaussian,
    precision 1 = quadratic approximation, RcppExports.R), then with the codes image_harris() itself always sends -- pkg.R:70-74
    evaluates `which(arg %in% match.arg(arg)) - 1L`, which is 0 for the default vector and for every single string: precise
    Gaussian, no sub-pixel step (the verbose call below)"""
    p = package(be, "image.CornerDetectionHarris")
    w, h = 128, 96
    frame = synth.frame(24, w, h)
    x = frame.T.astype(np.float64)
    out = p.call("_image_CornerDetectionHarris_detect_corners", *_harris_args(p, x, w, h, threshold=1.0))
    assert list(out) == ["x", "y", "strength"]                           # rcpp_harris.cpp:44-57
    ref = oracle.harris(frame.astype(np.float32), gaussian=1, precision=1, threshold=1.0)
    assert len(ref) > 5 and all(out[k].dtype == np.float64 for k in out)
    got = np.stack([out["x"], out["y"], out["strength"]], axis=1)
    assert np.array_equal(got[:, :2], ref[:, :2].astype(np.float64))
    assert np.all(np.abs(got[:, 2] - ref[:, 2]) <= 1e-4 * np.maximum(1.0, np.abs(ref[:, 2])))   # north_star tolerance (library default fir_mode)
    assert p.printed() == ""
    out_v = p.call("_image_CornerDetectionHarris_detect_corners", *_harris_args(p, x, w, h, verbose=True, gaussian=0, precision=0))
    log = p.printed()                                                    # harris.cpp:389-416, 504-538
    assert "Harris corner detection:" in log and f"[nx={w}, ny={h}, sigma_i=2.5" in log
    assert log.count("Time: ") == 7 and f"Number of corners detected: {len(out_v['x'])}" in log
    with pytest.raises(RError, match="x must hold nx\\*ny values"):
        p.call("_image_CornerDetectionHarris_detect_corners", *_harris_args(p, x, w, h + 1))
    empty = p.call("_image_CornerDetectionHarris_detect_corners", *_harris_args(p, x[:2], 2, h))   # harris.cpp:493: nothing, silently
    assert [len(empty[k]) for k in ("x", "y", "strength")] == [0, 0, 0]


def _rgb_array(rgb):
    """the R array dim = c(3, width, height) of an image [row, col, channel], as.integer'd: element [ch, c, r] at ch + 3*c + 3*cols*r"""
    return rgb.astype(np.int32).reshape(-1)


def test_fhog_list_as_the_r_wrapper_gets_it(be):
    """image_fhog(x): dlib_fhog(x, rows = height, cols = width, cell_size, filter_rows_padding, filter_cols_padding); the wrapper
    then reshapes $fhog with array(dim = c(hog_height, hog_width, 31)) (image_fhog.R:46)"""
    p = package(be, "image.dlib")
    rows, cols = 88, 120
    rgb = synth.frame_rgb(25, cols, rows)
    out = p.call("_image_dlib_dlib_fhog", p.integer(_rgb_array(rgb)), p.integer(rows), p.integer(cols), p.integer(8), p.integer(1), p.integer(1))
    assert list(out) == ["hog_height", "hog_width", "fhog", "hog_cell_size", "filter_rows_padding", "filter_cols_padding"]
    ref = oracle.fhog(rgb, 8, 1, 1)
    hh, hw = int(out["hog_height"][0]), int(out["hog_width"][0])
    assert (hh, hw) == ref.shape[:2] and out["fhog"].dtype == np.float64 and out["fhog"].shape == (31 * hh * hw,)
    cube = out["fhog"].reshape((hh, hw, 31), order="F")                 # what array(out$fhog, dim = ...) makes of it
    assert np.array_equal(cube, ref.astype(np.float64))
    assert (int(out["hog_cell_size"][0]), int(out["filter_rows_padding"][0]), int(out["filter_cols_padding"][0])) == (8, 1, 1)
    tiny = p.call("_image_dlib_dlib_fhog", p.integer(_rgb_array(rgb[:9, :9])), p.integer(9), p.integer(9), p.integer(8), p.integer(1), p.integer(1))
    assert int(tiny["hog_height"][0]) == 0 and len(tiny["fhog"]) == 0   # hog.clear(): an empty vector, no error
    with pytest.raises(RError, match="imgfd: .*fhog"):
        p.call("_image_dlib_dlib_fhog", p.integer(_rgb_array(rgb)), p.integer(rows), p.integer(cols), p.integer(0), p.integer(1), p.integer(1))


def test_surf_list_as_the_r_wrapper_gets_it(be):
    """image_surf(x): dlib_surf_points(x, rows = height, cols = width, max_points, detection_threshold)"""
    from test_surf import blobs
    p = package(be, "image.dlib")
    rows, cols = 240, 320
    rgb = blobs(26, cols, rows)
    out = p.call("_image_dlib_dlib_surf_points", p.integer(_rgb_array(rgb)), p.integer(rows), p.integer(cols), p.numeric(1000), p.numeric(30))
    assert list(out) == ["points", "x", "y", "angle", "pyramid_scale", "score", "laplacian", "surf"]   # rcpp_surf.cpp:45-52
    ref = oracle.surf(rgb, 1000, 30.0)
    n = len(ref["x"])
    assert n > 5 and float(out["points"][0]) == n and out["surf"].shape == (n, 64)                      # NumericMatrix(n, 64)
    for k in ("x", "y", "pyramid_scale", "score", "laplacian"):
        assert np.array_equal(out[k], np.asarray(ref[k], np.float64)), k
    assert np.allclose(out["angle"], ref["angle"], rtol=0, atol=1e-9)
    assert np.allclose(out["surf"], ref["surf"], rtol=0, atol=1e-9, equal_nan=True)
    none = p.call("_image_dlib_dlib_surf_points", p.integer(_rgb_array(rgb)), p.integer(rows), p.integer(cols), p.numeric(1000), p.numeric(1e12))
    assert float(none["points"][0]) == 0 and none["surf"].shape == (0, 64) and len(none["x"]) == 0


def test_unload_gives_the_context_back(be):
    """R_unload_<pkg> destroys the package's context; the next .Call creates a fresh one"""
    p = package(be, "image.CornerDetectionF9")
    frame = synth.frame(27, 64, 48)
    args = lambda: (r_matrix(p, frame.T), p.integer(64), p.integer(48), p.integer(64), p.logical(False), p.integer(30))
    a = p.call("_image_CornerDetectionF9_detect_corners", *args())
    p.unload()
    b = p.call("_image_CornerDetectionF9_detect_corners", *args())
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
