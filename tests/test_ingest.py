"""The R-native entry points (*_f64 / *_i32): the glue's narrowing casts applied on the device give exactly the
results of the byte / float entry points, including the wrap-around of (unsigned char) on out-of-range integers."""
import ctypes as C

import numpy as np
import pytest

import oracle
from image_amd import _binding, synth


def test_fast9_i32_matches_u8_and_truncates_like_unsigned_char(be):
    img = synth.frame(201, 120, 90).astype(np.int32)
    img[10:20, 10:30] += 256            # (unsigned char) 300 == 44
    img[40:50, 5:9] = -3                # (unsigned char) -3 == 253
    out = _binding.Points()
    be.check(be.lib.imgfd_fast9_i32(be.ctx, img.ctypes.data_as(C.c_void_p), 120, 90, 120, 20, 1, C.byref(out)), "fast9_i32")
    got = np.ctypeslib.as_array(C.cast(out.points, C.POINTER(C.c_int)), shape=(out.n, 2)).copy() if out.n else np.zeros((0, 2), np.int32)
    if out.n:
        be.lib.imgfd_free(out.points)
    ref = oracle.fast9((img & 0xFF).astype(np.uint8), 20, True)
    assert len(ref) > 0 and np.array_equal(got, ref)


def test_canny_i32_matches_u8(be):
    img = synth.frame(202, 96, 70)
    wide = img.astype(np.int32) + 512   # wraps back to the same bytes
    edges = np.zeros((70, 96), np.uint8); n = C.c_int64(0)
    be.check(be.lib.imgfd_canny_i32(be.ctx, wide.ctypes.data_as(C.c_void_p), 96, 70, 2.0, 3.0, 10.0, 1,
                                    edges.ctypes.data_as(C.c_void_p), C.byref(n)), "canny_i32")
    e2, n2 = be.canny(img)
    assert np.array_equal(edges, e2) and n.value == n2


def test_harris_f64_matches_f32(be):
    img = synth.frame(203, 160, 110).astype(np.float64) + 0.123456789   # not representable in float: the cast rounds
    out = _binding.Corners()
    be.set_fir_mode(0)
    be.check(be.lib.imgfd_harris_f64(be.ctx, img.ctypes.data_as(C.c_void_p), 160, 110, 0.06, 1.0, 2.5, 130.0, 0, 0, 0, 1, 0, 1, 0, 10, 0,
                                     C.byref(out)), "harris_f64")
    got = np.ctypeslib.as_array(C.cast(out.corners, C.POINTER(C.c_float)), shape=(out.n, 3)).copy()
    be.lib.imgfd_free(out.corners)
    ref = oracle.harris(img.astype(np.float32))
    assert len(ref) > 0 and np.array_equal(got.view(np.uint32), ref.view(np.uint32))


def test_fhog_and_surf_i32_match_bytes(be):
    rgb = synth.frame_rgb(204, 136, 104)
    wide = np.ascontiguousarray(rgb.astype(np.int32) + 256 * 7)
    hog = C.POINTER(C.c_float)(); nr = C.c_int(0); nc = C.c_int(0)
    be.check(be.lib.imgfd_fhog_i32(be.ctx, wide.ctypes.data_as(C.c_void_p), 104, 136, 8, 1, 1, C.byref(hog), C.byref(nr), C.byref(nc)), "fhog_i32")
    flat = np.ctypeslib.as_array(hog, shape=(31, nc.value, nr.value)).copy()
    be.lib.imgfd_free(hog)
    assert np.array_equal(np.ascontiguousarray(flat.transpose(2, 1, 0)).view(np.uint32), oracle.fhog(rgb).view(np.uint32))
    o = _binding.SurfOut()
    be.check(be.lib.imgfd_surf_i32(be.ctx, wide.ctypes.data_as(C.c_void_p), 104, 136, 100, 5.0, C.byref(o)), "surf_i32")
    ref = oracle.surf(rgb, 100, 5.0)
    assert o.n == len(ref["x"])
    if o.n:
        assert np.array_equal(np.ctypeslib.as_array(o.surf, shape=(o.n * 64,)), ref["surf"].ravel())
        be.lib.imgfd_free(o.data)


def test_f64out_entry_points_match_the_byte_and_float_ones(be):
    """imgfd_canny_f64out / imgfd_fhog_f64out: the same results as imgfd_canny_i32 / imgfd_fhog_i32, widened to doubles on the
    device and written into the caller's vector (the NumericMatrix / NumericVector the R glue allocates); sizes from
    imgfd_fhog_size, a vector that is too short is refused, an image too small for cells leaves it untouched"""
    img = synth.frame(205, 200, 150)
    i32 = np.ascontiguousarray(img.astype(np.int32) + 256 * 3)   # (unsigned char) truncation keeps the low byte
    e8 = np.zeros((150, 200), np.uint8); e64 = np.full((150, 200), -1.0); n8 = C.c_int64(0); n64 = C.c_int64(0)
    be.check(be.lib.imgfd_canny_i32(be.ctx, i32.ctypes.data_as(C.c_void_p), 200, 150, 2.0, 3.0, 10.0, 1, e8.ctypes.data_as(C.c_void_p), C.byref(n8)), "canny_i32")
    be.check(be.lib.imgfd_canny_f64out(be.ctx, i32.ctypes.data_as(C.c_void_p), 200, 150, 2.0, 3.0, 10.0, 1, e64.ctypes.data_as(C.c_void_p), C.byref(n64)), "canny_f64out")
    assert n8.value == n64.value > 0 and np.array_equal(e64, e8.astype(np.float64)) and set(np.unique(e64)) <= {0.0, 255.0}
    ref, rn = oracle.canny(img)
    assert rn == n64.value and np.count_nonzero((e64 > 0) != (ref > 0)) <= 3

    rgb = synth.frame_rgb(206, 136, 104)
    wide = np.ascontiguousarray(rgb.astype(np.int32))
    nr = C.c_int(0); nc = C.c_int(0)
    be.check(be.lib.imgfd_fhog_size(104, 136, 8, 1, 1, C.byref(nr), C.byref(nc)), "fhog_size")
    n = 31 * nr.value * nc.value
    out = np.full((n + 5,), -7.0)
    be.check(be.lib.imgfd_fhog_f64out(be.ctx, wide.ctypes.data_as(C.c_void_p), 104, 136, 8, 1, 1, out.ctypes.data_as(C.c_void_p), n, C.byref(nr), C.byref(nc)), "fhog_f64out")
    flat = out[:n].reshape(31, nc.value, nr.value)
    assert np.array_equal(np.ascontiguousarray(flat.transpose(2, 1, 0)).astype(np.float32).view(np.uint32), oracle.fhog(rgb).view(np.uint32))
    assert np.array_equal(flat.astype(np.float32).astype(np.float64), flat) and (out[n:] == -7.0).all()
    assert be.lib.imgfd_fhog_f64out(be.ctx, wide.ctypes.data_as(C.c_void_p), 104, 136, 8, 1, 1, out.ctypes.data_as(C.c_void_p), n - 1, C.byref(nr), C.byref(nc)) != 0
    tiny = np.zeros((10, 10, 3), np.int32)
    be.check(be.lib.imgfd_fhog_f64out(be.ctx, tiny.ctypes.data_as(C.c_void_p), 10, 10, 8, 1, 1, out.ctypes.data_as(C.c_void_p), n, C.byref(nr), C.byref(nc)), "fhog_f64out tiny")
    assert nr.value == 0 and nc.value == 0
