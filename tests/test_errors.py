"""Error behaviour of the C ABI: bad arguments come back as status codes with a message (nothing throws or longjmps
across the boundary), the reference's silent-empty cases stay silent, and a failed call leaves the context usable."""
import ctypes as C

import numpy as np
import pytest

from image_amd import _binding, synth

OK, INVALID, UNSUPPORTED = 0, 1, 5


def msg(be):
    return (be.lib.imgfd_last_error(be.ctx) or b"").decode()


def test_harris_bad_arguments(be):
    out = _binding.Corners()
    img = np.zeros((8, 8), np.float32)
    st = be.lib.imgfd_harris(be.ctx, None, 8, 8, 0.06, 1.0, 2.5, 130.0, 0, 0, 0, 1, 0, 1, 0, 10, 0, C.byref(out))
    assert st == INVALID and "imgfd_harris" in msg(be)
    st = be.lib.imgfd_harris(be.ctx, img.ctypes.data_as(C.c_void_p), -1, 8, 0.06, 1.0, 2.5, 130.0, 0, 0, 0, 1, 0, 1, 0, 10, 0, C.byref(out))
    assert st == INVALID
    # harris.cpp:493: fewer than 3 rows or columns -> silently no corners
    st = be.lib.imgfd_harris(be.ctx, img.ctypes.data_as(C.c_void_p), 2, 8, 0.06, 1.0, 2.5, 130.0, 0, 0, 0, 1, 0, 1, 0, 10, 0, C.byref(out))
    assert st == OK and out.n == 0
    # the context still works
    assert len(be.harris(synth.frame(1, 64, 48).astype(np.float32), threshold=1.0)) > 0


def test_fast9_bad_arguments(be):
    out = _binding.Points()
    img = np.zeros((16, 16), np.uint8)
    assert be.lib.imgfd_fast9(be.ctx, None, 16, 16, 16, 20, 0, C.byref(out)) == INVALID
    assert be.lib.imgfd_fast9(be.ctx, img.ctypes.data_as(C.c_void_p), 16, 16, 8, 20, 0, C.byref(out)) == INVALID  # stride < width
    assert "geometry" in msg(be)
    assert be.lib.imgfd_fast9(be.ctx, img.ctypes.data_as(C.c_void_p), 6, 6, 16, 20, 0, C.byref(out)) == OK and out.n == 0  # empty domain


def test_fast9_dev_degenerate_frames(be):
    """the batch entry point validates its frame geometry: zero or negative sizes are refused (a zero-row frame used to
    divide by zero), frames too small for a ring answer zero corners like the host entry point"""
    d = be.to_dev(np.zeros((2, 16, 16), np.uint8))
    cnt = be.empty((2,), np.int64)
    pts = be.empty((2, 8, 2), np.int32)
    for nx, ny in ((16, 0), (0, 16), (-3, 16), (16, -1)):
        fr = _binding.Frames(be.ptr(d), 2, nx, ny, 256, 16, 0)
        assert be.lib.imgfd_fast9_dev(be.ctx, C.byref(fr), 20, 0, be.ptr(pts), 8, be.ptr(cnt)) == INVALID, (nx, ny)
    fr = _binding.Frames(be.ptr(d), 2, 16, 16, 256, 8, 0)   # row stride < width
    assert be.lib.imgfd_fast9_dev(be.ctx, C.byref(fr), 20, 0, be.ptr(pts), 8, be.ptr(cnt)) == INVALID
    cnt2 = be.to_dev(np.full((2,), 77, np.int64))
    fr = _binding.Frames(be.ptr(d), 2, 6, 16, 256, 16, 0)   # narrower than a ring: empty search domain
    assert be.lib.imgfd_fast9_dev(be.ctx, C.byref(fr), 20, 0, be.ptr(pts), 8, be.ptr(cnt2)) == OK
    be.sync()
    assert be.to_host(cnt2).tolist() == [0, 0]


def test_canny_bad_arguments(be):
    img = np.zeros((16, 16), np.uint8)
    edges = np.zeros((16, 16), np.uint8)
    n = C.c_int64(0)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    assert be.lib.imgfd_canny(be.ctx, p(img), 16, 16, 0.0, 3.0, 10.0, 1, p(edges), C.byref(n)) == INVALID and "positive" in msg(be)
    assert be.lib.imgfd_canny(be.ctx, p(img), 0, 16, 2.0, 3.0, 10.0, 1, p(edges), C.byref(n)) == INVALID
    assert be.lib.imgfd_canny(be.ctx, p(img), 16, 16, 2.0, 3.0, 10.0, 1, None, C.byref(n)) == INVALID
    e, k = be.canny(synth.frame(2, 40, 30))
    assert k == np.count_nonzero(e)


def test_fhog_bad_arguments(be):
    rgb = np.zeros((32, 32, 3), np.uint8)
    hog = C.POINTER(C.c_float)(); nr = C.c_int(0); nc = C.c_int(0)
    p = rgb.ctypes.data_as(C.c_void_p)
    assert be.lib.imgfd_fhog(be.ctx, p, 32, 32, 0, 1, 1, C.byref(hog), C.byref(nr), C.byref(nc)) == INVALID
    assert be.lib.imgfd_fhog(be.ctx, p, 32, 32, 8, 0, 1, C.byref(hog), C.byref(nr), C.byref(nc)) == INVALID
    assert be.lib.imgfd_fhog_size(32, 32, 8, 1, 1, C.byref(nr), C.byref(nc)) == OK and (nr.value, nc.value) == (2, 2)
    assert be.lib.imgfd_fhog_size(32, 32, 8, 3, 5, C.byref(nr), C.byref(nc)) == OK and (nr.value, nc.value) == (4, 6)


def test_surf_bad_arguments(be):
    rgb = np.zeros((64, 64, 3), np.uint8)
    o = _binding.SurfOut()
    p = rgb.ctypes.data_as(C.c_void_p)
    assert be.lib.imgfd_surf(be.ctx, p, 64, 64, 0, 30.0, C.byref(o)) == INVALID      # max_points > 0 (surf.h:243)
    assert be.lib.imgfd_surf(be.ctx, p, 64, 64, 10, -1.0, C.byref(o)) == INVALID     # detection_threshold >= 0
    assert be.lib.imgfd_surf(be.ctx, p, 64, 64, 10, 30.0, C.byref(o)) == OK and o.n == 0  # flat image: no points


def test_dev_api_rejects_wrong_dtype(be):
    frames = synth.frame(3, 64, 48).astype(np.float32)[None]
    d = be.to_dev(frames)
    fr = be.frames(d, 1, 64, 48, 1)  # f32 frames are Harris-only
    pts = be.empty((1, 16, 2), np.int32); cnt = be.empty((1,), np.int64)
    assert be.lib.imgfd_fast9_dev(be.ctx, C.byref(fr), 20, 0, be.ptr(pts), 16, be.ptr(cnt)) == INVALID
    edges = be.empty((1, 48, 64), np.uint8)
    assert be.lib.imgfd_canny_dev(be.ctx, C.byref(fr), 2.0, 3.0, 10.0, 1, be.ptr(edges), be.ptr(cnt)) == INVALID


def test_counts_only_batch_calls_accept_null_record_buffers(be):
    """cap = 0 means counts only: the record buffers may be NULL (imgfd_harris_dev, imgfd_fast9_dev, imgfd_detect_dev)"""
    import oracle
    frames = np.stack([synth.frame(640 + f, 96, 72, n_rect=8) for f in range(2)])
    d = be.to_dev(frames)
    fr = be.frames(d, 2, 96, 72, 0)
    cnt = be.empty((3, 2), np.int64)
    edges = be.empty((2, 72, 96), np.uint8)
    p = _binding.StreamParams()
    be.lib.imgfd_stream_default_params(C.byref(p))
    p.threshold, p.fast9_threshold, p.suppress_non_max = 40.0, 15, 1
    be.set_fir_mode(0)
    st = be.lib.imgfd_detect_dev(be.ctx, C.byref(fr), C.byref(p), None, None, be.ptr(edges), be.ptr(cnt))
    assert st == OK, msg(be)
    be.sync()
    got = be.to_host(cnt)
    for f in range(2):
        assert got[0, f] == len(oracle.harris(frames[f].astype(np.float32), threshold=40.0))
        assert got[1, f] == len(oracle.fast9(frames[f], 15, True))
        assert got[2, f] == oracle.canny(frames[f])[1]
    # a record buffer is still required as soon as records are asked for
    p.corner_cap = 4
    assert be.lib.imgfd_detect_dev(be.ctx, C.byref(fr), C.byref(p), None, None, be.ptr(edges), be.ptr(cnt)) == INVALID
    # an empty batch is not an error
    fr0 = be.frames(d, 0, 96, 72, 0)
    p.corner_cap = 0
    assert be.lib.imgfd_detect_dev(be.ctx, C.byref(fr0), C.byref(p), None, None, be.ptr(edges), be.ptr(cnt)) == OK, msg(be)


def test_frames_beyond_32_bit_indexing_are_refused(be):
    """a frame of 2^31 pixels (RGB: bytes) or more would wrap the kernels' 32-bit pixel indices: refused at the boundary
    before anything is read (the pointers here are far too small for such a frame)"""
    big = 46400   # 46400^2 > 2^31
    img = np.zeros((16, 16), np.uint8)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    n = C.c_int64(0)
    assert be.lib.imgfd_canny(be.ctx, p(img), big, big, 2.0, 3.0, 10.0, 1, p(img), C.byref(n)) == INVALID
    out = _binding.Corners()
    assert be.lib.imgfd_harris(be.ctx, p(img), big, big, 0.06, 1.0, 2.5, 130.0, 0, 0, 0, 1, 0, 1, 0, 10, 0, C.byref(out)) == INVALID
    pts = _binding.Points()
    assert be.lib.imgfd_fast9(be.ctx, p(img), big, big, big, 20, 0, C.byref(pts)) == INVALID
    hog = C.POINTER(C.c_float)(); nr = C.c_int(0); nc = C.c_int(0)
    assert be.lib.imgfd_fhog(be.ctx, p(img), 26800, 26800, 8, 1, 1, C.byref(hog), C.byref(nr), C.byref(nc)) == INVALID   # x 3 bytes
    d = be.to_dev(np.zeros((1, 16, 16), np.uint8))
    cnt = be.empty((1,), np.int64)
    fr = _binding.Frames(be.ptr(d), 1, big, big, big * big, big, 0)
    assert be.lib.imgfd_canny_dev(be.ctx, C.byref(fr), 2.0, 3.0, 10.0, 1, be.ptr(d), be.ptr(cnt)) == INVALID
    assert be.lib.imgfd_fast9_dev(be.ctx, C.byref(fr), 20, 0, None, 0, be.ptr(cnt)) == INVALID
    e, k = be.canny(synth.frame(2, 40, 30))   # the context still works
    assert k == np.count_nonzero(e)
