"""Host frame streams (imgfd_stream_*): every frame that goes through the double-buffered upload
pipeline must come out exactly as the oracle computes it on that frame -- whatever the batch boundaries, whether the
frames were pageable, pinned or strided, and in submission order."""
import ctypes as C
import types

import numpy as np
import pytest

import oracle
from image_amd import _binding, framestream, synth

NX, NY = 112, 88


def _ctx(be):
    return types.SimpleNamespace(lib=be.lib, handle=be.ctx, check=lambda st, what="": be.check(st, what))


@pytest.fixture(scope="module")
def frames():
    return np.stack([synth.frame(900 + f, NX, NY, n_rect=14) for f in range(11)])


@pytest.fixture(scope="module")
def expected(frames):
    out = []
    for img in frames:
        h = oracle.harris(img.astype(np.float32), threshold=60.0)
        f9 = oracle.fast9(img, 20, True)
        e, n = oracle.canny(img)
        out.append((h, f9, e, n))
    return out


def _check(res_list, expected, first=0, harris_exact=True):
    seen = first
    for res in res_list:
        assert res["first_frame"] == seen
        for f in range(res["n_frames"]):
            h, f9, e, n = expected[seen + f]
            assert res["harris_counts"][f] == len(h)
            got = res["corners"][f]
            assert np.array_equal(got["x"], h[:, 0]) and np.array_equal(got["y"], h[:, 1])
            if harris_exact:
                assert np.array_equal(got["R"].view(np.uint32), h[:, 2].copy().view(np.uint32))
            else:
                assert np.allclose(got["R"], h[:, 2], rtol=1e-4)
            assert res["fast9_counts"][f] == len(f9)
            assert np.array_equal(res["points"][f]["x"], f9[:, 0]) and np.array_equal(res["points"][f]["y"], f9[:, 1])
            assert res["canny_counts"][f] == n
            assert np.array_equal(res["edges"][f], e)
        seen += res["n_frames"]
    return seen


def _open(be, **kw):
    be.set_fir_mode(0)
    p = dict(corner_cap=2048, point_cap=2048, keep_edges=True, threshold=60.0, fast9_threshold=20, suppress_non_max=1)
    p.update(kw)
    return framestream.FrameStream(NX, NY, batch=4, ctx=_ctx(be), **p)


def test_ragged_batches_of_pageable_frames(be, frames, expected):
    with _open(be) as fs:
        chunks = [frames[0:4], frames[4:7], frames[7:8], frames[8:11]]
        res = list(fs.run(chunks))
        assert [r["n_frames"] for r in res] == [4, 3, 1, 3]
        assert _check(res, expected) == 11
        assert fs.collect() is None                      # nothing pending


def test_pinned_and_strided_frames(be, frames, expected):
    with _open(be) as fs:
        pin = framestream.PinnedFrames(4, NY, NX, lib=be.lib)
        pin.array[:] = frames[0:4]
        fs.submit(pin.array)
        # a strided view: every other frame of a larger pageable block
        block = np.zeros((6, NY, NX), np.uint8)
        block[::2] = frames[4:7]
        fs.submit(block[::2])
        a = fs.collect()
        b = fs.collect()
        assert _check([a, b], expected) == 7
        pin.free()


def test_counts_only_and_single_detector(be, frames, expected):
    be.set_fir_mode(0)
    with framestream.FrameStream(NX, NY, batch=8, ctx=_ctx(be), fast9=False, canny=False, threshold=60.0) as fs:
        fs.submit(frames[:8])
        fs.submit(frames[8:])
        r0, r1 = fs.collect(), fs.collect()
        assert r0["fast9_counts"] is None and r0["canny_counts"] is None and "corners" not in r0
        counts = list(r0["harris_counts"]) + list(r1["harris_counts"])
        assert counts == [len(e[0]) for e in expected]


def test_canny_only_and_two_detectors(be, frames, expected):
    be.set_fir_mode(0)
    with framestream.FrameStream(NX, NY, batch=4, ctx=_ctx(be), harris=False, fast9=False, keep_edges=True) as fs:
        fs.submit(frames[:4])
        r = fs.collect()
        assert r["harris_counts"] is None and list(r["canny_counts"]) == [e[3] for e in expected[:4]]
        assert all(np.array_equal(r["edges"][f], expected[f][2]) for f in range(4))
    with framestream.FrameStream(NX, NY, batch=4, ctx=_ctx(be), harris=False, point_cap=512, fast9_threshold=20,
                                 suppress_non_max=1) as fs:                     # FAST-9 beside Canny: the two-stream schedule
        fs.submit(frames[4:8])
        r = fs.collect()
        assert list(r["canny_counts"]) == [e[3] for e in expected[4:8]]
        assert list(r["fast9_counts"]) == [len(e[1]) for e in expected[4:8]]
        assert np.array_equal(r["points"][2]["x"], expected[6][1][:, 0])


def test_pipeline_depth_and_argument_errors(be, frames):
    with _open(be) as fs:
        fs.submit(frames[0:2])
        fs.submit(frames[2:4])
        with pytest.raises(_binding.ImgfdError, match="two batches pending"):
            fs.submit(frames[4:6])
        assert fs.collect()["n_frames"] == 2
        fs.submit(frames[4:6])                            # a slot is free again
        with pytest.raises(_binding.ImgfdError):                # more frames than the batch size
            be.check(fs.lib.imgfd_stream_submit(fs.handle, C.c_void_p(frames.ctypes.data), 5, NX * NY), "submit")
        with pytest.raises(ValueError):
            fs.submit(frames[:, :, :50])
        assert fs.collect()["first_frame"] == 2 and fs.collect()["first_frame"] == 4
    p = _binding.StreamParams()
    be.lib.imgfd_stream_default_params(C.byref(p))
    assert (p.k, p.sigma_d, p.sigma_i, p.threshold) == (pytest.approx(0.06), 1.0, 2.5, 130.0)
    assert (p.fast9_threshold, p.suppress_non_max, p.s, p.low_thr, p.high_thr, p.accGrad) == (50, 0, 2.0, 3.0, 10.0, 1)
    h = C.c_void_p()
    p.harris = p.fast9 = p.canny = 0
    assert be.lib.imgfd_stream_open(be.ctx, NX, NY, 4, C.byref(p), C.byref(h)) == 1 and not h
