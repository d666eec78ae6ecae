"""The *_dev entry points cut large batches into sub-batches that fit their stage planes (imgfd_harris_dev / imgfd_canny_dev:
12 GiB, imgfd_fast9_dev: 1 GiB; config 3's 1024 frames cross that boundary at real size).  the lab switch "max_chunk_frames" lowers the
limit so that a batch of 7 small frames crosses two boundaries with a ragged tail: results must not depend on the cut."""
import numpy as np
import pytest

import oracle
from image_amd import synth

N, NX, NY = 7, 160, 96


@pytest.fixture
def frames():
    return np.stack([synth.frame(900 + f, NX, NY) for f in range(N)])


@pytest.mark.parametrize("chunk", ["3", "1"])
def test_harris_fast9_canny_batches_cut_into_sub_batches(be, frames, chunk):
    whole_h, cnt_h = be.harris_dev(frames, threshold=20.0)
    whole_f, cnt_f = be.fast9_dev(frames, 20, True)
    whole_e, cnt_e = be.canny_dev(frames)
    try:
        be.set_tuning("max_chunk_frames", int(chunk))
        cut_h, ccnt_h = be.harris_dev(frames, threshold=20.0)
        cut_f, ccnt_f = be.fast9_dev(frames, 20, True)
        cut_e, ccnt_e = be.canny_dev(frames)
    finally:
        be.set_tuning("max_chunk_frames", 0)
    assert np.array_equal(cnt_h, ccnt_h) and np.array_equal(cnt_f, ccnt_f) and np.array_equal(cnt_e, ccnt_e)
    assert np.array_equal(whole_e, cut_e)
    be.set_fir_mode(0)
    strict_h, _ = be.harris_dev(frames, threshold=20.0)
    be.set_fir_mode(1)
    for f in range(N):
        assert np.array_equal(whole_h[f], cut_h[f]) and np.array_equal(whole_f[f], cut_f[f])
        ref = oracle.harris(frames[f].astype(np.float32), threshold=20.0)
        assert len(ref) > 0 and np.array_equal(strict_h[f].view(np.uint32), ref.view(np.uint32))   # frame f's own corners, not a neighbour's
        assert np.array_equal(cut_f[f], oracle.fast9(frames[f], 20, True))
        e, n = oracle.canny(frames[f])
        assert int(ccnt_e[f]) == np.count_nonzero(cut_e[f]) and np.count_nonzero(cut_e[f] != e) <= 3


def test_batch_of_extreme_frames_through_the_three_detectors(be):
    """one batch whose frames are as unlike as frames get -- flat, one- and eight-pixel checkerboards, a step lattice, noise, a
    ramp -- through imgfd_harris_dev, imgfd_fast9_dev and imgfd_canny_dev: every frame as if it were alone (no state leaks from
    a dense frame into an empty neighbour: candidate lists, hysteresis flags, per-frame counts)"""
    import oracle
    from test_harris_api import extreme_frames
    fr = extreme_frames(192, 136)
    order = ["noise", "zeros", "checker8", "full", "steps", "checker1", "ramp", "noise"]
    frames = np.stack([fr[k].astype(np.uint8) for k in order])
    be.set_fir_mode(0)
    lists, counts = be.harris_dev(frames, threshold=1.0)
    pts, pc = be.fast9_dev(frames, 20, True)
    edges, ec = be.canny_dev(frames)
    for f, kind in enumerate(order):
        rh = oracle.harris(frames[f].astype(np.float32), threshold=1.0)
        assert int(counts[f]) == len(rh) and np.array_equal(lists[f].view(np.uint32), rh.view(np.uint32)), (kind, "harris")
        rf = oracle.fast9(frames[f], 20, True)
        assert int(pc[f]) == len(rf) and np.array_equal(pts[f], rf), (kind, "fast9")
        re, rn = oracle.canny(frames[f])
        bad = int(np.count_nonzero(edges[f] != re))
        assert bad <= (2 if kind == "ramp" else 0) and abs(int(ec[f]) - rn) <= bad, (kind, "canny", bad)   # (ramp: the exact-tie frame of test_canny.py)
