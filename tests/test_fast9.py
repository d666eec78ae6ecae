"""FAST-9: bit-exact coordinates, in order, against the reference's own output (golden) and the oracle."""
import numpy as np
import pytest

import oracle
from image_amd import synth


@pytest.mark.parametrize("thr", [20, 80, 100])
@pytest.mark.parametrize("nms", [0, 1])
def test_chairs_golden(be, golden, thr, nms):
    g = golden("fast9_chairs")
    got = be.fast9(g["image"], thr, bool(nms))
    assert np.array_equal(got, g[f"xy_t{thr}_n{nms}"])


def test_chairs_anchor(be, golden):
    img = golden("fast9_chairs")["image"]
    assert len(be.fast9(img, 80, False)) == 926 and len(be.fast9(img, 80, True)) == 347


@pytest.mark.parametrize("thr", [10, 20, 50])
@pytest.mark.parametrize("nms", [0, 1])
def test_synth_golden(be, golden, thr, nms):
    g = golden("fast9_synth_640x480_seed1")
    assert np.array_equal(be.fast9(g["image"], thr, bool(nms)), g[f"xy_t{thr}_n{nms}"])


@pytest.mark.parametrize("w,h", [(6, 6), (7, 7), (8, 30), (64, 7), (65, 9), (130, 67)])
def test_small_sizes_vs_oracle(be, w, h):
    img = synth.frame(3, max(w, 16), max(h, 16))[:h, :w]
    for thr, nms in [(0, False), (0, True), (15, True), (255, False)]:
        assert np.array_equal(be.fast9(img, thr, nms), oracle.fast9(img, thr, nms)), (w, h, thr, nms)


def test_stride_larger_than_width(be):
    buf = np.zeros((100, 128), np.uint8)
    buf[:, :100] = synth.frame(3, 100, 100)
    buf[:, 100:] = 255  # padding must never be read as image
    got = be.fast9(buf, 20, True, width=100)
    assert np.array_equal(got, oracle.fast9(buf, 20, True, width=100)) and len(got) > 0


def test_saturating_extremes(be):
    rng = np.random.default_rng(1)
    img = rng.choice(np.array([0, 1, 254, 255], np.uint8), size=(60, 90))
    for thr in (0, 1, 200, 254, 255):
        for nms in (False, True):
            assert np.array_equal(be.fast9(img, thr, nms), oracle.fast9(img, thr, nms)), (thr, nms)


def test_batch_dev(be):
    frames = np.stack([synth.frame(200 + f, 160, 120) for f in range(4)])
    for nms in (False, True):
        lists, counts = be.fast9_dev(frames, 20, nms)
        for f in range(4):
            ref = oracle.fast9(frames[f], 20, nms)
            assert counts[f] == len(ref) and np.array_equal(lists[f], ref)


@pytest.mark.parametrize("kind", ["zeros", "full", "checker", "noise", "steps"])
def test_extreme_images(be, kind):
    """flat black / white frames (no corner, whatever the threshold), a one-pixel checkerboard, uniform noise at the lowest
    thresholds (a corner at nearly every pixel: the candidate lists fill up) and saturated steps: the reference's points in
    the reference's order"""
    w, h = 150, 97
    rng = np.random.default_rng(5)
    img = {"zeros": np.zeros((h, w), np.uint8), "full": np.full((h, w), 255, np.uint8),
           "checker": ((np.add.outer(np.arange(h), np.arange(w)) & 1) * 255).astype(np.uint8),
           "noise": rng.integers(0, 256, (h, w)).astype(np.uint8),
           "steps": np.repeat(np.repeat(rng.integers(0, 2, (h // 8 + 1, w // 8 + 1)) * 255, 8, 0), 8, 1)[:h, :w].astype(np.uint8)}[kind]
    for thr in (0, 1, 127, 254, 255):
        for nms in (False, True):
            assert np.array_equal(be.fast9(img, thr, nms), oracle.fast9(img, thr, nms)), (kind, thr, nms)
