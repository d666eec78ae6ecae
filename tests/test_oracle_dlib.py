"""The dlib restatements (oracle/fhog_oracle.c, oracle/surf_oracle.c) against dlib itself compiled in place
(oracle/_ref/libref_dlib.so; only where /root/reference exists) and against the committed golden vectors."""
import numpy as np
import pytest

import oracle
from image_amd import synth

needs_ref = pytest.mark.skipif(not oracle.have_ref("dlib"), reason="oracle/_ref/libref_dlib.so not built")


@needs_ref
@pytest.mark.parametrize("w,h,cs,pr,pc", [(200, 150, 8, 1, 1), (203, 149, 8, 1, 1), (331, 257, 8, 3, 2), (120, 90, 6, 1, 1),
                                          (123, 97, 5, 2, 3), (64, 64, 16, 1, 1), (100, 100, 4, 1, 1), (20, 20, 8, 1, 1),
                                          (37, 29, 8, 1, 1), (511, 300, 8, 1, 1), (30, 41, 2, 1, 1), (90, 70, 3, 1, 1)])
def test_fhog_restatement_is_bit_identical_to_dlib(w, h, cs, pr, pc):
    rgb = synth.frame_rgb(7, w, h)
    a, b = oracle.ref_fhog(rgb, cs, pr, pc), oracle.fhog(rgb, cs, pr, pc)
    assert a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))


@needs_ref
@pytest.mark.parametrize("w,h,cs,pr,pc", [(96, 80, 32, 1, 1), (96, 80, 40, 1, 1), (96, 80, 64, 1, 1), (71, 53, 3, 1, 1), (71, 53, 7, 9, 9),
                                          (64, 64, 8, 12, 1), (33, 90, 11, 2, 5)])
def test_fhog_restatement_unusual_cell_sizes_and_paddings(w, h, cs, pr, pc):
    """the parameter sets of tests/test_fhog.py::test_unusual_cell_sizes_and_paddings, on the same noise"""
    rgb = np.random.default_rng(w + cs).integers(0, 256, (h, w, 3), dtype=np.uint8)
    a, b = oracle.ref_fhog(rgb, cs, pr, pc), oracle.fhog(rgb, cs, pr, pc)
    assert a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))


@needs_ref
@pytest.mark.parametrize("w,h,pr,pc", [(64, 48, 1, 1), (67, 35, 1, 1), (40, 30, 3, 2), (3, 3, 1, 1), (2, 5, 1, 1), (130, 17, 1, 1)])
def test_fhog_cell_size_1_restatement_is_bit_identical_to_dlib(w, h, pr, pc):
    rgb = synth.frame_rgb(4, max(w, 16), max(h, 16))[:h, :w]
    a, b = oracle.ref_fhog(rgb, 1, pr, pc), oracle.fhog(rgb, 1, pr, pc)
    assert a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))


@needs_ref
def test_fhog_colour_ties_and_gray():
    rng = np.random.default_rng(3)
    img = rng.integers(0, 4, (70, 93, 3)).astype(np.uint8) * 60   # many equal-length channel gradients
    a, b = oracle.ref_fhog(img), oracle.fhog(img)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    g = np.stack([synth.frame(9, 160, 120)] * 3, -1)
    assert np.array_equal(oracle.ref_fhog(g).view(np.uint32), oracle.fhog(g).view(np.uint32))


def test_fhog_golden(golden):
    g = golden("fhog_cruise_boat")
    for key, (cs, pr, pc) in {"hog_c8": (8, 1, 1), "hog_c4": (4, 1, 1), "hog_c8_p33": (8, 3, 3)}.items():
        got = oracle.fhog(g["image"], cs, pr, pc)
        assert got.shape == g[key].shape and np.array_equal(got.view(np.uint32), g[key].view(np.uint32)), key


# ----------------------------------------------------------------------------- SURF
def _blobs(seed, w, h):
    from test_surf import blobs
    return blobs(seed, w, h)


@needs_ref
@pytest.mark.parametrize("seed,w,h,thr", [(91, 320, 240, 30.0), (92, 517, 389, 5.0), (93, 700, 500, 30.0)])
def test_surf_restatement_is_bit_identical_to_dlib(seed, w, h, thr):
    rgb = _blobs(seed, w, h)
    assert np.array_equal(oracle.surf_integral(rgb), oracle.surf_integral(rgb, use_ref=True))
    a, b = oracle.surf_interest_points(rgb, thr), oracle.surf_interest_points(rgb, thr, use_ref=True)
    assert len(b) > 5 and a.shape == b.shape and np.array_equal(a, b)
    for mp in (40, 10000):
        x, y = oracle.surf(rgb, mp, thr), oracle.surf(rgb, mp, thr, use_ref=True)
        for k in y:
            assert x[k].shape == y[k].shape and np.array_equal(x[k], y[k]), (k, mp)


@needs_ref
@pytest.mark.parametrize("kind", ["zeros", "full", "noise", "checker8", "gray_ramp"])
def test_surf_restatement_on_extreme_images(kind):
    """the frames of tests/test_surf.py::test_extreme_images through dlib compiled in place and through the restatement"""
    w, h = 200, 152
    rng = np.random.default_rng(6)
    ramp = np.clip(np.add.outer(np.arange(h), np.arange(w)) // 2, 0, 255).astype(np.uint8)
    rgb = {"zeros": np.zeros((h, w, 3), np.uint8), "full": np.full((h, w, 3), 255, np.uint8),
           "noise": rng.integers(0, 256, (h, w, 3), dtype=np.uint8),
           "checker8": np.repeat(((np.add.outer(np.arange(h) // 8, np.arange(w) // 8) & 1) * 255).astype(np.uint8)[:, :, None], 3, axis=2),
           "gray_ramp": np.repeat(ramp[:, :, None], 3, axis=2)}[kind]
    for thr, max_points in ((0.0, 10000), (30.0, 7), (1e9, 1000)):
        a, b = oracle.surf_interest_points(rgb, thr), oracle.surf_interest_points(rgb, thr, use_ref=True)
        assert a.shape == b.shape and np.array_equal(a, b), (kind, thr)
        x, y = oracle.surf(rgb, max_points, thr), oracle.surf(rgb, max_points, thr, use_ref=True)
        for k in y:
            assert x[k].shape == y[k].shape and np.array_equal(x[k], y[k], equal_nan=True), (kind, thr, k)


def test_surf_golden(golden):
    g = golden("surf_cruise_boat")
    got = oracle.surf(g["image"], 1000, 30.0)
    for k in ("x", "y", "angle", "pyramid_scale", "score", "laplacian", "surf"):
        assert got[k].shape == g[k].shape and np.array_equal(got[k], g[k]), k
    assert len(g["x"]) > 100
