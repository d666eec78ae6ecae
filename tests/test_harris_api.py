"""imgfd_harris (host-pointer drop-in path) against the golden vectors produced by the reference's own
code (tests/golden/harris_*.npz, scripts/make_golden.py) and against the oracle on seeded frames."""
import numpy as np
import pytest

import oracle
from image_amd import synth

# gaussian code 1 (SII) cases are appended once K6 lands
CASES = {
    "default": dict(),
    "no_gaussian_unsupported_sii": None,  # NO_GAUSSIAN remaps the tensor smoothing to SII (harris.cpp:64-65)
    "sobel": dict(gradient=1),
    "shi_tomasi": dict(measure=1, threshold=1.0),
    "harmonic": dict(measure=2, threshold=1.0),
    "quartic": dict(precision=2),
    "sorted": dict(strategy=1),
    "n_corners": dict(strategy=2, Nselect=50),
    "distributed": dict(strategy=3, Nselect=100),
    "three_scales": dict(Nscales=3),
}
# gaussian codes 1 and 2 run the stacked-integral-images Gaussian (sii.hip): sequential float prefix sums, bit-exact
CASES.update({"rcpp_default": dict(gaussian=1, precision=1), "no_gaussian": dict(gaussian=2),
              "two_scales": dict(gaussian=1, Nscales=2)})


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.mark.parametrize("fixture", ["harris_building", "harris_synth_640x480_seed1"])
@pytest.mark.parametrize("case", [c for c, kw in CASES.items() if kw is not None])
def test_golden_strict_bit_exact(be, golden, fixture, case):
    g = golden(fixture)
    be.set_fir_mode(0)
    got = be.harris(g["image"], **CASES[case])
    ref = g["xyR_" + case]
    assert got.shape == ref.shape, (got.shape, ref.shape)
    assert np.array_equal(bits(got), bits(ref)), f"{fixture}/{case}"


@pytest.mark.parametrize("fixture", ["harris_building", "harris_synth_640x480_seed1"])
def test_golden_default_mode_coordinates_exact_strength_1e4(be, golden, fixture):
    """library default (fused f64 accumulate): coordinates exact, strength within north_star's 1e-4."""
    g = golden(fixture)
    be.set_fir_mode(1)
    got = be.harris(g["image"])
    be.set_fir_mode(0)
    ref = g["xyR_default"]
    assert got.shape == ref.shape
    assert np.array_equal(got[:, :2], ref[:, :2])
    assert np.all(np.abs(got[:, 2] - ref[:, 2]) <= 1e-4 * np.maximum(1.0, np.abs(ref[:, 2])))


def test_building_anchor(be, golden):
    """BASELINE.md anchor: 251 corners, sum x 75383, sum y 50407, first (6, 85, 30426.795)."""
    got = be.harris(golden("harris_building")["image"])
    assert len(got) == 251 and got[:, 0].sum() == 75383 and got[:, 1].sum() == 50407
    assert got[0, 0] == 6 and got[0, 1] == 85 and abs(got[0, 2] - 30426.795) < 0.01


@pytest.mark.parametrize("nx,ny", [(2, 50), (50, 2), (3, 3), (11, 11), (12, 12), (65, 70), (129, 31)])
def test_small_and_degenerate_sizes(be, nx, ny):
    img = synth.frame(5, max(nx, 8), max(ny, 8))[:ny, :nx].astype(np.float32)
    be.set_fir_mode(0)
    got = be.harris(img, threshold=1.0)
    ref = oracle.harris(img, threshold=1.0)
    assert got.shape == ref.shape and np.array_equal(bits(got), bits(ref))


def test_batch_dev_matches_single(be):
    frames = np.stack([synth.frame(100 + f, 200, 120) for f in range(3)])
    be.set_fir_mode(0)
    lists, counts = be.harris_dev(frames)
    for f in range(3):
        ref = oracle.harris(frames[f].astype(np.float32))
        assert counts[f] == len(ref)
        assert np.array_equal(bits(lists[f]), bits(ref))
    # f32 frames take the same path
    lists32, _ = be.harris_dev(frames.astype(np.float32))
    for f in range(3):
        assert np.array_equal(bits(lists32[f]), bits(lists[f]))


def test_batch_dev_cap_truncates_but_counts_all(be):
    frames = synth.frame(7, 320, 240)[None]
    full, counts = be.harris_dev(frames, threshold=10.0)
    assert counts[0] > 8
    part, counts2 = be.harris_dev(frames, cap=8, threshold=10.0)
    assert counts2[0] == counts[0] and len(part[0]) == 8
    assert np.array_equal(bits(part[0]), bits(full[0][:8]))


@pytest.mark.parametrize("kw", [dict(measure=1, threshold=1.0), dict(measure=2, threshold=1.0), dict(sigma_i=4.0),
                                dict(sigma_i=0.4, threshold=50.0), dict(gradient=1), dict(gaussian=2),
                                dict(sigma_i=3.0, threshold=5.0)])
def test_batch_dev_options(be, kw):
    """the fused response+NMS kernel (window radius <= 6), its two-kernel fallback (sigma_i 4 -> radius 8) and the
    other measures / gradients give the reference's corner list in the batch path too"""
    frames = np.stack([synth.frame(120 + f, 180, 131) for f in range(2)])
    be.set_fir_mode(0)
    lists, counts = be.harris_dev(frames, **kw)
    for f in range(2):
        ref = oracle.harris(frames[f].astype(np.float32), **kw)
        assert counts[f] == len(ref), kw
        assert np.array_equal(bits(lists[f]), bits(ref)), kw


@pytest.mark.parametrize("kw", [dict(sigma_i=22.0, threshold=0.5), dict(sigma_d=21.5, sigma_i=2.5, threshold=0.001)])
def test_gaussians_of_more_than_64_taps(be, kw):
    """gaussian.cpp:289-330 stops only at size > xdim: sigma 22 is 67 taps per side -- more than a kernel argument holds, so
    the generic FIR passes read them from a device buffer.  Host-pointer and batch entry against the restatement (bit for bit
    in strict mode) and the reference's own sources"""
    img = synth.frame(131, 260, 200)
    be.set_fir_mode(0)
    ref = oracle.harris(img.astype(np.float32), **kw)
    assert len(ref) >= 1
    got = be.harris(img.astype(np.float32), **kw)
    assert np.array_equal(bits(got), bits(ref)), kw
    lists, counts = be.harris_dev(img[None], **kw)
    assert counts[0] == len(ref) and np.array_equal(bits(lists[0]), bits(ref)), kw
    if oracle.have_ref("harris"):
        assert np.array_equal(bits(oracle.ref_harris(img.astype(np.float32), threads=2, **kw)), bits(ref))


def test_batch_dev_tile_borders(be):
    """sizes that straddle the 64x32 tiles of the fused kernel and its candidate list"""
    be.set_fir_mode(0)
    for nx, ny in [(64, 32), (65, 33), (127, 63), (130, 97), (13, 40)]:
        img = synth.frame(9, max(nx, 16), max(ny, 16))[:ny, :nx]
        lists, counts = be.harris_dev(img[None], threshold=1.0)
        ref = oracle.harris(img.astype(np.float32), threshold=1.0)
        assert counts[0] == len(ref) and np.array_equal(bits(lists[0]), bits(ref)), (nx, ny)


UNUSUAL = [dict(k=0.0, threshold=1.0), dict(k=-0.05, threshold=1.0), dict(k=0.25, threshold=0.0), dict(threshold=-10.0),
           dict(sigma_i=0.5, threshold=1.0), dict(sigma_i=0.3, threshold=1.0), dict(sigma_d=0.3, threshold=1.0), dict(sigma_d=0.0, threshold=1.0),
           dict(strategy=2, Nselect=100000, threshold=1.0), dict(strategy=2, Nselect=0, threshold=1.0), dict(strategy=1, Nselect=3, threshold=1.0),
           dict(strategy=3, Nselect=7, cells=1, threshold=1.0), dict(strategy=3, Nselect=500, cells=200, threshold=1.0),
           dict(strategy=3, Nselect=0, cells=4, threshold=1.0),
           dict(Nscales=6, threshold=1.0), dict(Nscales=6, gaussian=1, precision=2, threshold=1.0), dict(Nscales=0, threshold=1.0),
           dict(gaussian=2, gradient=1, measure=1, precision=1, threshold=0.5), dict(gaussian=1, gradient=1, measure=2, precision=2, threshold=0.1),
           dict(measure=2, threshold=0.0), dict(gaussian=1, sigma_d=3.0, sigma_i=6.0, threshold=0.01)]


@pytest.mark.parametrize("kw", UNUSUAL, ids=[",".join(f"{k}={v}" for k, v in d.items()) for d in UNUSUAL])
def test_unusual_parameters_follow_the_reference(be, kw):
    """parameter values at and beyond the edges of what image_harris() documents -- zero / negative k and threshold, sigmas
    that give the smallest Gaussian radius, selections that ask for more corners or cells than there are, more scales than
    the frame has octaves -- through imgfd_harris in strict mode: the restated reference's list, bit for bit"""
    img = synth.frame(41, 96, 72).astype(np.float32)
    be.set_fir_mode(0)
    got = be.harris(img, **kw)
    ref = oracle.harris(img, **kw)
    assert got.shape == ref.shape and np.array_equal(bits(got), bits(ref)), (kw, len(got), len(ref))


def extreme_frames(w=160, h=120):
    rng = np.random.default_rng(9)
    return {"zeros": np.zeros((h, w), np.float32), "full": np.full((h, w), 255, np.float32),
            "checker8": ((np.add.outer(np.arange(h) // 8, np.arange(w) // 8) & 1) * 255).astype(np.float32),
            "checker1": ((np.add.outer(np.arange(h), np.arange(w)) & 1) * 255).astype(np.float32),
            "noise": rng.integers(0, 256, (h, w)).astype(np.float32),
            "ramp": np.clip(np.add.outer(np.arange(h), np.arange(w)), 0, 255).astype(np.float32),
            "steps": np.repeat(np.repeat(rng.integers(0, 2, (h // 8, w // 8)) * 255, 8, 0), 8, 1).astype(np.float32)}


EXTREME_KW = (dict(threshold=1.0), dict(threshold=1.0, gaussian=1, precision=1), dict(threshold=0.001, measure=1))


@pytest.mark.parametrize("kind", ["zeros", "full", "checker8", "checker1", "noise", "ramp", "steps"])
def test_extreme_images(be, kind):
    """flat frames and a ramp (no corner), a one-pixel checkerboard (the smoothing erases it), an 8-pixel checkerboard and random
    8-pixel steps (a lattice of equally strong corners: exact ties of the response in rows and columns), noise: the restated
    reference's corners, bit for bit, in strict mode"""
    img = extreme_frames()[kind]
    be.set_fir_mode(0)
    for kw in EXTREME_KW:
        got, ref = be.harris(img, **kw), oracle.harris(img, **kw)
        assert got.shape == ref.shape and np.array_equal(bits(got), bits(ref)), (kind, kw, len(got), len(ref))
    assert kind not in ("checker8", "steps", "noise") or len(ref) > 50
