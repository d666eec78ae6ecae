"""The oracle itself (CPU, no GPU): the C restatements in oracle/ against

  * the golden vectors in tests/golden/ (outputs of the reference's own code, scripts/make_golden.py);
  * the reference compiled in place (oracle/_ref), when /root/reference was present at build time
    (this container; never on the GPU box) -- stage by stage and end to end, bit for bit;
  * for Canny (reference not compilable: FFTW3 absent -> PARITY UNPINNED) a literal numpy restatement of
    tools.c:166-185 (fft2 product) and the BASELINE.md anchors.
"""
import numpy as np
import pytest

import oracle
from image_amd import synth
from scripts_path import HARRIS_CASES

needs_ref = pytest.mark.skipif(not (oracle.have_ref("harris") and oracle.have_ref("f9")),
                               reason="oracle/_ref not built (needs /root/reference)")


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.mark.parametrize("fixture", ["harris_building", "harris_synth_640x480_seed1"])
@pytest.mark.parametrize("case", sorted(HARRIS_CASES))
def test_harris_restatement_matches_reference_golden(golden, fixture, case):
    g = golden(fixture)
    ref = g["xyR_" + case]
    got = oracle.harris(g["image"].astype(np.float32), **HARRIS_CASES[case])
    assert got.shape == ref.shape, (case, got.shape, ref.shape)
    assert np.array_equal(bits(got), bits(ref)), case


def test_harris_building_anchor(golden):
    """SURVEY 8c anchor: building.rds, R-default codes -> 251 corners, first = (6, 85, 30426.795)."""
    got = oracle.harris(golden("harris_building")["image"].astype(np.float32))
    assert got.shape[0] == 251
    assert got[0, 0] == 6 and got[0, 1] == 85 and abs(got[0, 2] - 30426.795) < 0.01
    assert int(got[:, 0].sum()) == 75383 and int(got[:, 1].sum()) == 50407


@pytest.mark.parametrize("fixture,thrs", [("fast9_chairs", (20, 50, 80, 100)), ("fast9_synth_640x480_seed1", (10, 20, 50))])
def test_fast9_restatement_matches_reference_golden(golden, fixture, thrs):
    g = golden(fixture)
    for thr in thrs:
        for nms in (0, 1):
            key = f"xy_t{thr}_n{nms}"
            if key not in g.files:
                continue
            assert np.array_equal(oracle.fast9(g["image"], thr, bool(nms)), g[key]), key


def test_fast9_chairs_anchor(golden):
    img = golden("fast9_chairs")["image"]
    assert oracle.fast9(img, 80, False).shape[0] == 926
    assert oracle.fast9(img, 80, True).shape[0] == 347
    assert oracle.fast9(img, 100, False).shape[0] == 411


@needs_ref
@pytest.mark.parametrize("seed,nx,ny", [(31, 97, 64), (32, 320, 200), (33, 641, 479)])
def test_harris_stages_against_compiled_reference(seed, nx, ny):
    img = synth.frame(seed, nx, ny).astype(np.float32)
    for sigma in (1.0, 2.5):
        a = oracle.harris_stage("gaussian", img, sigma=sigma, type=0)
        b = oracle.harris_stage("gaussian", img, sigma=sigma, type=0, use_ref=True)
        assert np.array_equal(bits(a), bits(b)), ("gaussian", sigma)
    Is = oracle.harris_stage("gaussian", img, sigma=1.0, type=0)
    for t in (0, 1):
        a = oracle.harris_stage("gradient", Is, type=t)
        b = oracle.harris_stage("gradient", Is, type=t, use_ref=True)
        assert all(np.array_equal(bits(x), bits(y)) for x, y in zip(a, b)), ("gradient", t)
    ix, iy = oracle.harris_stage("gradient", Is, type=0)
    a = oracle.harris_stage("autocorrelation", ix, iy, sigma=2.5, gauss=0)
    b = oracle.harris_stage("autocorrelation", ix, iy, sigma=2.5, gauss=0, use_ref=True)
    assert all(np.array_equal(bits(x), bits(y)) for x, y in zip(a, b)), "autocorrelation"
    for m in (0, 1, 2):
        ra = oracle.harris_stage("response", *a, measure=m, k=0.06)
        rb = oracle.harris_stage("response", *a, measure=m, k=0.06, use_ref=True)
        assert np.array_equal(bits(ra), bits(rb)), ("response", m)
    R = oracle.harris_stage("response", *a, measure=0, k=0.06)
    na = oracle.harris_stage("nms", R, Th=130.0, radius=5)
    nb = oracle.harris_stage("nms", R, Th=130.0, radius=5, use_ref=True)
    assert np.array_equal(bits(na), bits(nb)), "nms"


@needs_ref
@pytest.mark.parametrize("case", sorted(HARRIS_CASES))
def test_harris_end_to_end_against_compiled_reference(case):
    img = synth.frame(34, 400, 300).astype(np.float32)
    a = oracle.harris(img, **HARRIS_CASES[case])
    b = oracle.ref_harris(img, **HARRIS_CASES[case])
    assert a.shape == b.shape and np.array_equal(bits(a), bits(b)), case


@needs_ref
def test_fast9_against_compiled_reference():
    rng = np.random.default_rng(5)
    for i in range(6):
        img = (synth.frame(40 + i, 200, 120) if i % 2 else rng.integers(0, 256, (90, 131)).astype(np.uint8))
        for thr in (0, 7, 20, 50, 119):
            for nms in (False, True):
                assert np.array_equal(oracle.fast9(img, thr, nms), oracle.ref_fast9(img, thr, nms)), (i, thr, nms)


@needs_ref
def test_fast9_extreme_images_against_compiled_reference():
    """the frames of tests/test_fast9.py::test_extreme_images (flat, checkerboard, noise, saturated steps) at thresholds 0, 1,
    127, 254, 255"""
    w, h = 150, 97
    rng = np.random.default_rng(5)
    imgs = [np.zeros((h, w), np.uint8), np.full((h, w), 255, np.uint8),
            ((np.add.outer(np.arange(h), np.arange(w)) & 1) * 255).astype(np.uint8), rng.integers(0, 256, (h, w)).astype(np.uint8),
            np.repeat(np.repeat(rng.integers(0, 2, (h // 8 + 1, w // 8 + 1)) * 255, 8, 0), 8, 1)[:h, :w].astype(np.uint8)]
    for i, img in enumerate(imgs):
        for thr in (0, 1, 127, 254, 255):
            for nms in (False, True):
                assert np.array_equal(oracle.fast9(img, thr, nms), oracle.ref_fast9(img, thr, nms)), (i, thr, nms)


# ------------------------------------------------------------------ Canny
# pinned by the reference's own rcpp_canny.cpp + tools.c + adsf.c compiled in place (oracle/_ref/libref_canny.so); FFTW3,
# which the reference links but does not vendor, is replaced by the plain DFT of oracle/fftw_stub.c
needs_ref_canny = pytest.mark.skipif(not oracle.have_ref("canny"), reason="oracle/_ref/libref_canny.so not built (needs /root/reference)")


@needs_ref_canny
@pytest.mark.parametrize("nx,ny,kw", [(160, 120, {}), (200, 131, dict(accGrad=False)), (97, 64, dict(s=3.5)),
                                      (131, 90, dict(s=1.0, low_thr=2, high_thr=6)), (64, 64, dict(s=0.7, low_thr=0, high_thr=1)),
                                      (33, 47, dict(s=5.0, low_thr=0.5, high_thr=2.5)), (16, 9, {}),
                                      # thresholds the interface does not forbid: crossed, negative, equal, beyond every magnitude
                                      (90, 70, dict(low_thr=10, high_thr=3)), (90, 70, dict(low_thr=-5, high_thr=2)),
                                      (90, 70, dict(low_thr=-3.7, high_thr=-1.2)), (90, 70, dict(low_thr=3, high_thr=3)),
                                      (90, 70, dict(low_thr=200, high_thr=300))])
def test_canny_restatement_matches_reference(nx, ny, kw):
    img = synth.frame(60 + nx, max(nx, 16), max(ny, 16), n_rect=9)[:ny, :nx]
    ref_e, ref_n = oracle.ref_canny(img, **kw)
    e, n = oracle.canny(img, **kw)
    assert ref_n == int(np.count_nonzero(ref_e))
    assert n == ref_n and np.array_equal(e, ref_e)


@needs_ref_canny
@pytest.mark.parametrize("kind", ["zeros", "full", "checker1", "checker8", "ramp", "one_pixel"])
def test_canny_restatement_on_extreme_images(kind):
    """the frame kinds of tests/test_canny.py::test_extreme_images (smaller: the stand-in DFT is O(n^3)) through the compiled
    reference and the restatement: equal pixel for pixel, the tie-ridden ramp included"""
    w, h = 64, 48
    one = np.zeros((h, w), np.uint8); one[h // 2, w // 3] = 255
    img = {"zeros": np.zeros((h, w), np.uint8), "full": np.full((h, w), 255, np.uint8),
           "checker1": ((np.add.outer(np.arange(h), np.arange(w)) & 1) * 255).astype(np.uint8),
           "checker8": ((np.add.outer(np.arange(h) // 8, np.arange(w) // 8) & 1) * 255).astype(np.uint8),
           "ramp": np.clip(np.add.outer(np.arange(h) + 70, np.arange(w) + 100), 0, 255).astype(np.uint8), "one_pixel": one}[kind]
    for kw in (dict(), dict(accGrad=False), dict(s=0.8, low_thr=0, high_thr=1), dict(s=4.0, low_thr=1, high_thr=2)):
        ref_e, ref_n = oracle.ref_canny(img, **kw)
        e, n = oracle.canny(img, **kw)
        assert n == ref_n and np.array_equal(e, ref_e), (kind, kw)


@needs_ref_canny
def test_canny_reference_on_noise():
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (70, 101)).astype(np.uint8)
    for kw in (dict(), dict(accGrad=False, low_thr=20, high_thr=60)):
        ref_e, ref_n = oracle.ref_canny(img, **kw)
        e, n = oracle.canny(img, **kw)
        assert np.count_nonzero(e != ref_e) <= 1e-5 * e.size + 1 and abs(n - ref_n) <= 1      # <= 1e-5


@pytest.mark.parametrize("fixture,cases", [("canny_chairs", ("a0", "a1")),
                                           ("canny_synth_320x240_seed7", ("a0", "a1", "s1_t2_6", "s3p5_t1_4"))])
def test_canny_restatement_matches_reference_golden(golden, fixture, cases):
    """goldens written by the reference's own canny_edge_detector() (scripts/make_golden.py)"""
    from scripts_path import CANNY_CASES
    g = golden(fixture)
    assert int(g["pinned"]) == 1
    for case in cases:
        e, n = oracle.canny(g["image"], **CANNY_CASES[case])
        assert n == int(g[f"nonzero_{case}"]), case
        assert np.array_equal(np.packbits(e > 0), g[f"edges_bits_{case}"]), case


_numpy_gblur = oracle.fft_gblur   # tools.c:146-185 literally, through numpy's FFT


@pytest.mark.parametrize("nx,ny,s", [(64, 48, 2.0), (128, 100, 2.0), (90, 61, 1.0), (75, 40, 3.0)])
def test_canny_blur_equals_fft_product(nx, ny, s):
    img = synth.frame(50, nx, ny)
    _, _, dbg = oracle.canny(img, s=s, debug=True)
    fft = _numpy_gblur(img, s).astype(np.float64)
    # an FFT carries ~1e-13 absolute error on 0..255 data: equal up to the last float bit
    assert np.max(np.abs(dbg["blur"] - fft)) <= 2.0 ** -16


@pytest.mark.parametrize("seed,nx,ny", [(52, 640, 480), (53, 1000, 700), (2, 3840, 2160)])
def test_canny_edges_through_an_independent_fft(seed, nx, ny):
    """What the absent FFTW3 could change: the reference's blur is ifft2(fft2(x) fft2(g)) rounded to float (tools.c:166-185,
    :126).  An FFT in double carries ~1e-13 absolute error on 0..255 data against a float spacing of 1.5e-5: a pixel's float
    can only flip when the exact value lies within that error of a rounding boundary, ~1e-8 of the pixels.  Measured through
    pocketfft (numpy) -- an FFT with its own butterflies and rounding: NO float of the blur differs from the restatement's
    (direct circular sums in double) at 640x480, 1000x700 and the 4K bench frame, and no edge pixel.  The bounds asserted
    are SURVEY 8d's expected rate; the stages behind the blur are the same code."""
    img = synth.frame(seed, nx, ny)
    edges, n, dbg = oracle.canny(img, debug=True)
    fft_blur = _numpy_gblur(img, 2.0)
    assert np.count_nonzero(fft_blur.astype(np.float64) != dbg["blur"]) <= 1e-3 * img.size   # 1-ulp float differences, rare
    e2, n2 = oracle.canny_from_blur(fft_blur)
    assert np.count_nonzero(e2 != edges) <= 1e-5 * img.size
    same, n_same = oracle.canny_from_blur(dbg["blur"])                                      # the composition itself is exact
    assert n_same == n and np.array_equal(same, edges)


def test_canny_anchors(golden):
    g = golden("canny_chairs")
    _, n = oracle.canny(g["image"])
    assert n == 38012 == int(g["nonzero_a1"])
    _, n0 = oracle.canny(g["image"], accGrad=False)
    assert n0 == 24621


def test_canny_hysteresis_is_connected_components():
    """edges == marked pixels whose 8-connected component contains a strong pixel (adsf.c semantics)."""
    from scipy import ndimage
    img = synth.frame(51, 160, 120)
    edges, n, dbg = oracle.canny(img, debug=True)
    lab, k = ndimage.label(dbg["nms"] > 0, structure=np.ones((3, 3)))
    keep = np.zeros(k + 1, bool)
    keep[np.unique(lab[dbg["nms"] == 2])] = True
    keep[0] = False
    assert np.array_equal(edges > 0, keep[lab])
    assert n == int(np.count_nonzero(edges))


@pytest.mark.skipif(not oracle.have_ref("harris"), reason="oracle/_ref/libref_harris.so not built (needs /root/reference)")
def test_harris_restatement_matches_reference_on_unusual_parameters():
    """the parameter sets of tests/test_harris_api.py::test_unusual_parameters_follow_the_reference (zero / negative k and
    threshold, minimal sigmas, selections larger than the corner list, more scales than octaves ...) through the reference
    compiled in place and through its restatement: the same lists, bit for bit"""
    from test_harris_api import UNUSUAL
    img = synth.frame(41, 96, 72).astype(np.float32)
    for kw in UNUSUAL:
        a, b = oracle.harris(img, **kw), oracle.ref_harris(img, threads=1, **kw)
        assert a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32)), kw


@pytest.mark.skipif(not oracle.have_ref("harris"), reason="oracle/_ref/libref_harris.so not built (needs /root/reference)")
def test_harris_restatement_matches_reference_on_extreme_images():
    """the frames of tests/test_harris_api.py::test_extreme_images (flat, checkerboards, steps, ramp, noise) through the
    reference compiled in place (one thread: its NMS is schedule-dependent on ties) and through its restatement"""
    from test_harris_api import EXTREME_KW, extreme_frames
    for kind, img in extreme_frames().items():
        for kw in EXTREME_KW:
            a, b = oracle.harris(img, **kw), oracle.ref_harris(img, threads=1, **kw)
            assert a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32)), (kind, kw)
