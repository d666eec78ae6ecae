"""Canny: the device path against the oracle restatement and against edge maps written by the reference's own
canny_edge_detector() (compiled in place with a stand-in DFT for the absent FFTW3, see oracle/canny_oracle.c and
tests/test_oracle.py).  The device blur keeps the taps >= 1e-17 (the oracle >= 1e-22) and accumulates with fused
multiply-adds: the blurred doubles agree to ~1e-16 and, after the float cast, in all but astronomically rare values, so
almost every decision is reproduced exactly; the NMS replaces atan2->cos/sin by the unit vector, which can flip exact ties
only: the edge map is compared through a mismatch-rate bound."""
import numpy as np
import pytest

import oracle
from image_amd import synth


def mismatch(a, b):
    return int(np.count_nonzero(a != b))


@pytest.mark.parametrize("acc", [0, 1])
def test_chairs_golden(be, golden, acc):
    g = golden("canny_chairs")
    edges, n = be.canny(g["image"], accGrad=bool(acc))
    ref = np.unpackbits(g[f"edges_bits_a{acc}"]).reshape(edges.shape) * 255
    assert set(np.unique(edges)) <= {0, 255}
    assert n == int(np.count_nonzero(edges))
    bad = mismatch(edges, ref)
    assert bad <= 1e-4 * edges.size, (bad, n, int(g[f"nonzero_a{acc}"]))
    assert abs(n - int(g[f"nonzero_a{acc}"])) <= 1e-3 * int(g[f"nonzero_a{acc}"])


def test_reference_golden_all_parameter_sets(be, golden):
    """edge maps of the reference's own code on a synthetic frame, four parameter sets (scripts/make_golden.py)"""
    from scripts_path import CANNY_CASES
    g = golden("canny_synth_320x240_seed7")
    for case, kw in CANNY_CASES.items():
        edges, n = be.canny(g["image"], **kw)
        ref = np.unpackbits(g[f"edges_bits_{case}"]).reshape(edges.shape) * 255
        assert mismatch(edges, ref) <= 1e-5 * edges.size + 1, case     # mismatch rate <= 1e-5
        assert abs(n - int(g[f"nonzero_{case}"])) <= 1, case


def test_chairs_anchor(be, golden):
    """BASELINE.md anchor (survey restatement): chairs.pgm defaults -> pixels_nonzero 38 012."""
    _, n = be.canny(golden("canny_chairs")["image"])
    assert abs(n - 38012) <= 38


@pytest.mark.parametrize("nx,ny", [(1, 1), (2, 3), (5, 4), (31, 17), (64, 64), (100, 37), (257, 129)])
def test_small_and_odd_sizes(be, nx, ny):
    img = synth.frame(21, max(nx, 16), max(ny, 16))[:ny, :nx]
    for kw in (dict(), dict(accGrad=False), dict(s=1.0, low_thr=2.5, high_thr=6.9)):
        edges, n = be.canny(img, **kw)
        ref, rn = oracle.canny(img, **kw)
        assert mismatch(edges, ref) <= max(2, 1e-3 * edges.size), (nx, ny, kw)
        assert n == int(np.count_nonzero(edges))


def test_thresholds_are_int_truncated(be):
    img = synth.frame(22, 120, 90)
    a, _ = be.canny(img, low_thr=3.0, high_thr=10.0)
    b, _ = be.canny(img, low_thr=3.9, high_thr=10.9)
    assert np.array_equal(a, b)


def test_hysteresis_long_chain(be):
    """a weak chain that snakes across many 64x64 tiles must light up from a single strong seed"""
    img = np.full((200, 300), 60, np.uint8)
    img[100:, :] = 66          # faint horizontal step -> weak edge along the whole row
    img[100:, 5:9] = 140       # a short strong segment seeds it
    edges, n = be.canny(img, s=1.0, low_thr=2, high_thr=40)
    ref, rn = oracle.canny(img, s=1.0, low_thr=2, high_thr=40)
    assert rn > 150 and mismatch(edges, ref) <= 2, (n, rn)


def test_batch_dev(be):
    frames = np.stack([synth.frame(300 + f, 160, 96) for f in range(3)])
    edges, counts = be.canny_dev(frames)
    for f in range(3):
        ref, rn = oracle.canny(frames[f])
        assert mismatch(edges[f], ref) <= 3
        assert counts[f] == np.count_nonzero(edges[f])
