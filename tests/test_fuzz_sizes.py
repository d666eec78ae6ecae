"""Seeded size fuzz of the device-resident batch entry points: frame sizes that straddle every tile geometry the kernels
use (64-wide mask words, 16/24/64-row tiles, 16-byte aligned tile origins, 128-column marching strips), compared with the
oracle frame by frame.  Integer work exact; Harris strengths bit-equal in strict FIR mode; Canny within the 1e-5 bound."""
import numpy as np
import pytest

import oracle
from image_amd import synth

SIZES = [(3, 3), (7, 5), (15, 33), (16, 16), (17, 64), (63, 65), (64, 24), (65, 25), (80, 48), (96, 97), (127, 129),
         (128, 64), (129, 23), (144, 100), (191, 130), (200, 7), (256, 66), (272, 135), (333, 77)]


def _frames(nx, ny, n=2, seed=0):
    return np.stack([synth.frame(seed + 31 * f + nx + 7 * ny, max(nx, 16), max(ny, 16), n_rect=6)[:ny, :nx] for f in range(n)])


@pytest.mark.parametrize("nx,ny", SIZES)
def test_fast9_dev_sizes(be, nx, ny):
    fr = _frames(nx, ny)
    for nms in (False, True):
        lists, counts = be.fast9_dev(fr, 12, nms)
        for f in range(len(fr)):
            ref = oracle.fast9(fr[f], 12, nms)
            assert counts[f] == len(ref) and np.array_equal(lists[f], ref), (nx, ny, nms, f)


@pytest.mark.parametrize("nx,ny", SIZES)
def test_harris_dev_sizes(be, nx, ny):
    be.set_fir_mode(0)
    fr = _frames(nx, ny, seed=5)
    lists, counts = be.harris_dev(fr, threshold=20.0)
    for f in range(len(fr)):
        ref = oracle.harris(fr[f].astype(np.float32), threshold=20.0)
        assert counts[f] == len(ref), (nx, ny, f)
        assert np.array_equal(lists[f].view(np.uint32), ref.view(np.uint32)), (nx, ny, f)


@pytest.mark.parametrize("nx,ny", SIZES)
def test_canny_dev_sizes(be, nx, ny):
    fr = _frames(nx, ny, seed=9)
    edges, counts = be.canny_dev(fr)
    for f in range(len(fr)):
        ref, n = oracle.canny(fr[f])
        bad = int(np.count_nonzero(edges[f] != ref))
        assert bad <= 1e-5 * ref.size + 1, (nx, ny, f, bad)
        assert abs(int(counts[f]) - n) <= bad
