"""imgfd_frames need not be packed (include/imgfd.h: rows row_stride_bytes apart, frames frame_stride_bytes apart): the three
gray-frame detectors on frames with padded rows and gaps between frames return what they return on packed frames -- through
the aligned fast paths (padding a multiple of 16 bytes: the marching Gaussian/gradient kernel, 16-byte tile loads) and
through the general ones (odd padding)."""
import numpy as np
import pytest

from image_amd import synth


@pytest.mark.parametrize("row_pad,gap_rows", [(16, 0), (48, 3), (5, 1), (0, 2)])
def test_padded_rows_and_frame_gaps(be, row_pad, gap_rows):
    nx, ny, n = 512, 72, 3
    frames = np.stack([synth.frame(900 + f, nx, ny, n_rect=12) for f in range(n)])
    ref_h, ref_hc = be.harris_dev(frames, threshold=40.0)
    ref_p, ref_pc = be.fast9_dev(frames, 15, True)
    ref_e, ref_ec = be.canny_dev(frames)
    assert sum(int(c) for c in ref_hc) > 10 and sum(int(c) for c in ref_pc) > 10
    with be.padded(row_pad, gap_rows):
        n0 = be.get_counter("gauss_march_launches")
        h, hc = be.harris_dev(frames, threshold=40.0)
        marched = be.get_counter("gauss_march_launches") - n0
        p, pc = be.fast9_dev(frames, 15, True)
        e, ec = be.canny_dev(frames)
    assert marched == (1 if row_pad % 16 == 0 and ((ny + gap_rows) * (nx + row_pad)) % 16 == 0 else 0)
    assert np.array_equal(hc, ref_hc) and np.array_equal(pc, ref_pc) and np.array_equal(ec, ref_ec)
    for f in range(n):
        assert np.array_equal(h[f].view(np.uint32), ref_h[f].view(np.uint32))
        assert np.array_equal(p[f], ref_p[f])
    assert np.array_equal(e, ref_e)
