"""BASELINE.json's full sizes on the device (GPU only): the oracle still finishes in seconds at 3840x2160, so the 4K
frames are compared directly; the 4096x4096 RGB tiles (config 4) are checked on a band against the oracle and as a
whole through size-independent properties (tile-translation invariance of fHOG cells, determinism, bounds)."""
import numpy as np
import pytest

import oracle
from image_amd import synth

pytestmark = pytest.mark.gpu
NX, NY = 3840, 2160


@pytest.fixture(scope="module")
def gpu():
    import backends
    return backends.GpuBackend()


@pytest.fixture(scope="module")
def frame4k():
    return synth.frame(2, NX, NY)  # G(seed=2)


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def test_harris_4k_strict_bit_exact_and_default_within_tolerance(gpu, frame4k):
    ref = oracle.harris(frame4k.astype(np.float32))
    assert len(ref) > 1000
    gpu.set_fir_mode(0)
    lists, counts = gpu.harris_dev(frame4k[None], cap=65536)
    assert counts[0] == len(ref) and np.array_equal(bits(lists[0]), bits(ref))
    host = gpu.harris(frame4k.astype(np.float32))           # host-pointer drop-in path
    assert np.array_equal(bits(host), bits(ref))
    gpu.set_fir_mode(1)                                     # library default: fused f64 accumulate
    lists, counts = gpu.harris_dev(frame4k[None], cap=65536)
    gpu.set_fir_mode(0)
    got = lists[0]
    assert counts[0] == len(ref) and np.array_equal(got[:, :2], ref[:, :2])      # coordinates exact, in order
    assert np.all(np.abs(got[:, 2] - ref[:, 2]) <= 1e-4 * np.maximum(1.0, np.abs(ref[:, 2])))  # north_star tolerance
    assert int((bits(got[:, 2]) != bits(ref[:, 2])).sum()) <= 2                  # in fact (almost) the same bits


@pytest.mark.parametrize("thr,nms", [(50, False), (50, True), (20, False), (20, True)])   # thr 50 / 20, with and without NMS
def test_fast9_4k_bit_exact(gpu, frame4k, thr, nms):
    ref = oracle.fast9(frame4k, thr, nms)
    lists, counts = gpu.fast9_dev(frame4k[None], thr, nms, cap=1 << 20)
    assert counts[0] == len(ref) and np.array_equal(lists[0], ref)
    assert np.array_equal(gpu.fast9(frame4k, thr, nms), ref)


def test_canny_4k(gpu, frame4k):
    ref, n = oracle.canny(frame4k)
    edges, counts = gpu.canny_dev(frame4k[None])
    bad = int(np.count_nonzero(edges[0] != ref))
    assert bad <= 1e-5 * ref.size, (bad, n, int(counts[0]))   # expected mismatch rate <= 1e-5
    assert int(counts[0]) == int(np.count_nonzero(edges[0]))
    assert set(np.unique(edges[0])) <= {0, 255}


def test_canny_1080p_batch(gpu):
    """config 3 shape: a batch of 1920x1080 frames (a few of the 1024)"""
    frames = np.stack([synth.frame(1000 + f, 1920, 1080) for f in range(3)])
    edges, counts = gpu.canny_dev(frames)
    for f in range(3):
        ref, n = oracle.canny(frames[f])
        assert np.count_nonzero(edges[f] != ref) <= 1e-5 * ref.size
        assert int(counts[f]) == int(np.count_nonzero(edges[f]))


def test_fhog_tile_4096(gpu):
    """config 4 shape: one 4096x4096 RGB tile.  Exact against the oracle on the whole tile (the restatement runs in ~1 s)."""
    tile = synth.frame_rgb(3, 4096, 4096)
    got = gpu.fhog_dev(tile[None], 8, 1, 1)[0]
    assert got.shape == (510, 510, 31)
    ref = oracle.fhog(tile, 8, 1, 1)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    assert np.all(got >= 0) and np.all(got[..., :27] <= 0.4 + 1e-6)  # each feature sums 4 terms clipped at 0.2*|.| * 0.5
    # translation by whole cells: cells whose 3x3-cell support lies inside both crops agree exactly
    sub = np.ascontiguousarray(tile[64:64 + 1024, 128:128 + 1024])
    g2 = gpu.fhog(sub, 8, 1, 1)
    assert np.array_equal(g2[2:-2, 2:-2].view(np.uint32), got[8 + 2:8 + 126 - 2, 16 + 2:16 + 126 - 2].view(np.uint32))


def test_fhog_tile_4096_gray_replicated(gpu):
    """config 4's second input variant: a gray tile replicated into the three channels -- every channel ties in
    the gradient pick (fhog.h:821-845 keeps the first of equal magnitudes); whole tile against the oracle"""
    gray = synth.frame(4, 4096, 4096)
    tile = np.ascontiguousarray(np.repeat(gray[:, :, None], 3, axis=2))
    got = gpu.fhog_dev(tile[None], 8, 1, 1)[0]
    ref = oracle.fhog(tile, 8, 1, 1)
    assert got.shape == ref.shape == (510, 510, 31) and np.array_equal(got.view(np.uint32), ref.view(np.uint32))


def test_surf_tile_4096(gpu):
    """config 4 shape: interest points of a 4096x4096 tile, exact against the oracle (int32/f64 arithmetic)"""
    from test_surf import blobs
    tile = np.tile(blobs(5, 1024, 1024), (4, 4, 1))
    tile[::7, ::5] //= 2  # break the 4x4 periodicity
    got = gpu.surf_interest_points(tile, 30.0)
    ref = oracle.surf_interest_points(tile, 30.0)
    assert len(ref) > 200 and got.shape == ref.shape and np.array_equal(got, ref)
    for max_points in (1000, 10000):   # both caps (10000 keeps every point of this tile)
        s = gpu.surf(tile, max_points, 30.0)
        r = oracle.surf(tile, max_points, 30.0)
        assert 0 < len(r["x"]) <= min(max_points, len(ref))   # (points too close to the border for a descriptor are dropped)
        for k in r:
            assert s[k].shape == r[k].shape and np.array_equal(s[k], r[k]), (k, max_points)


def test_surf_bright_tile_whose_integral_image_wraps(gpu):
    """a 4096x4096 tile with a mean gray level of ~240: the int32 integral image wraps (sum 4.0e9 > 2^31) long before the last
    row, as dlib's does (integral_image.h:33-62 with promote<uchar> = int32).  Box sums are differences of four table entries,
    so the wrap cancels and every determinant -- formed from ONE int32 difference per Dxx / Dyy since round 5 -- equals dlib's."""
    from test_surf import blobs
    tile = (255 - np.tile(blobs(6, 1024, 1024), (4, 4, 1)) // 3).astype(np.uint8)
    tile[::5, ::9] -= 20
    gray = tile.astype(np.uint64).sum(2) // 3
    assert int(gray.sum()) > 2 ** 31 + 2 ** 30                      # it does wrap
    got = gpu.surf_interest_points(tile, 5.0)
    ref = oracle.surf_interest_points(tile, 5.0)
    assert len(ref) > 100 and got.shape == ref.shape and np.array_equal(got, ref)


def test_config5_stream_of_4k_frames_sampled_against_the_oracle():
    """config 5 shape: frames G(50000+f) generated on the device, Harris defaults + FAST-9 + Canny defaults through
    imgfd_detect_dev (what bench.py times); per-frame counts, and a sample frame checked in full against the oracle"""
    import torch
    from image_amd.device import DeviceDetector
    det = DeviceDetector(0)
    det.ctx.set_fir_mode(0)
    n, sample = 8, 5
    frames = det.synth_frames(n, NX, NY, seed0=50000)
    corners = torch.zeros((n, 16384, 3), dtype=torch.float32, device="cuda")
    points = torch.zeros((n, 16384, 2), dtype=torch.int32, device="cuda")
    edges = torch.zeros((n, NY, NX), dtype=torch.uint8, device="cuda")
    counts = torch.zeros((3, n), dtype=torch.int64, device="cuda")
    det.detect_all(frames, corners, points, edges, counts, fast9_threshold=20, suppress_non_max=1)
    det.ctx.sync()
    host = synth.frame(50000 + sample, NX, NY)
    assert np.array_equal(frames[sample].cpu().numpy(), host)          # device and host generators agree
    ref_h = oracle.harris(host.astype(np.float32))
    ref_f = oracle.fast9(host, 20, True)
    ref_e, ref_n = oracle.canny(host)
    c = counts.cpu().numpy()
    assert (c[0, sample], c[1, sample], c[2, sample]) == (len(ref_h), len(ref_f), ref_n)
    assert np.array_equal(bits(corners[sample, :len(ref_h)].cpu().numpy()), bits(ref_h))
    assert np.array_equal(points[sample, :len(ref_f)].cpu().numpy(), ref_f)
    assert np.array_equal(edges[sample].cpu().numpy(), ref_e)
    assert np.all(c > 0)                                               # every frame of the stream produced features


def test_one_8192x6000_frame_through_the_batch_entry_points(gpu):
    """index arithmetic beyond 2^25 pixels per frame (49 Mpixel: mask words, tile counts, union-find labels, candidate lists):
    FAST-9, Harris (strict) and Canny on one frame against the oracle"""
    nx, ny = 8192, 6000
    img = synth.frame(77, nx, ny, n_rect=900)
    gpu.set_fir_mode(0)
    pts, pc = gpu.fast9_dev(img[None], 20, True)
    ref = oracle.fast9(img, 20, True)
    assert int(pc[0]) == len(ref) and np.array_equal(pts[0], ref)
    lists, counts = gpu.harris_dev(img[None])
    ref = oracle.harris(img.astype(np.float32))
    assert int(counts[0]) == len(ref) > 1000 and np.array_equal(bits(lists[0]), bits(ref))
    edges, ec = gpu.canny_dev(img[None])
    ref, n = oracle.canny(img)
    assert int(np.count_nonzero(edges[0] != ref)) <= 1e-5 * ref.size and abs(int(ec[0]) - n) <= 1e-5 * ref.size
