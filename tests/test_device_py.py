"""image_amd.device.DeviceDetector (torch tensors in, torch tensors out): the wrapper bench.py drives.  detect_all (two HIP
streams, imgfd_detect_dev) must return exactly what the three separate calls return, and those what the oracle says."""
import numpy as np
import pytest

import oracle
from image_amd import synth

pytestmark = pytest.mark.gpu


def test_detect_all_equals_the_separate_calls_and_the_oracle():
    import torch
    from image_amd.device import DeviceDetector
    det = DeviceDetector(0)
    det.ctx.set_fir_mode(0)
    nx, ny, n = 448, 280, 5
    host = np.stack([synth.frame(400 + f, nx, ny, n_rect=25) for f in range(n)])
    frames = torch.from_numpy(host).cuda()
    corners, hc = det.harris(frames, cap=4096, threshold=50.0)
    points, fc = det.fast9(frames, threshold=15, suppress_non_max=True, cap=8192)
    edges, cc = det.canny(frames)
    c2 = torch.zeros_like(corners); p2 = torch.zeros_like(points); e2 = torch.zeros_like(edges)
    counts = torch.zeros((3, n), dtype=torch.int64, device="cuda")
    for rep in (0, 1):                                   # twice over (buffers reused, the companion context warm)
        c2.zero_(); p2.zero_(); e2.zero_(); counts.zero_()
        for _ in range(2):                               # twice: the companion context is created on the first call
            det.detect_all(frames, c2, p2, e2, counts, threshold=50.0, fast9_threshold=15, suppress_non_max=1)
        det.ctx.sync()
        assert torch.equal(counts[0], hc) and torch.equal(counts[1], fc) and torch.equal(counts[2], cc)
        assert torch.equal(e2, edges)
        for f in range(n):
            k, m = int(hc[f]), int(fc[f])
            assert torch.equal(c2[f, :k], corners[f, :k]) and torch.equal(p2[f, :m], points[f, :m])
    for f in range(n):
        k, m = int(hc[f]), int(fc[f])
        ref = oracle.harris(host[f].astype(np.float32), threshold=50.0)
        assert np.array_equal(corners[f, :k].cpu().numpy().view(np.uint32), ref.view(np.uint32))
        assert np.array_equal(points[f, :m].cpu().numpy(), oracle.fast9(host[f], 15, True))
        assert np.array_equal(edges[f].cpu().numpy(), oracle.canny(host[f])[0])


def test_small_batches_repeat_with_new_contents():
    """a repeating imgfd_detect_dev call on fewer than 8 frames (Canny's chain on the context's own stream, the other detectors on
    the companion's, their launches queued behind Canny's last one): same buffers, new frame CONTENTS every call, a call with other
    arguments in between -- every call's outputs equal the separate entry points' on that content"""
    import torch
    from image_amd.device import DeviceDetector
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        det = DeviceDetector(0)
        nx, ny, n = 448, 280, 3
        frames = torch.empty((n, ny, nx), dtype=torch.uint8, device="cuda")
        corners = torch.zeros((n, 4096, 3), dtype=torch.float32, device="cuda"); points = torch.zeros((n, 8192, 2), dtype=torch.int32, device="cuda")
        edges = torch.zeros((n, ny, nx), dtype=torch.uint8, device="cuda"); counts = torch.zeros((3, n), dtype=torch.int64, device="cuda")
        other = torch.zeros((3, n), dtype=torch.int64, device="cuda")
        for it in range(7):
            host = np.stack([synth.frame(500 + 10 * it + f, nx, ny, n_rect=25) for f in range(n)])
            frames.copy_(torch.from_numpy(host).cuda())
            if it == 4:
                det.detect_all(frames, corners, points, edges, other, threshold=60.0, fast9_threshold=15, suppress_non_max=1)
            det.detect_all(frames, corners, points, edges, counts, threshold=50.0, fast9_threshold=15, suppress_non_max=1)
            det.ctx.sync()
            got = (counts.clone(), corners.clone(), points.clone(), edges.clone())
            c1, hc = det.harris(frames, cap=4096, threshold=50.0)
            p1, fc = det.fast9(frames, threshold=15, suppress_non_max=True, cap=8192)
            e1, cc = det.canny(frames)
            det.ctx.sync()
            assert torch.equal(got[0][0], hc) and torch.equal(got[0][1], fc) and torch.equal(got[0][2], cc), it
            assert torch.equal(got[3], e1), it
            for f in range(n):
                k, m = int(hc[f]), int(fc[f])
                assert k > 0 and m > 0
                assert torch.equal(got[1][f, :k], c1[f, :k]) and torch.equal(got[2][f, :m], p1[f, :m]), (it, f)


def test_synth_frames_match_the_host_generator():
    from image_amd.device import DeviceDetector
    det = DeviceDetector(0)
    fr = det.synth_frames(3, 320, 200, seed0=77)
    for f in range(3):
        assert np.array_equal(fr[f].cpu().numpy(), synth.frame(77 + f, 320, 200))
