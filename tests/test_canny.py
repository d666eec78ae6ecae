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


@pytest.mark.parametrize("s", [12.0, 16.0])
def test_large_s_takes_the_taps_from_memory(be, s):
    """tools.c:146-185 blurs with any s; beyond 129 kept taps (s above ~10) the device reads the wrapped kernel from memory
    instead of a kernel argument.  320x240 against the restatement and, on a crop, against the reference's own sources
    (oracle/_ref/libref_canny.so; its stand-in DFT for FFTW3 is O(n^3))"""
    img = synth.frame(23, 320, 240)
    edges, n = be.canny(img, s=s, low_thr=1.0, high_thr=2.0)
    ref, rn = oracle.canny(img, s=s, low_thr=1.0, high_thr=2.0)
    assert n == int(np.count_nonzero(edges)) and rn > 50
    assert mismatch(edges, ref) <= max(2, 1e-3 * edges.size)
    if oracle.have_ref("canny"):
        small = img[:96, :128]
        e2, n2 = be.canny(small, s=s, low_thr=1.0, high_thr=2.0)
        r2, rn2 = oracle.ref_canny(small, s=s, low_thr=1.0, high_thr=2.0)
        assert mismatch(e2, r2) <= max(2, 1e-3 * e2.size)


@pytest.mark.parametrize("s", [0.5, 0.7])
@pytest.mark.parametrize("acc", [0, 1])
def test_small_s_on_sparse_impulses(be, s, acc):
    """isolated bright pixels under a narrow Gaussian: neighbouring blurred values lie dozens of binades apart, where the
    gradient kernel's shared row terms (d(y), e(y): a regrouping of rcpp_canny.cpp:157-163) are no longer sums of exactly
    representable partials -- the edge map must still be the reference's"""
    rng = np.random.default_rng(5)
    img = np.zeros((96, 192), np.uint8)
    ys, xs = rng.integers(4, 92, 60), rng.integers(4, 188, 60)
    img[ys, xs] = rng.integers(40, 256, 60)
    img[40:44, 100:140] = 255                      # and one solid bar
    for low, high in ((0, 1), (1, 3), (3, 10)):
        edges, n = be.canny(img, s=s, low_thr=low, high_thr=high, accGrad=bool(acc))
        ref, rn = oracle.canny(img, s=s, low_thr=low, high_thr=high, accGrad=bool(acc))
        assert mismatch(edges, ref) == 0 and n == rn, (s, acc, low, high, mismatch(edges, ref))


def test_thresholds_are_int_truncated(be):
    img = synth.frame(22, 120, 90)
    a, _ = be.canny(img, low_thr=3.0, high_thr=10.0)
    b, _ = be.canny(img, low_thr=3.9, high_thr=10.9)
    assert np.array_equal(a, b)


@pytest.mark.parametrize("low,high", [(0, 0), (3, 3), (10, 3), (-5, 2), (-3.7, -1.2), (0.9, 0.9), (200, 300), (0, 1e9)])
def test_unusual_thresholds_follow_the_reference(be, low, high):
    """thresholds the R interface does not forbid: equal, crossed (low > high), negative (truncated toward zero like the int
    cast of rcpp_canny.cpp:180), zero, beyond every magnitude -- `now <= low` rejects, `now >= high` marks strong (:98-104)"""
    img = synth.frame(23, 150, 110)
    edges, n = be.canny(img, low_thr=low, high_thr=high)
    ref, rn = oracle.canny(img, low_thr=low, high_thr=high)
    assert n == rn and np.array_equal(edges, ref), (low, high, n, rn)


@pytest.mark.parametrize("kind", ["zeros", "full", "checker1", "checker8", "noise", "ramp", "one_pixel"])
def test_extreme_images(be, kind):
    """flat frames (every gradient zero: atan2(0, 0), magnitude 0 <= low), one- and eight-pixel checkerboards (exact ties in the
    non-maximum test), uniform noise (edge pixels everywhere: the densest hysteresis), a ramp (constant gradient), a single
    bright pixel"""
    w, h = 168, 120
    rng = np.random.default_rng(8)
    one = np.zeros((h, w), np.uint8); one[h // 2, w // 3] = 255
    img = {"zeros": np.zeros((h, w), np.uint8), "full": np.full((h, w), 255, np.uint8),
           "checker1": ((np.add.outer(np.arange(h), np.arange(w)) & 1) * 255).astype(np.uint8),
           "checker8": ((np.add.outer(np.arange(h) // 8, np.arange(w) // 8) & 1) * 255).astype(np.uint8),
           "noise": rng.integers(0, 256, (h, w)).astype(np.uint8),
           "ramp": np.clip(np.add.outer(np.arange(h), np.arange(w)), 0, 255).astype(np.uint8), "one_pixel": one}[kind]
    for kw in (dict(), dict(accGrad=False), dict(s=0.8, low_thr=0, high_thr=1), dict(s=4.0, low_thr=1, high_thr=2)):
        edges, n = be.canny(img, **kw)
        ref, rn = oracle.canny(img, **kw)
        if kind != "ramp":
            assert n == rn and np.array_equal(edges, ref), (kind, kw, n, rn, int(np.count_nonzero(edges != ref)))
            continue
        # The ramp's gradient is constant, and at the frame border clamp-to-edge hands the non-maximum test its own pixel
        # as a neighbour: `now` against an interpolation of values equal to `now`, decided by the last bit of
        # (1 - a) g + a g.  The reference gets `a` from cos / sin(atan2(v, h)), the device from the unit vector (SURVEY
        # Appendix A: ~1e-16 apart, "only exact-tie comparisons can flip") -- this frame is where they do: at most two
        # pixels, on the border (the restatement and the compiled reference agree with each other here, tests/test_oracle.py).
        ys, xs = np.nonzero(edges != ref)
        assert len(ys) <= 2 and abs(n - rn) <= 2, (kw, len(ys))
        assert all(y in (0, h - 1) or x in (0, w - 1) for y, x in zip(ys, xs)), list(zip(ys, xs))


def test_hysteresis_long_chain(be):
    """a weak edge across many 64-pixel words must light up from its strong left end: the contrast of a (slightly
    slanted) step decays smoothly along x -- one unbroken line whose left end alone is above the high threshold.  The step
    has an intermediate row (a two-level step makes neighbouring rows tie exactly in the non-maximum test, and ties are
    where the device's unit-vector form of atan2 -> cos/sin may legitimately flip a pixel)."""
    yy, xx = np.mgrid[0:200, 0:300]
    t = (100 + xx / 37.0).astype(int)
    c = 6 + 90 * np.exp(-xx / 8.0)
    img = np.round(60 + (yy > t) * c + (yy == t) * c * 0.3).astype(np.uint8)
    edges, n = be.canny(img, **SERP_KW)
    ref, rn, dbg = oracle.canny(img, debug=True, **SERP_KW)
    assert np.count_nonzero(dbg["nms"] == 2) < 300 and rn > 1000        # ~200 strong pixels light ~1100
    assert (ref > 0)[:, 150:290].any(axis=0).all() and not (dbg["nms"][:, 150:290] == 2).any()   # weak there, yet lit
    assert mismatch(edges, ref) == 0 and n == rn


def _serpentine(nx=640, ny=400):
    """one weak edge that snakes up and down across the whole image, lit from its far left end only: the region enclosed
    by the snake stands out from the background by a contrast that decays smoothly along x (strong for x < ~30, weak
    beyond -- a junction with a separate strong blob would not do: NMS cuts the weak line next to it), with a rim at 30 %
    of it (no exact ties, see above).  The hysteresis has to carry the seed through every part of the frame, upwards and
    downwards."""
    inside = np.zeros((ny, nx), bool)
    x, up = 10, True
    while x + 60 < nx:
        inside[16:ny - 16, x:x + 14] = True
        if up: inside[16:30, x:x + 54] = True
        else: inside[ny - 30:ny - 16, x:x + 54] = True
        up = not up
        x += 40
    core = inside.copy()
    core[1:] &= inside[:-1]; core[:-1] &= inside[1:]; core[:, 1:] &= inside[:, :-1]; core[:, :-1] &= inside[:, 1:]
    contrast = (7 + 90 * np.exp(-np.arange(nx) / 6.0))[None, :]
    return np.round(60 + core * contrast + (inside & ~core) * 0.3 * contrast).astype(np.uint8)


SERP_KW = dict(s=1.0, low_thr=5, high_thr=60)


def test_serpentine_really_propagates():
    img = _serpentine()
    edges, n, dbg = oracle.canny(img, debug=True, **SERP_KW)
    strong, marked = np.count_nonzero(dbg["nms"] == 2), np.count_nonzero(dbg["nms"] >= 1)
    cols = np.nonzero(edges.any(axis=0))[0]
    assert strong < 600 and n > 0.9 * marked > 8000 and cols.max() > 600   # a few hundred seeds light ~10 000 pixels


@pytest.mark.parametrize("sweeps,nx,ny", [(14, 384, 200), (3, 384, 200), (1, 320, 160), (1, 200, 100), (2, 1000, 260), (1, 64, 64), (1, 130, 70)])
def test_hysteresis_device_side_termination(be, sweeps, nx, ny):
    """The hysteresis never reports to the host: a fixed number of sweeps is queued and a union-find kernel completes whatever
    they left -- runs of still-unlit marked pixels united across rows and words, lit when their root touches a strong pixel.
    Few sweeps force that kernel to do most of the work, on chains that cross many tiles, word boundaries and rows in both
    directions; widths that are no multiple of 64 exercise the last, partial word of a row."""
    if be.name == "emu" and nx >= 1000:
        pytest.skip("512 fibers per workgroup x a 1000 x 260 frame: half a minute on the emulator; the GPU run covers it")
    try:
        be.set_tuning("hyst_sweeps", int(sweeps))
        img = _serpentine(nx, ny)
        kw = SERP_KW
        ref, rn, dbg = oracle.canny(img, debug=True, **kw)
        assert rn > 3 * np.count_nonzero(dbg["nms"] == 2) or nx < 100        # most of what is lit was only marked
        edges, n = be.canny(img, **kw)
        assert n == rn and mismatch(edges, ref) == 0, (sweeps, n, rn)
        frames = np.stack([img, synth.frame(31, nx, ny), img[::-1].copy()])
        e, c = be.canny_dev(frames, **kw)
        for f in range(3):
            r, k = oracle.canny(frames[f], **kw)
            assert c[f] == k and mismatch(e[f], r) <= (0 if f != 1 else 3)
    finally:
        be.set_tuning("hyst_sweeps", 0)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_union_find_on_noise(be, seed):
    """random frames with thresholds that mark a third of the pixels and make few of them strong: dense, branching components
    of every shape, one sweep queued -- the union-find kernel decides nearly everything"""
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 256, size=(150, 333), dtype=np.uint8)
    kw = dict(s=1.2, low_thr=2, high_thr=40)
    try:
        be.set_tuning("hyst_sweeps", 1)
        ref, rn = oracle.canny(img, **kw)
        edges, n = be.canny(img, **kw)
        assert n == rn and mismatch(edges, ref) == 0, (n, rn)
    finally:
        be.set_tuning("hyst_sweeps", 0)


@pytest.mark.parametrize("n_frames", [3, 14])
def test_hysteresis_both_sweep_kernels(be, n_frames):
    """up to a dozen frames the sweeps walk one-word tiles in blocks of 2 x 4 per workgroup, larger batches four-word tiles in blocks
    of 2 x 2; the tiles of a block exchange their outlines through LDS inside one launch, odd launches group the tiles half a block
    up and left.  The fixpoint is unique, so either kernel and any number of queued sweeps (the union-find part completes what they
    leave) gives the oracle's edge map -- on the serpentine (one chain through every tile, both directions, block outlines
    included: 1000 x 260 spans several blocks) and on a synthetic frame."""
    if be.name == "emu" and n_frames > 3:
        pytest.skip("14 frames of 512-fiber workgroups: minutes on the emulator; the GPU run covers the batch kernel")
    img = _serpentine(1000, 260) if be.name != "emu" else _serpentine(520, 200)
    kinds = [img, synth.frame(45, img.shape[1], img.shape[0]), img[::-1, ::-1].copy()]
    frames = np.stack([kinds[f % 3] for f in range(n_frames)])
    refs = [oracle.canny(k, **SERP_KW) for k in kinds]
    try:
        for sweeps in ((0, 1, 3) if be.name != "emu" else (2,)):
            be.set_tuning("hyst_sweeps", sweeps)
            e, c = be.canny_dev(frames, **SERP_KW)
            for f in range(n_frames):
                r, k = refs[f % 3]
                assert c[f] == k and mismatch(e[f], r) <= (0 if f % 3 != 1 else 3), (n_frames, sweeps, f)
            if n_frames == 3:
                edges, n = be.canny(img, **SERP_KW)
                assert n == refs[0][1] and mismatch(edges, refs[0][0]) == 0, sweeps
    finally:
        be.set_tuning("hyst_sweeps", 0)


def test_batch_dev(be):
    frames = np.stack([synth.frame(300 + f, 160, 96) for f in range(3)])
    edges, counts = be.canny_dev(frames)
    for f in range(3):
        ref, rn = oracle.canny(frames[f])
        assert mismatch(edges[f], ref) <= 3
        assert counts[f] == np.count_nonzero(edges[f])


def test_many_unconverged_frames_in_one_batch(be):
    """canny_finish completes frames the queued sweeps did not finish with two barriers across the workgroups of THAT frame inside one
    launch.  48 frames, ONE sweep queued, every frame left to it: 48 x 128 workgroups -- three times what the chip holds at once --
    must neither hang (workgroups are dispatched in order; a frame's workgroups wait for nobody else) nor differ from the oracle."""
    nx, ny = 384, 200
    base = _serpentine(nx, ny)
    frames = np.stack([np.roll(base, 7 * f, axis=1) if f % 3 else base[::-1].copy() for f in range(48 if be.name != "emu" else 3)])
    try:
        be.set_tuning("hyst_sweeps", 1)
        e, c = be.canny_dev(frames, **SERP_KW)
        if be.name != "emu":
            assert be.get_counter("canny_frames_unconverged") == len(frames)      # all of them took the union-find path
        seen = {}
        for f in range(len(frames)):
            key = frames[f].tobytes()
            if key not in seen: seen[key] = oracle.canny(frames[f], **SERP_KW)
            r, k = seen[key]
            assert c[f] == k and mismatch(e[f], r) == 0, f
    finally:
        be.set_tuning("hyst_sweeps", 0)


@pytest.mark.gpu
@pytest.mark.parametrize("cus", [32, 4, 1])
def test_unconverged_frames_on_a_cu_masked_stream(cus):
    """a stream that may use only `cus` of the device's compute units (hipExtStreamCreateWithCUMask: 1/8 of an MI355X, 4 units, one),
    with the Harris chain of another context queued in front.  48 frames the one queued sweep leaves unfinished.  canny_finish --
    the kernel whose workgroups wait for each other at frame barriers -- is launched only on streams that may use the whole
    device (canny_finish_blocks asks the stream for its CU mask); a masked stream takes the three launches without any barrier.
    Must neither hang nor differ from the oracle."""
    import ctypes as C

    import torch
    from backends import GpuBackend
    hip = C.CDLL("libamdhip64.so")
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count
    words = (n_cu + 31) // 32
    mask = (C.c_uint32 * words)()
    for i in range(cus):
        mask[i >> 5] |= 1 << (i & 31)
    stream = C.c_void_p()
    assert hip.hipExtStreamCreateWithCUMask(C.byref(stream), words, mask) == 0
    masked, other = GpuBackend(stream), GpuBackend()
    nx, ny = 384, 200
    base = _serpentine(nx, ny)
    frames = np.stack([np.roll(base, 7 * f, axis=1) if f % 3 else base[::-1].copy() for f in range(48)])
    big = np.stack([synth.frame(900 + f, 1920, 1080) for f in range(8)])
    try:
        masked.set_tuning("hyst_sweeps", 1)
        d_big, fr_big = other.upload_frames_u8(big)
        cnt_big = other.empty((8,), np.int64)
        for _ in range(6):   # ~10 ms of structure-tensor launches on every compute unit, queued before the masked call and running beside it
            other.check(other.lib.imgfd_harris_dev(other.ctx, C.byref(fr_big), 0.06, 1.0, 2.5, 130.0, 0, 0, 0, None, 0, other.ptr(cnt_big)), "harris_dev")
        e, c = masked.canny_dev(frames, **SERP_KW)
        assert masked.get_counter("canny_frames_unconverged") == len(frames)
        other.sync()
        seen = {}
        for f in range(len(frames)):
            key = frames[f].tobytes()
            if key not in seen: seen[key] = oracle.canny(frames[f], **SERP_KW)
            r, k = seen[key]
            assert c[f] == k and mismatch(e[f], r) == 0, (cus, f)
    finally:
        masked.close(); other.close()
        hip.hipStreamDestroy(stream)


@pytest.mark.gpu
def test_small_batches_queue_the_sweeps_recent_calls_needed():
    """up to 12 frames per call: canny_finish reports through pinned memory how many sweeps the frames needed, and the next calls on
    the context queue that many and one more instead of eight (an idle launch is 5-6 us of a single frame's chain).  Same edges
    every time; the number queued falls from 8 to working + 1; a frame that suddenly needs more than were queued (the serpentine:
    dozens of sweeps) is finished by the union-find part -- still the oracle's edges -- and the calls after it queue more again;
    another frame size starts from 8."""
    from backends import GpuBackend
    be = GpuBackend()
    try:
        img = synth.frame(77, 1920, 1080)
        ref, rn = oracle.canny(img)
        queued = []
        for _ in range(6):
            e, c = be.canny_dev(img[None])
            assert c[0] == rn and mismatch(e[0], ref) == 0
            queued.append(be.get_counter("canny_sweeps_queued"))
        working = be.get_counter("canny_sweeps_working")
        assert queued[0] == 8 and queued[-1] == min(8, max(2, working + 1)) and queued[-1] <= queued[0], (queued, working)
        serp = _serpentine(1920, 1080)
        rs, ks = oracle.canny(serp, **SERP_KW)
        e, c = be.canny_dev(serp[None], **SERP_KW)      # needs far more sweeps than the few that are queued now
        assert c[0] == ks and mismatch(e[0], rs) == 0
        assert be.get_counter("canny_frames_unconverged") == 1
        e, c = be.canny_dev(serp[None], **SERP_KW)
        assert c[0] == ks and mismatch(e[0], rs) == 0 and be.get_counter("canny_sweeps_queued") == 8   # the call before said "not finished": all the sweeps there are
        small = synth.frame(78, 640, 480)
        e, c = be.canny_dev(small[None])
        r2, k2 = oracle.canny(small)
        assert c[0] == k2 and mismatch(e[0], r2) == 0 and be.get_counter("canny_sweeps_queued") == 8        # a new size: nothing known yet
    finally:
        be.close()
