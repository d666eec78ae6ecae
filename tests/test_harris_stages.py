"""Plane-by-plane parity of the Harris device stages against the oracle (bit-exact in strict mode)."""
import numpy as np
import pytest

import oracle
from image_amd import synth


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def assert_bits_equal(a, b, what):
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    assert a.shape == b.shape, what
    bad = bits(a) != bits(b)
    assert not bad.any(), f"{what}: {bad.sum()} of {bad.size} values differ, first at {np.argwhere(bad)[0]}, " \
                          f"max rel {np.max(np.abs(a - b) / np.maximum(1e-30, np.abs(b)))}"


SIZES = [(64, 48), (200, 150), (331, 257), (640, 480)]


@pytest.mark.parametrize("nx,ny", SIZES)
def test_smoothing_sigma1_bit_exact(be, nx, ny):
    img = synth.frame(11, nx, ny).astype(np.float32)
    be.set_fir_mode(0)
    got = be.k_gaussian(img, 1.0, 0)
    assert_bits_equal(got, oracle.harris_stage("gaussian", img, sigma=1.0, type=0), "discrete_gaussian sigma 1")


@pytest.mark.parametrize("sigma", [0.5, 1.25, 2.0, 2.5, 4.0])
def test_gaussian_other_sigmas_bit_exact(be, sigma):
    img = synth.frame(12, 150, 97).astype(np.float32)
    be.set_fir_mode(0)
    got = be.k_gaussian(img, sigma, 0)
    assert_bits_equal(got, oracle.harris_stage("gaussian", img, sigma=sigma, type=0), f"discrete_gaussian {sigma}")


@pytest.mark.parametrize("sigma", [1.0, 2.5, 0.3, 6.0])
@pytest.mark.parametrize("nx,ny", [(64, 48), (201, 77), (70, 130)])
def test_sii_gaussian_bit_exact(be, nx, ny, sigma):
    """gaussian code 1: stacked integral images, gaussian.cpp:61-281 (sequential float prefix sums)"""
    img = synth.frame(20, nx, ny).astype(np.float32)
    got = be.k_gaussian(img, sigma, 1)
    assert_bits_equal(got, oracle.harris_stage("gaussian", img, sigma=sigma, type=1), f"SII gaussian {sigma}")


@pytest.mark.parametrize("gauss", [1, 2])
def test_structure_tensor_sii_bit_exact(be, gauss):
    ix, iy = _gradients(21, 200, 150)
    got = be.k_structure_tensor(ix, iy, 2.5, gauss)
    ref = oracle.harris_stage("autocorrelation", ix, iy, sigma=2.5, gauss=gauss)
    for g, r, nm in zip(got, ref, "ABC"):
        assert_bits_equal(g, r, f"structure tensor {nm} gauss {gauss}")


def test_gaussian_copy_cases(be):
    img = synth.frame(13, 40, 30).astype(np.float32)
    assert_bits_equal(be.k_gaussian(img, 1.0, 2), img, "NO_GAUSSIAN is a copy")
    assert_bits_equal(be.k_gaussian(img, 0.0, 0), img, "sigma<=0 is a copy")
    tiny = synth.frame(13, 5, 30).astype(np.float32)
    assert_bits_equal(be.k_gaussian(tiny, 2.5, 0), tiny, "size>xdim leaves the image untouched")


@pytest.mark.parametrize("type", [0, 1])
@pytest.mark.parametrize("nx,ny", [(3, 3), (64, 48), (201, 77)])
def test_gradient_bit_exact(be, nx, ny, type):
    img = oracle.harris_stage("gaussian", synth.frame(14, nx, ny).astype(np.float32), sigma=1.0, type=0)
    ix, iy = be.k_gradient(img, type)
    rx, ry = oracle.harris_stage("gradient", img, type=type)
    assert_bits_equal(ix, rx, "Ix")
    assert_bits_equal(iy, ry, "Iy")


@pytest.mark.parametrize("type", [0, 1])
@pytest.mark.parametrize("nx,ny", SIZES + [(4, 5), (5, 70), (64, 61), (67, 130), (129, 16), (260, 200)])
def test_fused_gauss_grad_u8_strict_bit_exact(be, nx, ny, type):
    """The kernel the batch path runs on u8 frames (marching strips: segments of rows, 15-row steps, border strips, strips
    that start on the last column) against the two reference stages, bit for bit in strict mode."""
    be.set_fir_mode(0)
    img = synth.frame(23, nx, ny)
    ix, iy = be.k_gauss_grad_u8(img, 1.0, type)
    sm = oracle.harris_stage("gaussian", img.astype(np.float32), sigma=1.0, type=0)
    rx, ry = oracle.harris_stage("gradient", sm, type=type)
    assert_bits_equal(ix, rx, f"Ix {nx}x{ny}")
    assert_bits_equal(iy, ry, f"Iy {nx}x{ny}")


@pytest.mark.parametrize("type", [0, 1])
@pytest.mark.parametrize("nx,ny,seg", [(256, 16, 0), (256, 41, 0), (496, 33, 10), (480, 100, 24), (512, 17, 16), (736, 130, 0), (736, 130, 43),
                                       (240 * 3 + 16, 66, 33), (1040, 50, 49)])
def test_marching_gauss_grad_u8_bit_exact(be, nx, ny, seg, type):
    """gauss_grad_march (u8 frames whose width is a multiple of 16): strips of 240 columns -- first, interior, a last strip of
    16 columns --, segments of every length class (one chunk, a last segment of two rows, the top / bottom clamp of the
    gradient inside a segment), against the two reference stages bit for bit in strict mode; with fused accumulation the
    tile kernel's bits"""
    img = synth.frame(29, nx, ny)
    sm = oracle.harris_stage("gaussian", img.astype(np.float32), sigma=1.0, type=0)
    rx, ry = oracle.harris_stage("gradient", sm, type=type)
    try:
        be.set_tuning("gauss_march_seg", seg)
        be.set_fir_mode(0)
        n0 = be.get_counter("gauss_march_launches")
        ix, iy = be.k_gauss_grad_u8(img, 1.0, type)
        assert be.get_counter("gauss_march_launches") == n0 + 1, "the marching kernel did not run"
        assert_bits_equal(ix, rx, f"Ix {nx}x{ny}")
        assert_bits_equal(iy, ry, f"Iy {nx}x{ny}")
        be.set_fir_mode(1)
        ix, iy = be.k_gauss_grad_u8(img, 1.0, type)
        be.set_tuning("gauss_march", 0)
        tx, ty = be.k_gauss_grad_u8(img, 1.0, type)
        assert_bits_equal(ix, tx, "Ix, fused accumulation"); assert_bits_equal(iy, ty, "Iy, fused accumulation")
    finally:
        be.set_tuning("gauss_march", 1); be.set_tuning("gauss_march_seg", 0); be.set_fir_mode(0)


def test_marching_gauss_grad_random_shapes(be):
    """twenty random shapes the marching kernel serves (widths 256..1264 in steps of 16, heights 16..160, random segment
    lengths, both gradient types): the tile kernel's bits in both accumulation modes"""
    rng = np.random.default_rng(77)
    try:
        for _ in range(20):
            nx, ny = 16 * int(rng.integers(16, 80)), int(rng.integers(16, 161))
            seg, typ, mode = int(rng.choice([0, 2, 9, 24, 40, 57])), int(rng.integers(0, 2)), int(rng.integers(0, 2))
            img = rng.integers(0, 256, (ny, nx), dtype=np.uint8)
            be.set_fir_mode(mode)
            be.set_tuning("gauss_march", 1); be.set_tuning("gauss_march_seg", seg)
            n0 = be.get_counter("gauss_march_launches")
            ix, iy = be.k_gauss_grad_u8(img, 1.0, typ)
            assert be.get_counter("gauss_march_launches") == n0 + 1
            be.set_tuning("gauss_march", 0)
            tx, ty = be.k_gauss_grad_u8(img, 1.0, typ)
            assert_bits_equal(ix, tx, f"Ix {nx}x{ny} seg {seg} type {typ} mode {mode}")
            assert_bits_equal(iy, ty, f"Iy {nx}x{ny} seg {seg} type {typ} mode {mode}")
    finally:
        be.set_tuning("gauss_march", 1); be.set_tuning("gauss_march_seg", 0); be.set_fir_mode(0)


def _gradients(seed, nx, ny):
    img = oracle.harris_stage("gaussian", synth.frame(seed, nx, ny).astype(np.float32), sigma=1.0, type=0)
    return oracle.harris_stage("gradient", img, type=0)


@pytest.mark.parametrize("nx,ny", SIZES)
def test_structure_tensor_strict_bit_exact(be, nx, ny):
    ix, iy = _gradients(15, nx, ny)
    be.set_fir_mode(0)
    got = be.k_structure_tensor(ix, iy, 2.5, 0)
    ref = oracle.harris_stage("autocorrelation", ix, iy, sigma=2.5, gauss=0)
    for g, r, nm in zip(got, ref, "ABC"):
        assert_bits_equal(g, r, f"structure tensor {nm} (strict)")


@pytest.mark.parametrize("sigma", [0.625, 1.25, 3.0])
def test_structure_tensor_other_sigmas(be, sigma):
    ix, iy = _gradients(16, 160, 120)
    be.set_fir_mode(0)
    got = be.k_structure_tensor(ix, iy, sigma, 0)
    ref = oracle.harris_stage("autocorrelation", ix, iy, sigma=sigma, gauss=0)
    for g, r, nm in zip(got, ref, "ABC"):
        assert_bits_equal(g, r, f"structure tensor {nm} sigma {sigma}")


@pytest.mark.parametrize("nx,ny", [(128, 16), (132, 40), (256, 31), (388, 100), (8, 70), (1004, 37)])
@pytest.mark.parametrize("sigma", [2.5, 1.25, 0.625])
def test_tensor_response_kernel_bit_exact(be, nx, ny, sigma):
    """K3 with the response epilogue (the kernel the batch path runs on image_harris() defaults): strips that end in the
    middle of a tile, single-strip images, segments shorter than a chunk; strict mode is bit-exact"""
    if nx < int(3 * sigma) + 1:
        pytest.skip("kernel wider than the image: the reference skips the smoothing")
    ix, iy = _gradients(23, nx, ny)
    be.set_fir_mode(0)
    A, B, Cc = oracle.harris_stage("autocorrelation", ix, iy, sigma=sigma, gauss=0)
    ref = oracle.harris_stage("response", A, B, Cc, measure=0, k=0.06)
    assert_bits_equal(be.k_tensor_response(ix, iy, sigma, 0.06), ref, f"tensor+response {nx}x{ny} sigma {sigma}")


@pytest.mark.parametrize("nx,ny", [(130, 33), (67, 20), (259, 50)])
def test_structure_tensor_rows_that_are_no_whole_quads(be, nx, ny):
    """nx % 4 != 0: the element-load instance of the marching kernel"""
    ix, iy = _gradients(24, nx, ny)
    be.set_fir_mode(0)
    got = be.k_structure_tensor(ix, iy, 2.5, 0)
    ref = oracle.harris_stage("autocorrelation", ix, iy, sigma=2.5, gauss=0)
    for g, r, nm in zip(got, ref, "ABC"):
        assert_bits_equal(g, r, f"structure tensor {nm} {nx}x{ny}")


@pytest.mark.parametrize("workers", [1, 3, 7, 11])
@pytest.mark.parametrize("out", ["abc", "response"])
def test_tensor_kernel_workers_walk_several_tiles(be, workers, out):
    """the structure-tensor workgroups are persistent: the batch's (frame, strip) columns form ONE line of 16-row chunk units
    and a worker marches its equal share of it as one pipelined sequence.  520 x 77 with sigma 2.5: 3 strips x 6 units = 18
    units -- 7 and 11 workers get 3 and 2 units each, so shares begin and end in the middle of a column (a warm-up chunk
    where a share begins), 3 workers get one whole column each, one worker walks all three."""
    try:
        be.set_tuning("tensor_workers", workers)
        nx, ny = 520, 77
        ix, iy = _gradients(26, nx, ny)
        be.set_fir_mode(0)
        A, B, Cc = oracle.harris_stage("autocorrelation", ix, iy, sigma=2.5, gauss=0)
        if out == "abc":
            for g, r, nm in zip(be.k_structure_tensor(ix, iy, 2.5, 0), (A, B, Cc), "ABC"):
                assert_bits_equal(g, r, f"structure tensor {nm}, {workers} workers")
        else:
            ref = oracle.harris_stage("response", A, B, Cc, measure=0, k=0.06)
            assert_bits_equal(be.k_tensor_response(ix, iy, 2.5, 0.06), ref, f"tensor+response, {workers} workers")
    finally:
        be.set_tuning("tensor_workers", 0)


def test_tensor_response_unsupported_shapes_are_refused(be):
    ix, iy = _gradients(25, 130, 20)
    with pytest.raises(Exception, match="imgfd_status 5"):
        be.k_tensor_response(ix, iy, 2.5)


def test_structure_tensor_fused_accumulate_within_tolerance(be):
    """fir_mode 1 (fma inside the f64 accumulation): north_star tolerance is 1e-4 relative; the
    planes differ from strict by at most 1 float ulp and almost nowhere."""
    ix, iy = _gradients(17, 640, 480)
    be.set_fir_mode(1)
    got = be.k_structure_tensor(ix, iy, 2.5, 0)
    ref = oracle.harris_stage("autocorrelation", ix, iy, sigma=2.5, gauss=0)
    be.set_fir_mode(0)
    total_bad = 0
    for g, r in zip(got, ref):
        rel = np.abs(g - r) / np.maximum(np.abs(r), 1e-30)
        assert rel.max() <= 1.2e-7, rel.max()          # <= 1 ulp
        total_bad += int((bits(g) != bits(r)).sum())
    assert total_bad <= 5, total_bad                      # expected ~0.1 per frame of this size


@pytest.mark.parametrize("measure", [0, 1, 2])
def test_response_bit_exact(be, measure):
    ix, iy = _gradients(18, 200, 150)
    A, B, Cc = oracle.harris_stage("autocorrelation", ix, iy, sigma=2.5, gauss=0)
    got = be.k_response(A, B, Cc, measure, 0.06)
    assert_bits_equal(got, oracle.harris_stage("response", A, B, Cc, measure=measure, k=0.06), f"response {measure}")


@pytest.mark.parametrize("quads", [False, True], ids=["tiled", "quads"])
@pytest.mark.parametrize("radius,Th", [(5, 130.0), (1, 10.0), (3, 1000.0), (8, 0.5)])
def test_nms_window_rule_matches_scanline(be, radius, Th, quads):
    ix, iy = _gradients(19, 320, 240)
    A, B, Cc = oracle.harris_stage("autocorrelation", ix, iy, sigma=2.5, gauss=0)
    R = oracle.harris_stage("response", A, B, Cc, measure=0, k=0.06)
    got = be.k_nms(R, Th, radius, quads=quads)
    ref = oracle.harris_stage("nms", R, Th=Th, radius=radius)
    assert got.shape == ref.shape, (got.shape, ref.shape)
    assert np.array_equal(bits(got), bits(ref))


@pytest.mark.parametrize("quads", [False, True], ids=["tiled", "quads"])
def test_nms_start_of_row_rule_on_exact_ties(be, quads):
    """harris.cpp:177 skips 'the downhill at the beginning' of every row: a window maximum whose left neighbour TIES it and
    that lies in the initial non-increasing run (from column `radius`) is never emitted, although the window rule accepts
    it.  Planes with exact plateaus at the start of rows, in the middle of rows (where the tie does not block) and below
    the threshold."""
    rng = np.random.default_rng(5)
    radius, Th = 3, 100.0
    R = (rng.random((40, 200)) * 50).astype(np.float32)           # background below the threshold
    R[10, 2:9] = 500.0                                              # plateau from column radius-1: initial run, blocked
    R[14, 3:6] = 400.0                                              # plateau starting at column radius itself
    R[18, 0:5] = 300.0; R[18, 5] = 350.0                            # a rise after the plateau: column 5 is a proper peak
    R[22, 60:64] = 450.0                                            # the same tie in mid-row, after the scan line got going
    R[26, 3:40] = np.linspace(900, 200, 37).astype(np.float32)      # a long downhill from the start, then ...
    R[26, 50:53] = 600.0                                            # ... a tie plateau further along the row
    R[30, 2] = 120.0; R[30, 3] = 120.0                              # tie exactly at column radius with column radius-1
    got = be.k_nms(R, Th, radius, quads=quads)
    ref = oracle.harris_stage("nms", R, Th=Th, radius=radius)
    assert len(ref) >= 2 and got.shape == ref.shape and np.array_equal(bits(got), bits(ref)), (got, ref)
    Rt = np.ascontiguousarray(np.tile(R, (3, 3)))                   # larger plane: several 64x64 tiles of the fused kernel
    ref = oracle.harris_stage("nms", Rt, Th=Th, radius=radius)
    got = be.k_nms(Rt, Th, radius, quads=quads)
    assert got.shape == ref.shape and np.array_equal(bits(got), bits(ref))


@pytest.mark.parametrize("radius", [1, 5])
def test_nms_dense_candidates(be, radius):
    """every pixel at or above the threshold (noise, Th = 0): a workgroup's 256 mask words carry far more candidate bits
    than its list holds (4096), so the sparse kernel takes them wave by wave -- same corners as the scan line"""
    R = (np.random.default_rng(11).random((96, 640)) * 1000 + 1).astype(np.float32)
    ref = oracle.harris_stage("nms", R, Th=0.0, radius=radius)
    assert len(ref) > 50
    for quads in (True, False):
        got = be.k_nms(R, 0.0, radius, quads=quads)
        assert got.shape == ref.shape and np.array_equal(bits(got), bits(ref)), quads


def test_nms_small_image_is_empty(be):
    R = np.random.default_rng(0).random((11, 40)).astype(np.float32) * 1000
    assert be.k_nms(R, 0.0, 5).shape[0] == 0
    assert be.k_nms(R, 0.0, 5, quads=True).shape[0] == 0


def test_nms_quads_plateaus(be):
    """The quad byte drops a pixel that a neighbour inside its quad beats; plateaus (ties inside a quad and across two) and
    peaks on every column residue must come out as the scan line has them.  (Plateaus that tie a plateau in the row above
    are left out: the scan line's skip marks of earlier rows, which then reach into the start-of-row rule, are not modelled
    by either NMS kernel -- LOG.md, round-5 section 4.)"""
    rng = np.random.default_rng(11)
    R = (rng.random((64, 132)) * 90).astype(np.float32)
    for i, c in enumerate(range(8, 120, 7)):          # isolated peaks on columns of every residue mod 4
        R[6 + (i * 5) % 50, c] = 500.0 + i
    R[40, 20:24] = 700.0; R[57, 98:103] = 720.0       # plateaus inside one quad and across two
    R[50, 65:67] = 650.0; R[54, 63:65] = 640.0         # two-pixel ties in a quad's middle and across a quad (and word) border
    for radius in (1, 2, 5):
        ref = oracle.harris_stage("nms", R, Th=100.0, radius=radius)
        for quads in (False, True):
            got = be.k_nms(R, 100.0, radius, quads=quads)
            assert got.shape == ref.shape and np.array_equal(bits(got), bits(ref)), (radius, quads, len(got), len(ref))


_PLATEAU_ROWS = [
    [3, 3, 4, 1, 3, 1, 3, 3, 1, 2, 3, 4, 1, 3, 3, 2, 1, 3, 4, 1, 4, 2, 2, 2, 1], [3, 3, 1, 4, 3, 2, 2, 3, 4, 4, 1, 2, 1, 3, 4, 3, 4, 3, 3, 3, 3, 2, 2, 1, 3],
    [1, 4, 2, 1, 3, 4, 4, 1, 4, 4, 3, 1, 1, 3, 3, 4, 1, 4, 1, 3, 2, 4, 4, 2, 1], [2, 4, 1, 3, 4, 2, 4, 3, 2, 4, 2, 3, 3, 4, 1, 2, 2, 2, 3, 2, 3, 2, 3, 2, 1],
    [4, 3, 4, 1, 1, 3, 1, 4, 2, 2, 4, 4, 2, 1, 3, 4, 1, 1, 2, 4, 2, 2, 1, 2, 4], [3, 4, 3, 1, 1, 3, 2, 4, 1, 3, 4, 3, 2, 4, 3, 3, 3, 4, 2, 1, 2, 4, 4, 3, 4],
    [1, 4, 1, 1, 1, 2, 1, 1, 2, 4, 3, 2, 3, 2, 2, 4, 4, 2, 2, 4, 2, 3, 4, 4, 3], [2, 2, 1, 2, 1, 2, 1, 3, 4, 1, 2, 3, 1, 2, 2, 4, 3, 1, 3, 4, 2, 2, 2, 1, 1],
    [2, 3, 4, 2, 2, 4, 4, 1, 3, 2, 4, 2, 2, 3, 4, 3, 4, 1, 4, 1, 3, 1, 1, 2, 4], [4, 4, 2, 4, 2, 3, 3, 4, 2, 3, 1, 4, 3, 2, 4, 4, 2, 4, 2, 3, 4, 4, 4, 3, 4],
    [1, 4, 2, 4, 2, 4, 1, 3, 1, 4, 2, 4, 4, 2, 4, 3, 4, 4, 2, 1, 1, 1, 4, 2, 3]]


# The reference's row loop is an OpenMP parallel-for (harris.cpp:169-172; src/Makevars builds with SHLIB_OPENMP_CXXFLAGS), and a
# candidate marks pixels of the rows BELOW it as skipped while it scans its window (:218); those marks steer the scan of the
# later rows (:175-177, :181).  What a row sees of them depends on the thread schedule -- with exact ties the reference has
# no single answer.  Two schedules are deterministic: one thread (every mark of the rows above is seen: oracle orc_nms) and
# rows independent (no row sees another row's marks: orc_nms_rows_independent).  The device computes the second.
_TIED_PLANE = [[1, 4, 1, 1, 4, 2, 3], [4, 1, 4, 3, 3, 2, 2], [3, 1, 4, 4, 4, 1, 2], [3, 3, 4, 1, 4, 3, 2]]


@pytest.mark.xfail(strict=True, reason="KNOWN DIVERGENCE FROM THE ONE-THREAD SCHEDULE, exact ties only: the rejected candidate (x=2, y=1) marks (1..3, 2) as skipped "
                                       "(harris.cpp:218), which extends row 2's 'downhill at the beginning' (:175-177) over the plateau 4 4 4: one thread emits no corner, "
                                       "the rows-independent schedule -- and nms.hip -- emit the plateau's right end (4, 2).  Real responses are floats of a smoothed "
                                       "image: exact ties in adjacent rows do not occur (0 differences on every fixture and at 4K).")
def test_nms_skip_marks_carried_across_rows_reference_divergence(be):
    R = np.asarray(_TIED_PLANE, np.float32) * 100
    ref = oracle.harris_stage("nms", R, Th=150.0, radius=1)   # the one-thread schedule (pinned to the compiled reference below)
    got = be.k_nms(R, 150.0, 1)
    assert got.shape == ref.shape and np.array_equal(bits(got), bits(ref))


@pytest.mark.skipif(not oracle.have_ref("harris"), reason="oracle/_ref not built")
def test_nms_tied_plane_one_thread_reference_equals_its_restatement():
    """the plane of the xfail above through the reference compiled in place (an OpenMP build: pinned to one thread for the
    call) and through the restatement of that schedule: no corner; the rows-independent schedule: one"""
    R = np.asarray(_TIED_PLANE, np.float32) * 100
    L = oracle.ref("harris")
    L.ref_set_threads(1)
    try:
        ref = oracle.harris_stage("nms", R, Th=150.0, radius=1, use_ref=True)
    finally:
        L.ref_set_threads(L.ref_max_threads())
    seq = oracle.harris_stage("nms", R, Th=150.0, radius=1)
    assert len(ref) == 0 and len(seq) == 0
    ind = oracle.harris_stage("nms", R, Th=150.0, radius=1, rows_independent=True)
    assert ind.tolist() == [[4.0, 2.0, 400.0]]


def test_nms_tied_plane_on_which_the_openmp_reference_races(be):
    """four integer levels, radius 1: both deterministic schedules give 18 corners, and so does the device; the reference
    compiled here with OpenMP returned 17 on about one call in eight at 8 and 16 threads (a schedule in between: some marks
    seen, some not) -- found by random search against it in round 4, which is how the race was noticed"""
    R = np.asarray(_PLATEAU_ROWS, np.float32) * 100
    seq = oracle.harris_stage("nms", R, Th=150.0, radius=1)
    ind = oracle.harris_stage("nms", R, Th=150.0, radius=1, rows_independent=True)
    assert len(seq) == 18 and np.array_equal(bits(seq), bits(ind))
    got = be.k_nms(R, 150.0, 1)
    assert got.shape == seq.shape and np.array_equal(bits(got), bits(seq))


@pytest.mark.parametrize("radius", [1, 2, 5])
def test_nms_on_tied_planes_is_the_reference_with_rows_independent(be, radius):
    """the device computes the schedule in which no row sees another row's marks -- bit for bit, on planes made of a few integer
    levels (ties everywhere) -- and the one-thread schedule differs from it on some of the same planes (the strict xfail above
    pins one)"""
    rng = np.random.default_rng(900 + radius)
    differ = 0
    for case in range(40):
        nx, ny = int(rng.integers(2 * radius + 2, 70)), int(rng.integers(2 * radius + 2, 40))
        if case % 2: nx = (nx + 3) // 4 * 4
        R = (rng.integers(0, 4 + case % 3, size=(ny, nx)) * 100).astype(np.float32)
        ind = oracle.harris_stage("nms", R, Th=150.0, radius=radius, rows_independent=True)
        seq = oracle.harris_stage("nms", R, Th=150.0, radius=radius)
        got = be.k_nms(R, 150.0, radius)
        assert got.shape == ind.shape and np.array_equal(bits(got), bits(ind)), (case, nx, ny)
        if nx % 4 == 0:   # the batch path's kernel (threshold quads -> sparse NMS) takes rows of whole quads
            got = be.k_nms(R, 150.0, radius, quads=True)
            assert got.shape == ind.shape and np.array_equal(bits(got), bits(ind)), (case, nx, ny, "quads")
        differ += int(seq.shape != ind.shape or not np.array_equal(seq, ind))
    if radius == 1:
        assert differ > 0   # the schedules do differ on such planes: the test exercises the tie cases
