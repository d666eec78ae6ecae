"""SURF: device stages (integral image, Hessian pyramid, maxima + interpolation) and the whole imgfd_surf against the
restatement oracle (bit-identical to dlib compiled in place, tests/test_oracle_dlib.py; SURF is unpinned upstream: no
dlib test covers surf.h / hessian_pyramid.h) and against golden vectors written by dlib's own code.
All arithmetic is int32 / f64 in the reference's operation order, so everything is compared exactly."""
import numpy as np
import pytest

import oracle
from image_amd import synth


def blobs(seed, w, h, n=60):
    """synthetic RGB frame with soft blobs of many sizes (the rectangle frames alone give few Hessian maxima)"""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    img = synth.frame_rgb(seed, w, h).astype(np.float64) * 0.35
    for _ in range(n):
        cx, cy, s = rng.uniform(0, w), rng.uniform(0, h), rng.uniform(2.5, 14)
        amp = rng.uniform(-120, 160, 3)
        img += amp[None, None, :] * np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * s * s))[..., None]
    return np.clip(img, 0, 255).astype(np.uint8)


@pytest.mark.parametrize("w,h", [(64, 48), (257, 129), (400, 300), (1031, 517), (1028, 70), (260, 300), (256, 256), (1284, 97), (2052, 33),
                                 (16, 4200), (20, 4127), (8192, 9), (8196, 8), (320, 205), (772, 96), (512, 129), (4100, 31), (64, 1025)])
def test_integral_image_exact(be, w, h):
    """widths that are multiples of 4 take the band / strip form (surf_int_sums / _carry / _apply: partial strips, partial
    bands, a single band, several workgroups per band; more than 128 bands: a wave of the carry kernel walks its run of bands in
    memory instead of holding it in registers; 8192 columns = the 32 strips the carry kernel's scan holds, 8196: one more, back to
    the row scan + column scan; a last band of one row, of 31; a last strip of four columns); the others the row scan + column scan"""
    rgb = synth.frame_rgb(81, w, h)
    assert np.array_equal(be.surf_integral(rgb), oracle.surf_integral(rgb))


def test_integral_image_wraps_like_int32(be):
    if be.name == "emu":
        pytest.skip("8.6 Mpixel image: too slow on the host emulator (the wrap is integer arithmetic, identical there)")
    rgb = np.full((2900, 2960, 3), 255, np.uint8)      # 255 * 2900 * 2960 > 2^31: the reference's int32 sums wrap
    a = be.surf_integral(rgb)
    assert a.min() < 0
    assert np.array_equal(a, oracle.surf_integral(rgb))


@pytest.mark.parametrize("seed,w,h,thr", [(82, 320, 240, 30.0), (83, 517, 389, 30.0), (84, 400, 300, 5.0), (85, 256, 256, 200.0)])
def test_interest_points_exact(be, seed, w, h, thr):
    rgb = blobs(seed, w, h)
    got, ref = be.surf_interest_points(rgb, thr), oracle.surf_interest_points(rgb, thr)
    assert len(ref) >= 3
    assert got.shape == ref.shape and np.array_equal(got, ref)


@pytest.mark.parametrize("w,h", [(1, 1), (9, 9), (30, 30), (65, 40), (100, 90)])
def test_small_images(be, w, h):
    rgb = blobs(86, max(w, 16), max(h, 16))[:h, :w]
    got, ref = be.surf_interest_points(rgb, 1.0), oracle.surf_interest_points(rgb, 1.0)
    assert got.shape == ref.shape and np.array_equal(got, ref)
    s = be.surf(rgb, 1000, 1.0)
    assert len(s["x"]) == len(oracle.surf(rgb, 1000, 1.0)["x"])


@pytest.mark.parametrize("max_points,thr", [(1000, 30.0), (50, 30.0), (10000, 5.0)])
def test_surf_points_and_descriptors_exact(be, max_points, thr):
    rgb = blobs(87, 480, 360)
    got, ref = be.surf(rgb, max_points, thr), oracle.surf(rgb, max_points, thr)
    assert len(ref["x"]) > 3
    for k in ("x", "y", "pyramid_scale", "score", "laplacian", "angle", "surf"):
        assert got[k].shape == ref[k].shape, k
        assert np.array_equal(got[k], ref[k]), k


@pytest.mark.parametrize("kind", ["zeros", "full", "noise", "checker8", "gray_ramp"])
def test_extreme_images(be, kind):
    """flat frames (no interest point at any threshold), uniform noise at threshold 0 (every local maximum of the determinant
    is a point: thousands on a small frame), an 8-pixel checkerboard (exact ties everywhere), a gray ramp: points and
    descriptors as the restated reference has them"""
    w, h = 200, 152
    rng = np.random.default_rng(6)
    ramp = np.clip(np.add.outer(np.arange(h), np.arange(w)) // 2, 0, 255).astype(np.uint8)
    rgb = {"zeros": np.zeros((h, w, 3), np.uint8), "full": np.full((h, w, 3), 255, np.uint8),
           "noise": rng.integers(0, 256, (h, w, 3), dtype=np.uint8),
           "checker8": np.repeat(((np.add.outer(np.arange(h) // 8, np.arange(w) // 8) & 1) * 255).astype(np.uint8)[:, :, None], 3, axis=2),
           "gray_ramp": np.repeat(ramp[:, :, None], 3, axis=2)}[kind]
    for thr, max_points in ((0.0, 10000), (30.0, 7), (1e9, 1000)):
        p, rp = be.surf_interest_points(rgb, thr), oracle.surf_interest_points(rgb, thr)
        assert p.shape == rp.shape and np.array_equal(p, rp), (kind, thr)
        got, ref = be.surf(rgb, max_points, thr), oracle.surf(rgb, max_points, thr)
        for k in ("x", "y", "pyramid_scale", "score", "laplacian", "angle", "surf"):
            assert got[k].shape == ref[k].shape and np.array_equal(got[k], ref[k], equal_nan=True), (kind, thr, k)


def test_batch_points_dev(be):
    frames = np.stack([blobs(95 + f, 256, 192) for f in range(3)])
    lists, counts = be.surf_points_dev(frames, 10.0)
    for f in range(3):
        ref = oracle.surf_interest_points(frames[f], 10.0)
        assert counts[f] == len(ref) and np.array_equal(lists[f], ref)


def test_batch_surf_dev_descriptors_on_the_device(be):
    """imgfd_surf_dev: K19 with the device libm.  Points are exact; angles and descriptors may differ from the
    reference in the last bits of atan2/sin/cos (SURVEY 8d asks for 1e-6; we hold 1e-9)."""
    frames = np.stack([blobs(120 + f, 320, 256) for f in range(3)])
    got = be.surf_dev(frames, max_points=200, threshold=10.0)
    for f in range(3):
        ref = oracle.surf(frames[f], 200, 10.0)
        assert len(ref["x"]) > 3
        for k in ("x", "y", "pyramid_scale", "score", "laplacian"):
            assert np.array_equal(got[f][k], ref[k]), k
        assert np.allclose(got[f]["angle"], ref["angle"], rtol=0, atol=1e-9)
        assert np.abs(got[f]["surf"] - ref["surf"]).max() <= 1e-9
    # cap below max_points truncates the ranked list
    few = be.surf_dev(frames[:1], max_points=200, threshold=10.0, cap=7)[0]
    assert len(few["x"]) <= 7 and np.array_equal(few["score"], np.sort(few["score"])[::-1])


@pytest.mark.parametrize("w,h", [(384, 256), (400, 304), (640, 272), (528, 162), (396, 260), (330, 200)])
def test_upper_octaves_in_both_table_layouts(be, w, h):
    """octaves 1-3 (points from more than one octave): widths that are a multiple of 16 keep the integral image in the residue
    layout only -- the first octave's LDS window, the buffer-load gathers of surf_pyramid_taps, the maximum test's look-ups for
    the intervals that are not built and K19 all address it --, the other widths (396, 330) the plain table with
    surf_pyramid_plain: the reference's interest points and descriptors, bit for bit"""
    img = blobs(170 + w, w, h)
    ref = oracle.surf_interest_points(img, 2.0)
    assert len(ref) > 20 and len({int(round(np.log2(p[2]))) for p in ref}) >= 2   # points from more than one octave
    got = be.surf_interest_points(img, 2.0)
    assert got.shape == ref.shape and np.array_equal(got.view(np.uint64), ref.view(np.uint64))
    full, rfull = be.surf(img, 300, 2.0), oracle.surf(img, 300, 2.0)
    for k in ("x", "y", "pyramid_scale", "score", "laplacian", "angle", "surf"):
        assert full[k].shape == rfull[k].shape and np.array_equal(full[k], rfull[k], equal_nan=True), k


@pytest.mark.parametrize("w,h", [(256, 200), (330, 170)])
def test_threshold_zero_lists_every_level_pixel(be, w, h):
    """detection threshold 0: every level pixel is marked, a workgroup of the maximum test hands its 16384 listed pixels out in 64
    trips of 256 (the list in LDS holds one trip); the reference's points, bit for bit"""
    img = blobs(250 + w, w, h)
    ref = oracle.surf_interest_points(img, 0.0)
    got = be.surf_interest_points(img, 0.0)
    assert len(ref) > 50 and got.shape == ref.shape and np.array_equal(got.view(np.uint64), ref.view(np.uint64))


def test_single_tile_calls_reuse_their_buffers(be):
    """a call with one tile runs octaves 1-3 on the companion context's stream beside octave 0 and the maximum test waits for
    both; repeated calls reuse the buffers (the second tile's integral image must not overtake the first tile's gather kernel)"""
    frames = np.stack([blobs(230 + f, 320, 240) for f in range(2)])
    ref = [oracle.surf(frames[f], 300, 4.0) for f in range(2)]
    pts = [oracle.surf_interest_points(frames[f], 4.0) for f in range(2)]
    for rep in range(2):
        for f in (0, 1):
            got = be.surf_dev(frames[f:f + 1], max_points=300, threshold=4.0)[0]
            assert len(ref[f]["x"]) > 20
            assert np.array_equal(got["score"], ref[f]["score"]), (rep, f)
            key = lambda d: sorted(zip(d["score"], d["x"], d["y"], d["pyramid_scale"], d["laplacian"]))  # (equal scores may swap places)
            assert key(got) == key(ref[f]), (rep, f)
            p = be.surf_interest_points(frames[f], 4.0)
            assert p.shape == pts[f].shape and np.array_equal(p.view(np.uint64), pts[f].view(np.uint64)), (rep, f)


@pytest.mark.parametrize("max_points", [1, 9, 25])
@pytest.mark.parametrize("sort_cap", [2048, 48, 4])
def test_surf_dev_ranks_and_cuts_on_the_device(be, max_points, sort_cap):
    """imgfd_surf_dev orders the candidates on the device (radix select of the max_points best, rank, box test, compaction):
    with fewer slots than candidates the strongest survive, in the reference's order.  sort_cap 2048: every candidate of these
    frames goes straight into the LDS sort; 48: the radix select runs until winners + open candidates fit 48 slots, then the
    sort ranks both together (what a 4096^2 tile does at 2048); 4: the select runs to the end, all-pairs ranking"""
    frames = np.stack([blobs(140 + f, 384, 256) for f in range(2)])
    try:
        be.set_tuning("surf_sort_cap", sort_cap)
        got = be.surf_dev(frames, max_points=max_points, threshold=5.0)
    finally:
        be.set_tuning("surf_sort_cap", 2048)
    for f in range(2):
        allp = oracle.surf_interest_points(frames[f], 5.0)
        assert len(allp) > max_points                           # the cut really cuts
        ref = oracle.surf(frames[f], max_points, 5.0)
        assert np.array_equal(got[f]["score"], ref["score"])    # the same scores in the same (descending) order
        sc = np.sort(allp[:, 3])[::-1]
        if sc[max_points - 1] != sc[max_points]:                # no exact tie across the cut: the same points (ties inside may swap)
            key = lambda d: sorted(zip(d["score"], d["x"], d["y"], d["pyramid_scale"], d["laplacian"]))
            assert key(got[f]) == key(ref)
        untied = np.ones(len(ref["score"]), bool)
        untied[1:] &= ref["score"][1:] != ref["score"][:-1]; untied[:-1] &= ref["score"][:-1] != ref["score"][1:]
        for k in ("x", "y", "pyramid_scale", "laplacian"):
            assert np.array_equal(got[f][k][untied], ref[k][untied]), (k, max_points)
        if untied.any():
            assert np.abs(got[f]["surf"][untied] - ref["surf"][untied]).max() <= 1e-9


def test_surf_dev_redoes_tiles_whose_candidates_overflow(be):
    """a tile with more candidates than the record buffer holds (lab switch surf_rec_cap lowers the buffer to 16 records)
    reports -candidates (imgfd_surf_dev never waits for the host); imgfd_surf_dev_redo redoes it with a larger buffer: same
    features as with room for everything"""
    frames = np.stack([blobs(160 + f, 320, 224) for f in range(3)])
    ref = be.surf_dev(frames, max_points=20, threshold=5.0)
    ncand = [len(oracle.surf_interest_points(frames[f], 5.0)) for f in range(3)]
    assert min(ncand) > 20
    try:
        be.set_tuning("surf_rec_cap", 16)
        got = be.surf_dev(frames, max_points=20, threshold=5.0)
        for f in range(3):
            assert len(got[f]["x"]) == len(ref[f]["x"]) > 0
            for k in ("x", "y", "score", "pyramid_scale", "laplacian", "angle", "surf"):
                assert np.array_equal(got[f][k], ref[f][k]), (f, k)
        assert be.last_surf_redone == 3
        raw = be.surf_dev_counts(frames, max_points=20, threshold=5.0).tolist()   # -(room to redo the tile with): survivors of the screening >= candidates
        assert all(-r >= c for r, c in zip(raw, ncand)) and all(-r < 4 * c for r, c in zip(raw, ncand))
    finally:
        be.set_tuning("surf_rec_cap", 1 << 18)


@pytest.mark.parametrize("group,lanes", [(1, 1), (2, 1), (3, 2), (8, 2), (4, 4), (16, 3)])
def test_surf_dev_groups_of_tiles(be, group, lanes):
    """imgfd_surf_dev handles the tiles in groups ("surf_group"): the fronts of a group (integral image, pyramid) go round-robin
    over "surf_lanes" streams, each tile into a buffer set of its own, and the back stages (maximum test, ranking, orientation,
    descriptor) are ONE launch each for the whole group (blockIdx.y = tile).  Seven tiles of different content: full groups, a
    short last group, a group larger than the batch, more lanes than tiles in the last group; a buffer set read too early or
    a wrong stride between the sets shows as another tile's points.  Twice: the second call reuses the sets while nothing
    of the first is waited for in between."""
    frames = np.stack([blobs(300 + f, 320, 240, n=25 + 9 * f) for f in range(7)])
    ref = [oracle.surf(frames[f], 150, 3.0) for f in range(7)]
    try:
        be.set_tuning("surf_group", group); be.set_tuning("surf_lanes", lanes)
        for _ in range(2):
            got = be.surf_dev(frames, max_points=150, threshold=3.0)
            for f in range(7):
                assert len(ref[f]["x"]) > 5
                for k in ("x", "y", "pyramid_scale", "score", "laplacian"):
                    assert np.array_equal(got[f][k], ref[f][k]), (group, lanes, f, k)
                assert np.abs(got[f]["surf"] - ref[f]["surf"]).max() <= 1e-9
    finally:
        be.set_tuning("surf_group", 4); be.set_tuning("surf_lanes", 3)


def test_surf_dev_exact_score_ties_keep_emission_order(be):
    """two identical blobs give pairs of exactly equal scores: the device breaks the tie by emission order (what a stable
    sort would do); the set of points is the reference's either way"""
    one = blobs(150, 192, 160)
    img = np.concatenate([one, one], axis=1)                    # the same content twice, side by side
    got = be.surf_dev(img[None], max_points=400, threshold=5.0)[0]
    ref = oracle.surf(img, 400, 5.0)
    assert len(got["x"]) == len(ref["x"]) > 4
    assert np.array_equal(got["score"], ref["score"])           # same multiset, same (descending) order of scores
    key = lambda d: sorted(zip(d["score"], d["x"], d["y"]))
    assert key(got) == key(ref)
    s = got["score"]
    tied = np.nonzero(s[1:] == s[:-1])[0]
    assert len(tied) > 0
    # emission order within a tie: octave/interval first (equal here), then row, then column -> the left copy first
    assert all(got["x"][i] < got["x"][i + 1] or got["y"][i] < got["y"][i + 1] for i in tied)


def test_golden_dlib(be, golden):
    """vectors written by dlib's own get_surf_points on the reference's example image"""
    g = golden("surf_cruise_boat")
    got = be.surf(g["image"], 1000, 30.0)
    for k in ("x", "y", "angle", "pyramid_scale", "score", "laplacian", "surf"):
        assert got[k].shape == g[k].shape and np.allclose(got[k], g[k], rtol=1e-9, atol=1e-12), k
        assert np.array_equal(got[k], g[k]), k


def test_r_level_mirror(be):
    if be.name != "gpu":
        pytest.skip("image_amd.api binds the product library")
    from image_amd import api
    rgb = blobs(88, 300, 220)
    x = rgb.transpose(2, 1, 0).astype(np.int32)   # (3, width, height)
    out = api.image_surf(x, max_points=200, detection_threshold=30)
    ref = oracle.surf(rgb, 200, 30.0)
    assert out["points"] == len(ref["x"]) and out["surf"].shape == (out["points"], 64)
    assert np.array_equal(out["x"], ref["x"]) and np.array_equal(out["surf"], np.nan_to_num(ref["surf"]))
