"""The wave-autonomous structure-tensor kernel (image_amd/csrc/fir_tensor_wave.hip: every wave marches its own 64-column
strip through all three planes, no workgroup barrier) against the oracle -- bit for bit in strict mode -- and against the
workgroup-marching kernel of fir_tensor.hip that it stands in for.  Reference: compute_autocorrelation_matrix(),
image.CornerDetectionHarris/src/harris.cpp:44-70; compute_corner_response() :78-133 (Harris measure)."""
import numpy as np
import pytest

import oracle
from image_amd import synth
from test_harris_stages import _gradients, assert_bits_equal

# strips that end mid-tile, a single strip, one strip + one quad, last strips of 4 / 60 valid columns, many strips
SHAPES = [(64, 16), (64, 48), (68, 40), (124, 33), (128, 16), (256, 31), (388, 100), (1004, 37), (640, 480)]


def _wave(be, on):
    be.set_tuning("tensor_wave", 1 if on else 0)


@pytest.fixture
def wave(be):
    _wave(be, True)
    yield be
    _wave(be, True)
    be.set_tuning("tensor_workers", 0); be.set_tuning("tensor_seg", 0)


@pytest.mark.parametrize("nx,ny", SHAPES)
def test_abc_strict_bit_exact(wave, nx, ny):
    be = wave
    ix, iy = _gradients(31, nx, ny)
    be.set_fir_mode(0)
    n0 = be.get_counter("tensor_wave_launches")
    got = be.k_structure_tensor(ix, iy, 2.5, 0)
    assert be.get_counter("tensor_wave_launches") == n0 + 1, "the wave kernel did not run"
    ref = oracle.harris_stage("autocorrelation", ix, iy, sigma=2.5, gauss=0)
    for g, r, nm in zip(got, ref, "ABC"):
        assert_bits_equal(g, r, f"wave structure tensor {nm} {nx}x{ny}")


@pytest.mark.parametrize("sigma", [1.25, 0.625])  # radius 3 and 1
@pytest.mark.parametrize("nx,ny", [(64, 20), (200, 150), (332, 57)])
def test_abc_other_radii(wave, nx, ny, sigma):
    be = wave
    ix, iy = _gradients(32, nx, ny)
    be.set_fir_mode(0)
    n0 = be.get_counter("tensor_wave_launches")
    got = be.k_structure_tensor(ix, iy, sigma, 0)
    assert be.get_counter("tensor_wave_launches") == n0 + 1
    ref = oracle.harris_stage("autocorrelation", ix, iy, sigma=sigma, gauss=0)
    for g, r, nm in zip(got, ref, "ABC"):
        assert_bits_equal(g, r, f"wave structure tensor {nm} sigma {sigma} {nx}x{ny}")


@pytest.mark.parametrize("sigma", [2.5, 1.25, 0.625])
@pytest.mark.parametrize("nx,ny", [(64, 16), (132, 40), (256, 31), (388, 100), (1004, 37)])
def test_response_strict_bit_exact(wave, nx, ny, sigma):
    be = wave
    ix, iy = _gradients(33, nx, ny)
    be.set_fir_mode(0)
    A, B, Cc = oracle.harris_stage("autocorrelation", ix, iy, sigma=sigma, gauss=0)
    ref = oracle.harris_stage("response", A, B, Cc, measure=0, k=0.06)
    n0 = be.get_counter("tensor_wave_launches")
    got = be.k_tensor_response(ix, iy, sigma, 0.06)
    assert be.get_counter("tensor_wave_launches") == n0 + 1
    assert_bits_equal(got, ref, f"wave tensor+response {nx}x{ny} sigma {sigma}")


@pytest.mark.parametrize("workers", [1, 3, 8, 11])
@pytest.mark.parametrize("out", ["abc", "response"])
def test_waves_walk_several_tiles(wave, workers, out):
    """few workers (= waves) and short segments: every wave crosses tile boundaries (strip change, segment change, the last
    short segment), workgroups hold waves with and without work"""
    be = wave
    be.set_tuning("tensor_workers", workers); be.set_tuning("tensor_seg", 18)
    nx, ny = 520, 77
    ix, iy = _gradients(34, nx, ny)
    be.set_fir_mode(0)
    A, B, Cc = oracle.harris_stage("autocorrelation", ix, iy, sigma=2.5, gauss=0)
    if out == "abc":
        for g, r, nm in zip(be.k_structure_tensor(ix, iy, 2.5, 0), (A, B, Cc), "ABC"):
            assert_bits_equal(g, r, f"wave structure tensor {nm}, {workers} workers")
    else:
        ref = oracle.harris_stage("response", A, B, Cc, measure=0, k=0.06)
        assert_bits_equal(be.k_tensor_response(ix, iy, 2.5, 0.06), ref, f"wave tensor+response, {workers} workers")


def test_fused_accumulate_equals_the_workgroup_kernel(wave):
    """fir_mode 1 (fma inside the f64 accumulation): both kernels issue the same chains, so they agree bit for bit"""
    be = wave
    ix, iy = _gradients(35, 328, 90)
    try:
        be.set_fir_mode(1)
        a = be.k_structure_tensor(ix, iy, 2.5, 0)
        ra = be.k_tensor_response(ix, iy, 2.5, 0.06)
        _wave(be, False)
        n0 = be.get_counter("tensor_wave_launches")
        b = be.k_structure_tensor(ix, iy, 2.5, 0)
        rb = be.k_tensor_response(ix, iy, 2.5, 0.06)
        assert be.get_counter("tensor_wave_launches") == n0, "tensor_wave 0 must select the workgroup kernel"
        for g, r, nm in zip(a, b, "ABC"):
            assert_bits_equal(g, r, f"fma mode {nm}")
        assert_bits_equal(ra, rb, "fma mode response")
    finally:
        be.set_fir_mode(0)


@pytest.mark.parametrize("nx,ny,n", [(128, 96, 3), (320, 240, 2), (708, 64, 2)])
def test_batch_path_corners_equal(wave, nx, ny, n):
    """imgfd_harris_dev (threshold quads from the response epilogue -> sparse NMS): the corner lists of a batch are those of
    the oracle, and the same through both kernels"""
    be = wave
    frames = np.stack([synth.frame(40 + f, nx, ny) for f in range(n)])
    be.set_fir_mode(0)
    n0 = be.get_counter("tensor_wave_launches")
    got, cnt = be.harris_dev(frames, threshold=130.0)
    assert be.get_counter("tensor_wave_launches") > n0
    _wave(be, False)
    old, cnt_old = be.harris_dev(frames, threshold=130.0)
    for f in range(n):
        ref = oracle.harris(frames[f].astype(np.float32))
        assert int(cnt[f]) == len(ref) == int(cnt_old[f])
        assert np.array_equal(got[f], ref), f"frame {f}"
        assert np.array_equal(old[f], ref), f"frame {f} (workgroup kernel)"
