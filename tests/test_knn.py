"""Descriptor matching (imgfd_knn): the exact k-nearest-neighbour query the reference's README runs
with FNN::get.knnx on two images' SURF matrices (image.dlib/README.md:19-37).  FNN is not available here, so the oracle
restates its contract and is itself checked against an independent exact search (scipy's cKDTree); the device result is
compared with the oracle bit for bit (same summation order, same tie rule)."""
import ctypes as C

import numpy as np
import pytest

import oracle


def unit_rows(rng, n, dim=64):
    a = rng.standard_normal((n, dim))
    return a / np.linalg.norm(a, axis=1, keepdims=True)      # SURF descriptors are unit vectors


def test_oracle_agrees_with_an_independent_exact_search():
    from scipy.spatial import cKDTree
    rng = np.random.default_rng(5)
    data, query = unit_rows(rng, 700), unit_rows(rng, 300)
    idx, dist = oracle.knn(data, query, 3)
    d2, i2 = cKDTree(data).query(query, k=3)
    assert np.array_equal(idx, i2) and np.allclose(dist, d2, rtol=1e-13, atol=0)
    assert np.all(np.diff(dist, axis=1) >= 0)


@pytest.mark.parametrize("nd,nq,dim,k", [(500, 333, 64, 1), (500, 333, 64, 2), (1000, 70, 64, 8), (129, 65, 5, 3),
                                          (64, 8, 64, 1), (3, 9, 64, 1), (1, 1, 1, 1),
                                          (200, 8200, 64, 2)])   # >= 8192 queries: the four-queries-per-wave kernel
def test_knn_matches_oracle_exactly(be, nd, nq, dim, k):
    rng = np.random.default_rng(nd * 7 + nq)
    data, query = unit_rows(rng, nd, dim), unit_rows(rng, nq, dim)
    ref_i, ref_d = oracle.knn(data, query, k)
    for cm in (False, True):
        got_i, got_d = be.knn(data, query, k, column_major=cm)
        assert np.array_equal(got_i, ref_i), cm
        assert np.array_equal(got_d, ref_d), cm


def test_ties_and_short_data(be):
    rng = np.random.default_rng(11)
    base = unit_rows(rng, 40)
    data = np.concatenate([base, base[::-1], base[:5]])       # every row appears at least twice
    query = base[:17]
    ref_i, ref_d = oracle.knn(data, query, 4)
    got_i, got_d = be.knn(data, query, 4)
    assert np.array_equal(got_i, ref_i) and np.array_equal(got_d, ref_d)
    assert np.all(got_d[:, 0] == 0) and np.all(got_i[:, 0] == np.arange(17))   # the earliest copy wins the tie
    # fewer data rows than k: the tail is (-1, inf)
    got_i, got_d = be.knn(data[:2], query, 4)
    ref_i, ref_d = oracle.knn(data[:2], query, 4)
    assert np.array_equal(got_i, ref_i) and np.array_equal(got_d, ref_d)
    assert np.all(got_i[:, 2:] == -1) and np.all(np.isinf(got_d[:, 2:]))


def test_strided_device_form_matches_feature_records(be):
    """imgfd_knn_dev on the 70-double records imgfd_surf_dev writes: descriptor = columns 6..69"""
    rng = np.random.default_rng(23)
    a, b = unit_rows(rng, 150), unit_rows(rng, 90)
    ra = np.zeros((150, 70)); ra[:, 6:] = a; ra[:, :6] = rng.standard_normal((150, 6)) * 100
    rb = np.zeros((90, 70)); rb[:, 6:] = b
    ref_i, ref_d = oracle.knn(a, b, 2)
    got_i, got_d = be.knn_dev((ra.ravel()[6:].copy(), 150), (rb.ravel()[6:].copy(), 90), 2, (70, 1), (70, 1), 64)
    assert np.array_equal(got_i, ref_i) and np.array_equal(got_d, ref_d)


def test_argument_errors(be):
    z = np.zeros((4, 64))
    i = np.zeros((4, 1), np.int32); d = np.zeros((4, 1))
    vp = lambda x: x.ctypes.data_as(C.c_void_p)
    assert be.lib.imgfd_knn(be.ctx, vp(z), 4, vp(z), 4, 65, 1, 0, vp(i), vp(d)) == 1      # dim > 64
    assert be.lib.imgfd_knn(be.ctx, vp(z), 4, vp(z), 4, 64, 9, 0, vp(i), vp(d)) == 1      # k > 8
    assert be.lib.imgfd_knn(be.ctx, vp(z), 4, vp(z), 4, 64, 0, 0, vp(i), vp(d)) == 1
    assert be.lib.imgfd_knn(be.ctx, vp(z), 4, None, 4, 64, 1, 0, vp(i), vp(d)) == 1
    assert be.lib.imgfd_knn(be.ctx, vp(z), 4, vp(z), 0, 64, 1, 0, None, None) == 0        # no queries: nothing to do


@pytest.mark.gpu
def test_readme_use_case_on_the_device():
    """image_surf() on two views of a scene, then get_knnx: every point of the shifted view finds its original"""
    from image_amd import api
    from test_surf import blobs
    img = blobs(301, 420, 300)
    a = img[:, :400]; b = img[:, 20:]                          # same scene, shifted by 20 columns
    to_r = lambda m: np.transpose(m, (2, 1, 0)).astype(np.int32)   # R's (3, W, H) integer array
    sp1, sp2 = api.image_surf(to_r(a), max_points=60), api.image_surf(to_r(b), max_points=60)
    assert sp1["points"] > 10 and sp2["points"] > 10
    knn = api.get_knnx(sp1["surf"], sp2["surf"], k=1)
    ref_i, ref_d = oracle.knn(sp1["surf"], sp2["surf"], 1)
    assert np.array_equal(knn["nn.index"], ref_i + 1) and np.array_equal(knn["nn.dist"], ref_d)
    good = knn["nn.dist"][:, 0] < 0.05
    assert good.sum() >= 5
    to = knn["nn.index"][good, 0] - 1
    assert np.allclose(sp1["x"][to] - sp2["x"][good], 20, atol=1.0) and np.allclose(sp1["y"][to], sp2["y"][good], atol=1.0)
