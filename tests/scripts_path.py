"""Option sets shared by scripts/make_golden.py (which wrote tests/golden/*.npz) and the tests."""
HARRIS_CASES = {
    "default": dict(),
    "rcpp_default": dict(gaussian=1, precision=1),
    "no_gaussian": dict(gaussian=2),
    "sobel": dict(gradient=1),
    "shi_tomasi": dict(measure=1, threshold=1.0),
    "harmonic": dict(measure=2, threshold=1.0),
    "quartic": dict(precision=2),
    "sorted": dict(strategy=1),
    "n_corners": dict(strategy=2, Nselect=50),
    "distributed": dict(strategy=3, Nselect=100),
    "two_scales": dict(gaussian=1, Nscales=2),
    "three_scales": dict(Nscales=3),
}

# tests/golden/canny_*.npz: key suffix -> canny_edge_detector() arguments
CANNY_CASES = {
    "a0": dict(accGrad=False),
    "a1": dict(accGrad=True),
    "s1_t2_6": dict(s=1.0, low_thr=2, high_thr=6),
    "s3p5_t1_4": dict(s=3.5, low_thr=1, high_thr=4),
}
