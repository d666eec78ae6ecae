"""Harris option sets shared by scripts/make_golden.py (which wrote tests/golden/harris_*.npz) and the tests."""
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
