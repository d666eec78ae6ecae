"""The .Call glue a maintainer drops into the four R packages (r/<pkg>/src/glue.c): R is not installed here, so the files
are compile-checked against r/stub/Rinternals.h (R's own signatures) and include/imgfd.h, and the registered symbols
and arities are compared with the reference's RcppExports."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKGS = {
    "image.CornerDetectionHarris": {"_image_CornerDetectionHarris_detect_corners": 16},
    "image.CornerDetectionF9": {"_image_CornerDetectionF9_detect_corners": 6},
    "image.CannyEdges": {"_image_CannyEdges_canny_edge_detector": 7},
    "image.dlib": {"_image_dlib_dlib_fhog": 6, "_image_dlib_dlib_surf_points": 5},
}


@pytest.mark.parametrize("pkg", sorted(PKGS))
def test_glue_compiles_against_the_c_abi(pkg):
    src = os.path.join(ROOT, "r", pkg, "src", "glue.c")
    cmd = ["gcc", "-std=gnu99", "-Wall", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "r", "stub"),
           "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "r"), src]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


@pytest.mark.parametrize("pkg", sorted(PKGS))
def test_registered_symbols_and_arity_match_the_reference(pkg):
    txt = open(os.path.join(ROOT, "r", pkg, "src", "glue.c")).read()
    entries = dict((m.group(1), int(m.group(2))) for m in re.finditer(r'\{"(_image_\w+)",\s*\(DL_FUNC\)&\w+,\s*(\d+)\}', txt))
    assert entries == PKGS[pkg]
    for name, arity in PKGS[pkg].items():   # the C definition takes that many SEXPs
        sig = re.search(r"SEXP %s\(([^)]*)\)" % name, txt, re.S).group(1)
        assert sig.count("SEXP") == arity, name
    assert ("R_init_" + pkg.replace(".", "_")) in txt
    ref = os.path.join("/root/reference", pkg, "src", "RcppExports.cpp")
    if os.path.exists(ref):                 # only in the build container
        rtxt = open(ref).read()
        for name, arity in PKGS[pkg].items():
            assert re.search(r'\{"%s",\s*\(DL_FUNC\)\s*&%s,\s*%d\}' % (name, name, arity), rtxt), name


def test_no_per_pixel_loop_on_the_r_thread():
    """the per-pixel / per-cell results are widened to R's doubles on the device and land in the vector R allocated
    (imgfd_canny_f64out, imgfd_fhog_f64out): the reference's element-by-element fills (rcpp_canny.cpp:226-233,
    rcpp_fhog.cpp:29-38) have no counterpart in the glue"""
    canny = open(os.path.join(ROOT, "r", "image.CannyEdges", "src", "glue.c")).read()
    assert "imgfd_canny_f64out(" in canny and "REAL(m)" in canny and not re.search(r"for\s*\(", canny)
    dlib = open(os.path.join(ROOT, "r", "image.dlib", "src", "glue.c")).read()
    fhog = dlib[dlib.index("_image_dlib_dlib_fhog("):dlib.index("_image_dlib_dlib_surf_points(")]
    assert "imgfd_fhog_f64out(" in fhog and not re.search(r"for\s*\(", fhog)


def test_r_check_kit_is_complete_and_current():
    """r/check/*.R (the scripts a maintainer with R runs, r/README.md) name their expected values through gold("file"): every such file
    exists under r/check/golden/, and the Harris / FAST-9 tables there are what tests/golden/*.npz hold (scripts/export_golden_for_r.py)"""
    import glob
    import re

    import numpy as np
    gdir = os.path.join(ROOT, "r", "check", "golden")
    named = set()
    for f in glob.glob(os.path.join(ROOT, "r", "check", "check_*.R")):
        txt = open(f).read()
        named |= set(re.findall(r'"([A-Za-z0-9_]+\.(?:csv|pgm|ppm|txt|f32))"', txt))   # file names without a sprintf pattern
    assert len(named) >= 12
    for n in named:
        assert os.path.exists(os.path.join(gdir, n)), n
    g = np.load(os.path.join(ROOT, "tests", "golden", "harris_building.npz"))
    t = np.loadtxt(os.path.join(gdir, "harris_building_default.csv"), delimiter=",", skiprows=1)
    assert t.shape == (251, 3) and np.array_equal(t.astype(np.float32), g["xyR_default"])
    f9 = np.load(os.path.join(ROOT, "tests", "golden", "fast9_chairs.npz"))
    t = np.loadtxt(os.path.join(gdir, "fast9_chairs_t80_n1.csv"), delimiter=",", skiprows=1)
    assert np.array_equal(t[:, 0], f9["xy_t80_n1"][:, 1]) and np.array_equal(t[:, 1], 512 - f9["xy_t80_n1"][:, 0])   # f9_rcpp.cpp:29-30
