"""Generate tests/golden/*.npz from the reference's own code (oracle/_ref, built by
`make -C oracle ref` from /root/reference) on the reference's own example images
and on seeded synthetic frames.  Run in the build container only:

    python scripts/make_golden.py

The .npz files carry the input image (so the GPU box, which has no
/root/reference, can replay them) and the expected outputs.  Canny's vectors
come from the reference's own rcpp_canny.cpp / tools.c / adsf.c compiled in
place (oracle/_ref/libref_canny.so); the FFTW3 calls of tools.c are served by
the plain DFT of oracle/fftw_stub.c, as FFTW3 itself is absent here.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))

import fixtures  # noqa: E402
import oracle  # noqa: E402
from image_amd import pnm, synth  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
REF = fixtures.REF

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


def harris_golden(name, img):
    out = {"image": img.astype(np.uint8)}
    for case, kw in HARRIS_CASES.items():
        out["xyR_" + case] = oracle.ref_harris(img, **kw)
    np.savez_compressed(os.path.join(OUT, name), **out)
    print(name, {k: v.shape for k, v in out.items()})


def fhog_golden():
    """dlib's own extract_fhog_features on the reference's example image (image.dlib/inst/extdata)."""
    from PIL import Image
    img = np.asarray(Image.open(os.path.join(REF, "image.dlib", "inst", "extdata", "cruise_boat.png")).convert("RGB"))
    out = {"image": img, "hog_c8": oracle.ref_fhog(img, 8, 1, 1), "hog_c4": oracle.ref_fhog(img, 4, 1, 1),
           "hog_c8_p33": oracle.ref_fhog(img, 8, 3, 3)}
    np.savez_compressed(os.path.join(OUT, "fhog_cruise_boat"), **out)
    print("fhog_cruise_boat", {k: v.shape for k, v in out.items()})


def fhog_dlib_kat():
    """dlib's OWN known-answer vectors for extract_fhog_features: the `face.dng` image and the serialized feature arrays
    embedded in dlib/test/fhog.cpp:156-214 (RGB at two cell sizes, grayscale), dumped by oracle/_ref/dlib_kat (built from
    that test source where it lies, `make -C oracle dlib_kat`).  dlib's criterion for them is max |diff| < 1e-6 (:33-52)."""
    import struct
    import subprocess
    import tempfile
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "dlib_kat"])
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "kat.bin")
        subprocess.check_call([os.path.join(ROOT, "oracle", "_ref", "dlib_kat"), path])   # exits non-zero if dlib itself misses its vectors
        b = open(path, "rb").read()
    rows, cols = struct.unpack_from("<ii", b, 0)
    off = 8
    out = {"rgb": np.frombuffer(b, np.uint8, rows * cols * 3, off).reshape(rows, cols, 3).copy()}
    off += rows * cols * 3
    out["gray"] = np.frombuffer(b, np.uint8, rows * cols, off).reshape(rows, cols).copy()
    off += rows * cols
    for name in ("rgb_a", "rgb_b", "gray_a"):
        sbin, nr, nc = struct.unpack_from("<iii", b, off)
        off += 12
        out["cell_" + name] = np.array(sbin)
        out["hog_" + name] = np.frombuffer(b, np.float32, nr * nc * 31, off).reshape(nr, nc, 31).copy()
        off += nr * nc * 31 * 4
    assert off == len(b)
    np.savez_compressed(os.path.join(OUT, "fhog_dlib_kat"), **out)
    print("fhog_dlib_kat", {k: (v.shape if v.ndim else int(v)) for k, v in out.items()})


def surf_golden():
    """dlib's own get_surf_points (max_points 1000, threshold 30: the R defaults) on the reference's example image."""
    from PIL import Image
    img = np.asarray(Image.open(os.path.join(REF, "image.dlib", "inst", "extdata", "cruise_boat.png")).convert("RGB"))
    out = {"image": img, **oracle.surf(img, 1000, 30.0, use_ref=True)}
    np.savez_compressed(os.path.join(OUT, "surf_cruise_boat"), **out)
    print("surf_cruise_boat", {k: v.shape for k, v in out.items()})


def fast9_golden(name, img, thresholds):
    out = {"image": img.astype(np.uint8)}
    for thr in thresholds:
        for nms in (0, 1):
            out[f"xy_t{thr}_n{nms}"] = oracle.ref_fast9(img, thr, bool(nms))
    np.savez_compressed(os.path.join(OUT, name), **out)
    print(name, {k: v.shape for k, v in out.items()})


CANNY_CASES = {
    "a0": dict(accGrad=False),                       # the C++ default
    "a1": dict(accGrad=True),                        # the R wrapper's default (canny_edges_detector.R:63)
    "s1_t2_6": dict(s=1.0, low_thr=2, high_thr=6),
    "s3p5_t1_4": dict(s=3.5, low_thr=1, high_thr=4),
}


def canny_golden(name, img, cases=("a0", "a1")):
    """edge maps written by the reference's own canny_edge_detector(); the restatement must agree before anything is saved"""
    out = {"image": img.astype(np.uint8), "pinned": np.array(1)}
    for case in cases:
        kw = CANNY_CASES[case]
        edges, n = oracle.ref_canny(img, **kw)
        o_edges, o_n = oracle.canny(img, **kw)
        assert n == o_n and np.array_equal(edges, o_edges), f"oracle/canny_oracle.c disagrees with the reference on {name}/{case}"
        out[f"edges_bits_{case}"] = np.packbits(edges > 0)
        out[f"nonzero_{case}"] = np.array(n)
    np.savez_compressed(os.path.join(OUT, name), **out)
    print(name, {k: (v.shape, int(v) if v.ndim == 0 else None) for k, v in out.items()})


def main():
    os.makedirs(OUT, exist_ok=True)
    building = fixtures.read_rds_int_matrix(
        f"{REF}/image.CornerDetectionHarris/inst/extdata/building.rds")
    chairs = pnm.read_pgm(f"{REF}/image.CornerDetectionF9/inst/extdata/chairs.pgm")
    harris_golden("harris_building", building)
    harris_golden("harris_synth_640x480_seed1", synth.frame(1, 640, 480))
    fast9_golden("fast9_chairs", chairs, (20, 80, 100))
    fast9_golden("fast9_synth_640x480_seed1", synth.frame(1, 640, 480), (10, 20, 50))
    # R hands Canny the column-major memory of grey[row, col], i.e. the transposed raster
    canny_golden("canny_chairs", np.ascontiguousarray(chairs.T))
    canny_golden("canny_synth_320x240_seed7", synth.frame(7, 320, 240, n_rect=20), tuple(CANNY_CASES))
    fhog_dlib_kat()
    fhog_golden()   # tests/golden/fhog_cruise_boat.npz
    surf_golden()   # tests/golden/surf_cruise_boat.npz


if __name__ == "__main__":
    main()
