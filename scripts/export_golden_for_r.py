"""tests/golden/*.npz -> r/check/golden/: the inputs as binary PGM / PPM and the expected R-LEVEL return values as CSV, for the
R scripts under r/check/ (someone with an R installation runs them; this image has none).  The expected values are the
reference's own outputs (scripts/make_golden.py wrote the .npz files from the reference sources compiled in place), mapped to
what each R function returns:
  image_harris():            list(x, y, strength)                         = the corner records (rcpp_harris.cpp:44-57)
  image_detect_corners():    list(x = out.y, y = width - out.x)           (f9_rcpp.cpp:29-30)
  image_canny_edge_detector(): edges[nx, ny] of 0/255 in C index x + nx*y = the R matrix's own column-major memory (rcpp_canny.cpp:226-233)
  image_fhog():              fhog[hog_height, hog_width, 31]              (rcpp_fhog.cpp:29-45)
  image_surf():              x, y, angle, pyramid_scale, score, laplacian, surf[N, 64] (rcpp_surf.cpp:45-52)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
O = os.path.join(ROOT, "r", "check", "golden")
os.makedirs(O, exist_ok=True)


def pnm(name, img):
    with open(os.path.join(O, name), "wb") as f:
        if img.ndim == 2:
            f.write(b"P5\n%d %d\n255\n" % (img.shape[1], img.shape[0]))
        else:
            f.write(b"P6\n%d %d\n255\n" % (img.shape[1], img.shape[0]))
        f.write(np.ascontiguousarray(img, np.uint8).tobytes())


def csv(name, header, cols, fmt="%.9g"):
    np.savetxt(os.path.join(O, name), np.column_stack(cols), delimiter=",", header=",".join(header), comments="", fmt=fmt)


g = np.load(os.path.join(G, "harris_building.npz"))
pnm("harris_building.pgm", g["image"])
for k in ("default", "rcpp_default", "sorted", "n_corners", "two_scales"):
    a = g["xyR_" + k]
    csv(f"harris_building_{k}.csv", ["x", "y", "strength"], [a[:, 0], a[:, 1], a[:, 2]])

g = np.load(os.path.join(G, "fast9_chairs.npz"))
pnm("chairs.pgm", g["image"])
w = g["image"].shape[1]     # the R matrix has `width` rows: image_detect_corners(x) passes nrow(x) as width
for k in ("t80_n0", "t80_n1", "t20_n1"):
    a = g["xy_" + k]
    csv(f"fast9_chairs_{k}.csv", ["x", "y"], [a[:, 1], w - a[:, 0]], fmt="%d")

g = np.load(os.path.join(G, "canny_chairs.npz"))
for acc in (0, 1):
    e = np.unpackbits(g[f"edges_bits_a{acc}"]).reshape(g["image"].shape) * 255
    pnm(f"canny_chairs_edges_accGrad{acc}.pgm", e.astype(np.uint8))
    with open(os.path.join(O, f"canny_chairs_nonzero_accGrad{acc}.txt"), "w") as f:
        f.write("%d\n" % int(g[f"nonzero_a{acc}"]))

g = np.load(os.path.join(G, "fhog_cruise_boat.npz"))
pnm("cruise_boat.ppm", g["image"])
h = g["hog_c8"]            # [rows, cols, 31]; R's as.vector() runs the first index fastest
# 119 040 floats: as little-endian float32 (what dlib computes; readBin(..., size = 4) in R), 476 KB instead of a 3.5 MB text column
h.transpose(2, 1, 0).ravel().astype("<f4").tofile(os.path.join(O, "fhog_cruise_boat_c8.f32"))
with open(os.path.join(O, "fhog_cruise_boat_c8_dim.txt"), "w") as f:
    f.write("%d %d %d\n" % h.shape)

g = np.load(os.path.join(G, "surf_cruise_boat.npz"))
csv("surf_cruise_boat_points.csv", ["x", "y", "angle", "pyramid_scale", "score", "laplacian"],
    [g[k] for k in ("x", "y", "angle", "pyramid_scale", "score", "laplacian")], fmt="%.17g")
csv("surf_cruise_boat_descriptors.csv", [f"d{i}" for i in range(64)], [np.nan_to_num(g["surf"])], fmt="%.17g")
print("wrote", len(os.listdir(O)), "files to", O, file=sys.stderr)
