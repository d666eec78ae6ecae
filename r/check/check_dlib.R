# Rscript r/check/check_dlib.R
source(file.path(Sys.getenv("IMGFD_HOME", "."), "r", "check", "common.R"))
library(image.dlib)
p <- read_pnm(gold("cruise_boat.ppm"))
x <- array(p$bytes, dim = c(3L, p$w, p$h))        # what image_fhog() / image_surf() take: integer RGB, (3, width, height)
f <- image_fhog(x, cell_size = 8L, filter_rows_padding = 1L, filter_cols_padding = 1L)
d <- scan(gold("fhog_cruise_boat_c8_dim.txt"), quiet = TRUE)
ref <- readBin(gold("fhog_cruise_boat_c8.f32"), "numeric", n = prod(d), size = 4, endian = "little")   # dlib's floats, first index fastest
ok(sprintf("image_fhog(cruise_boat): [%d, %d, 31], max |diff| %.3g", f$hog_height, f$hog_width, max(abs(as.vector(f$fhog) - ref))),
   identical(dim(f$fhog), as.integer(d)) && max(abs(as.vector(f$fhog) - ref)) <= 1e-6)          # dlib's own tolerance (test/fhog.cpp)
s <- image_surf(x, max_points = 1000, detection_threshold = 30)
pts <- read.csv(gold("surf_cruise_boat_points.csv")); des <- as.matrix(read.csv(gold("surf_cruise_boat_descriptors.csv")))
same <- s$points == nrow(pts) && all(s$x == pts$x) && all(s$y == pts$y) && all(s$laplacian == pts$laplacian)
ok(sprintf("image_surf(cruise_boat): %d points (expected %d)", s$points, nrow(pts)), same)
if (same) ok(sprintf("  scale / score / angle / descriptors: max |diff| %.3g", max(abs(s$surf - des))),
             all(abs(s$pyramid_scale - pts$pyramid_scale) <= 1e-9 * abs(pts$pyramid_scale)) && all(abs(s$score - pts$score) <= 1e-9 * abs(pts$score)) &&
             max(abs(s$angle - pts$angle)) <= 1e-9 && max(abs(s$surf - des)) <= 1e-6)
