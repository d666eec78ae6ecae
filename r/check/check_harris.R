# Rscript r/check/check_harris.R   (IMGFD_HOME = the checkout; the package installed from r/image.CornerDetectionHarris, see r/README.md)
source(file.path(Sys.getenv("IMGFD_HOME", "."), "r", "check", "common.R"))
library(image.CornerDetectionHarris)
x <- gray_matrix(read_pnm(gold("harris_building.pgm")))
cmp <- function(got, file, what) {
  ref <- read.csv(gold(file))
  ok(what, length(got$x) == nrow(ref) && all(got$x == ref$x) && all(got$y == ref$y) &&
           all(abs(got$strength - ref$strength) <= 1e-4 * pmax(1, abs(ref$strength))))   # north_star: coordinates exact, strength within 1e-4 (strict mode: equal)
}
r <- image_harris(x)
stopifnot(inherits(r, "image.harris"), identical(names(r), c("x", "y", "strength")), is.double(r$x))
cmp(r, "harris_building_default.csv", "image_harris(building): 251 corners, first (6, 85, 30426.795)")
dc <- function(...) image.CornerDetectionHarris:::detect_corners(as.numeric(x), nrow(x), ncol(x), ...)
cmp(dc(k = 0.06, sigma_d = 1, sigma_i = 2.5, threshold = 130, gaussian = 1L, gradient = 0L, strategy = 0L, Nselect = 1L, measure = 0L,
       Nscales = 1L, precision = 1L, cells = 10L, verbose = FALSE), "harris_building_rcpp_default.csv", "detect_corners(gaussian 1, precision 1): 247 corners")
cmp(dc(0.06, 1, 2.5, 130, 0L, 0L, 1L, 1L, 0L, 1L, 0L, 10L, FALSE), "harris_building_sorted.csv", "strategy 1 (sorted)")
cmp(dc(0.06, 1, 2.5, 130, 0L, 0L, 2L, 50L, 0L, 1L, 0L, 10L, FALSE), "harris_building_n_corners.csv", "strategy 2 (50 strongest)")
cmp(dc(0.06, 1, 2.5, 130, 1L, 0L, 0L, 1L, 0L, 2L, 0L, 10L, FALSE), "harris_building_two_scales.csv", "gaussian 1, Nscales 2: 132 corners")
