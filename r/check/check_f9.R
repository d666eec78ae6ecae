# Rscript r/check/check_f9.R
source(file.path(Sys.getenv("IMGFD_HOME", "."), "r", "check", "common.R"))
library(image.CornerDetectionF9)
x <- gray_matrix(read_pnm(gold("chairs.pgm")))
cmp <- function(got, file, what) {
  ref <- read.csv(gold(file))
  ok(what, inherits(got, "image.corners") && identical(names(got), c("x", "y")) && is.double(got$x) &&
           length(got$x) == nrow(ref) && all(got$x == ref$x) && all(got$y == ref$y))   # bit-exact, order included
}
cmp(image_detect_corners(x, threshold = 80, suppress_non_max = FALSE), "fast9_chairs_t80_n0.csv", "chairs.pgm, threshold 80: 926 corners")
cmp(image_detect_corners(x, threshold = 80, suppress_non_max = TRUE), "fast9_chairs_t80_n1.csv", "chairs.pgm, threshold 80, non-max: 347 corners")
cmp(image_detect_corners(x, threshold = 20, suppress_non_max = TRUE), "fast9_chairs_t20_n1.csv", "chairs.pgm, threshold 20, non-max: 1919 corners")
