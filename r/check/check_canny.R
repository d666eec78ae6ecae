# Rscript r/check/check_canny.R   (expected maps: the reference's own code over a stand-in DFT for FFTW3, tests/golden/canny_chairs.npz)
source(file.path(Sys.getenv("IMGFD_HOME", "."), "r", "check", "common.R"))
library(image.CannyEdges)
x <- gray_matrix(read_pnm(gold("chairs.pgm")))
for (acc in c(TRUE, FALSE)) {
  r <- image_canny_edge_detector(x, s = 2, low_thr = 3, high_thr = 10, accGrad = acc)
  stopifnot(inherits(r, "image_canny"), identical(names(r), c("edges", "pixels_nonzero", "nx", "ny", "s", "low_thr", "high_thr", "accGrad")),
            is.double(r$edges), identical(dim(r$edges), c(512L, 512L)), all(r$edges %in% c(0, 255)))
  ref <- read_pnm(gold(sprintf("canny_chairs_edges_accGrad%d.pgm", acc)))$bytes     # raster order = the matrix's own memory order
  n <- as.integer(readLines(gold(sprintf("canny_chairs_nonzero_accGrad%d.txt", acc))))
  bad <- sum(as.vector(r$edges) != ref)
  ok(sprintf("chairs.pgm accGrad %s: %d edge pixels (anchor %d), %d differ", acc, r$pixels_nonzero, n, bad),
     bad <= 1e-5 * length(ref) + 1 && abs(r$pixels_nonzero - n) <= 1 && r$pixels_nonzero == sum(r$edges > 0))   # SURVEY 8d config 3: mismatch rate <= 1e-5
}
