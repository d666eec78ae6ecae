# r/check/common.R -- helpers of the four check scripts (base R only)
gold <- function(f) file.path(Sys.getenv("IMGFD_HOME", "."), "r", "check", "golden", f)
# binary PGM (P5) / PPM (P6), maxval 255 -> list(w, h, channels, bytes in raster order)
read_pnm <- function(path) {
  con <- file(path, "rb"); on.exit(close(con))
  hdr <- character(0)
  while (length(hdr) < 4) {
    tok <- scan(con, what = "", n = 1, quiet = TRUE, comment.char = "#")
    hdr <- c(hdr, tok)
  }
  w <- as.integer(hdr[2]); h <- as.integer(hdr[3]); ch <- if (hdr[1] == "P6") 3L else 1L
  list(w = w, h = h, channels = ch, bytes = as.integer(readBin(con, "raw", n = w * h * ch)))
}
# a gray image as the matrix the R wrappers expect: nrow = width, ncol = height (x[i, j] = pixel column i, row j)
gray_matrix <- function(p) matrix(p$bytes, nrow = p$w, ncol = p$h)
ok <- function(what, cond) { cat(sprintf("%-62s %s\n", what, if (isTRUE(cond)) "OK" else "MISMATCH")); invisible(isTRUE(cond)) }
