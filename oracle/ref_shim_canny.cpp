/*
 * oracle/ref_shim_canny.cpp -- TEST INFRASTRUCTURE.
 * extern "C" doorway into the UNMODIFIED reference Canny sources (image.CannyEdges/src/rcpp_canny.cpp, tools.c, adsf.c),
 * which the Makefile compiles in place from /root/reference (never copied into this repo) into
 * oracle/_ref/libref_canny.so.  Rcpp and FFTW3 are absent here: oracle/stub/Rcpp.h supplies the few Rcpp types the glue
 * function uses and oracle/fftw_stub.c the DFTs behind tools.c's FFTW calls (plain Fourier sums; see there).
 * Everything else -- kernel construction and normalisation, the float cast of the blurred image, gradient, hypot/atan2,
 * bilinear non-maximum suppression, thresholds, the union-find hysteresis -- is the reference's own code.
 */
#include <Rcpp.h>

#include <string.h>

Rcpp::List canny_edge_detector(Rcpp::IntegerVector image, int X, int Y, double s, double low_thr, double high_thr, bool accGrad);

extern "C" __attribute__((visibility("default"))) long ref_canny(const int *image, int nx, int ny, double s, double low_thr,
                                                                 double high_thr, int accGrad, unsigned char *edges)
{
    const Rcpp::IntegerVector iv(image, (size_t)nx * (size_t)ny);
    const Rcpp::List z = canny_edge_detector(iv, nx, ny, s, low_thr, high_thr, accGrad != 0);
    const std::vector<double> &e = z.get("edges").matrix.data();
    for (size_t i = 0; i < e.size(); i++) edges[i] = (unsigned char)e[i];
    return (long)z.get("pixels_nonzero").scalar;
}
