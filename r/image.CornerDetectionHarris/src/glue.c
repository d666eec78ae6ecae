/* Replaces image.CornerDetectionHarris/src/{RcppExports.cpp, rcpp_harris.cpp} and the bundled algorithm sources.
 * R/RcppExports.R stays as it is: .Call('_image_CornerDetectionHarris_detect_corners', ...16 args...). */
#include "imgfd_glue.h"

SEXP _image_CornerDetectionHarris_detect_corners(SEXP x, SEXP nx, SEXP ny, SEXP k, SEXP sigma_d, SEXP sigma_i,
        SEXP threshold, SEXP gaussian, SEXP gradient, SEXP strategy, SEXP Nselect, SEXP measure, SEXP Nscales,
        SEXP precision, SEXP cells, SEXP verbose)
{
    const int w = Rf_asInteger(nx), h = Rf_asInteger(ny);
    SEXP xr = PROTECT(Rf_coerceVector(x, REALSXP)); /* NumericVector x */
    if (XLENGTH(xr) < (R_xlen_t)w * h) Rf_error("x must hold nx*ny values");
    imgfd_corners out;
    /* REAL(x) goes down as it is; the (float) x[i] narrowing of rcpp_harris.cpp:34-35 happens on the device */
    imgfd_glue_check(imgfd_harris_f64(imgfd_glue_ctx(), REAL(xr), w, h, (float)Rf_asReal(k), (float)Rf_asReal(sigma_d),
                                      (float)Rf_asReal(sigma_i), (float)Rf_asReal(threshold), Rf_asInteger(gaussian),
                                      Rf_asInteger(gradient), Rf_asInteger(strategy), Rf_asInteger(Nselect),
                                      Rf_asInteger(measure), Rf_asInteger(Nscales), Rf_asInteger(precision),
                                      Rf_asInteger(cells), Rf_asInteger(verbose), &out));
    if (Rf_asInteger(verbose)) { /* harris.cpp:389-416, 504-538 */
        static const char *nm[7] = {" 1.Smoothing the image: \t \t", " 2.Computing the gradient: \t \t",
                                    " 3.Computing the autocorrelation: \t", " 4.Computing corner strength function: \t",
                                    " 5.Non-maximum suppression:  \t\t", " 6.Selecting output corners:  \t\t",
                                    " 7.Calculating subpixel accuracy: \t"};
        Rprintf("\nHarris corner detection:\n[nx=%d, ny=%d, sigma_i=%f]\n", w, h, Rf_asReal(sigma_i));
        for (int i = 0; i < 7; i++) Rprintf("%sTime: %fs\n", nm[i], out.stage_seconds[i]);
        Rprintf(" * Number of corners detected: %ld\n", (long)out.n);
    }
    SEXP xs = PROTECT(Rf_allocVector(REALSXP, out.n)), ys = PROTECT(Rf_allocVector(REALSXP, out.n)),
         st = PROTECT(Rf_allocVector(REALSXP, out.n));
    for (int64_t i = 0; i < out.n; i++) { /* rcpp_harris.cpp:44-57 */
        REAL(xs)[i] = out.corners[i].x;
        REAL(ys)[i] = out.corners[i].y;
        REAL(st)[i] = out.corners[i].R;
    }
    imgfd_free(out.corners);
    const char *names[] = {"x", "y", "strength", ""};
    SEXP res = PROTECT(Rf_mkNamed(VECSXP, names));
    SET_VECTOR_ELT(res, 0, xs);
    SET_VECTOR_ELT(res, 1, ys);
    SET_VECTOR_ELT(res, 2, st);
    UNPROTECT(5);
    return res; /* pkg.R:104 adds class "image.harris" */
}

static const R_CallMethodDef CallEntries[] = {
    {"_image_CornerDetectionHarris_detect_corners", (DL_FUNC)&_image_CornerDetectionHarris_detect_corners, 16},
    {NULL, NULL, 0}};

void R_init_image_CornerDetectionHarris(DllInfo *dll) /* RcppExports.cpp:35-43 */
{
    R_registerRoutines(dll, NULL, CallEntries, NULL, NULL);
    R_useDynamicSymbols(dll, FALSE);
}

void R_unload_image_CornerDetectionHarris(DllInfo *dll)
{
    (void)dll;
    imgfd_glue_unload();
}
