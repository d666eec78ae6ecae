/* Replaces image.CornerDetectionF9/src/{RcppExports.cpp, f9_rcpp.cpp, f9.cpp}. */
#include "imgfd_glue.h"

SEXP _image_CornerDetectionF9_detect_corners(SEXP x, SEXP width, SEXP height, SEXP bytes_per_row, SEXP suppress_non_max,
                                             SEXP threshold)
{
    const int w = Rf_asInteger(width), h = Rf_asInteger(height), bpr = Rf_asInteger(bytes_per_row);
    SEXP xi = PROTECT(Rf_coerceVector(x, INTSXP)); /* IntegerVector x */
    if (XLENGTH(xi) < (R_xlen_t)bpr * h) Rf_error("x must hold bytes_per_row*height values");
    imgfd_points out;
    /* INTEGER(x) as it is; (unsigned char) x[i] of f9_rcpp.cpp:10-11 happens on the device */
    imgfd_glue_check(imgfd_fast9_i32(imgfd_glue_ctx(), INTEGER(xi), w, h, bpr, (uint8_t)Rf_asInteger(threshold),
                                     Rf_asLogical(suppress_non_max), &out));
    SEXP cx = PROTECT(Rf_allocVector(REALSXP, out.n)), cy = PROTECT(Rf_allocVector(REALSXP, out.n));
    for (int64_t i = 0; i < out.n; i++) { /* f9_rcpp.cpp:29-30 */
        REAL(cx)[i] = out.points[i].y;
        REAL(cy)[i] = w - out.points[i].x;
    }
    imgfd_free(out.points);
    SEXP res = PROTECT(Rf_allocVector(VECSXP, 2)); /* unnamed list; image_detect_corners.R:57-58 names and classes it */
    SET_VECTOR_ELT(res, 0, cx);
    SET_VECTOR_ELT(res, 1, cy);
    UNPROTECT(4);
    return res;
}

static const R_CallMethodDef CallEntries[] = {
    {"_image_CornerDetectionF9_detect_corners", (DL_FUNC)&_image_CornerDetectionF9_detect_corners, 6}, {NULL, NULL, 0}};

void R_init_image_CornerDetectionF9(DllInfo *dll)
{
    R_registerRoutines(dll, NULL, CallEntries, NULL, NULL);
    R_useDynamicSymbols(dll, FALSE);
}

void R_unload_image_CornerDetectionF9(DllInfo *dll)
{
    (void)dll;
    imgfd_glue_unload();
}
