/* Replaces image.CannyEdges/src/{RcppExports.cpp, rcpp_canny.cpp, tools.c, adsf.c} and drops the FFTW3 / libpng
 * system requirements (src/Makevars:1). */
#include "imgfd_glue.h"

SEXP _image_CannyEdges_canny_edge_detector(SEXP image, SEXP X, SEXP Y, SEXP s, SEXP low_thr, SEXP high_thr, SEXP accGrad)
{
    const int nx = Rf_asInteger(X), ny = Rf_asInteger(Y);
    const R_xlen_t n = (R_xlen_t)nx * ny;
    SEXP xi = PROTECT(Rf_coerceVector(image, INTSXP));
    if (XLENGTH(xi) < n) Rf_error("image must hold X*Y values");
    int64_t nonzero = 0;
    SEXP m = PROTECT(Rf_allocMatrix(REALSXP, nx, ny)); /* NumericMatrix(nx, ny), rcpp_canny.cpp:226-233 */
    /* the edge map is widened to doubles on the device and lands in the matrix itself: no per-element loop on the R thread */
    imgfd_glue_check(imgfd_canny_f64out(imgfd_glue_ctx(), INTEGER(xi), nx, ny, Rf_asReal(s), Rf_asReal(low_thr),
                                        Rf_asReal(high_thr), Rf_asLogical(accGrad), REAL(m), &nonzero));
    const char *names[] = {"edges", "pixels_nonzero", "nx", "ny", "s", "low_thr", "high_thr", "accGrad", ""};
    SEXP res = PROTECT(Rf_mkNamed(VECSXP, names)); /* :236-243 */
    SET_VECTOR_ELT(res, 0, m);
    SET_VECTOR_ELT(res, 1, Rf_ScalarInteger((int)nonzero));
    SET_VECTOR_ELT(res, 2, Rf_ScalarReal(nx));
    SET_VECTOR_ELT(res, 3, Rf_ScalarReal(ny));
    SET_VECTOR_ELT(res, 4, Rf_ScalarReal(Rf_asReal(s)));
    SET_VECTOR_ELT(res, 5, Rf_ScalarReal(Rf_asReal(low_thr)));
    SET_VECTOR_ELT(res, 6, Rf_ScalarReal(Rf_asReal(high_thr)));
    SET_VECTOR_ELT(res, 7, Rf_ScalarLogical(Rf_asLogical(accGrad)));
    UNPROTECT(3);
    return res;
}

static const R_CallMethodDef CallEntries[] = {
    {"_image_CannyEdges_canny_edge_detector", (DL_FUNC)&_image_CannyEdges_canny_edge_detector, 7}, {NULL, NULL, 0}};

void R_init_image_CannyEdges(DllInfo *dll)
{
    R_registerRoutines(dll, NULL, CallEntries, NULL, NULL);
    R_useDynamicSymbols(dll, FALSE);
}

void R_unload_image_CannyEdges(DllInfo *dll)
{
    (void)dll;
    imgfd_glue_unload();
}
