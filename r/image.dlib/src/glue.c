/* Replaces image.dlib/src/{RcppExports.cpp, rcpp_fhog.cpp, rcpp_surf.cpp, dlib-core.cpp} (no dlib unity build). */
#include "imgfd_glue.h"

SEXP _image_dlib_dlib_fhog(SEXP x, SEXP rows, SEXP cols, SEXP cell_size, SEXP frp, SEXP fcp)
{
    const int nr = Rf_asInteger(rows), nc = Rf_asInteger(cols);
    SEXP xi = PROTECT(Rf_coerceVector(x, INTSXP)); /* std::vector<int> x: element [ch, c, r] at ch + 3*c + 3*cols*r */
    if (XLENGTH(xi) < (R_xlen_t)3 * nr * nc) Rf_error("x must hold 3*rows*cols values");
    int hr = 0, hc = 0;
    /* (a context-free call: it leaves no message behind for imgfd_glue_check) */
    if (imgfd_fhog_size(nr, nc, Rf_asInteger(cell_size), Rf_asInteger(frp), Rf_asInteger(fcp), &hr, &hc) != IMGFD_OK)
        Rf_error("imgfd: dlib_fhog: rows, cols >= 0 and cell_size, filter_rows_padding, filter_cols_padding >= 1 "
                 "(DLIB_ASSERT of fhog.h:712-720)");
    const R_xlen_t n = (R_xlen_t)31 * hr * hc;
    SEXP f = PROTECT(Rf_allocVector(REALSXP, n)); /* filled in the order of rcpp_fhog.cpp:29-38, widened on the device */
    if (n)
        imgfd_glue_check(imgfd_fhog_f64out(imgfd_glue_ctx(), INTEGER(xi), nr, nc, Rf_asInteger(cell_size), Rf_asInteger(frp),
                                           Rf_asInteger(fcp), REAL(f), (int64_t)n, &hr, &hc));
    const char *names[] = {"hog_height", "hog_width", "fhog", "hog_cell_size", "filter_rows_padding", "filter_cols_padding", ""};
    SEXP res = PROTECT(Rf_mkNamed(VECSXP, names)); /* rcpp_fhog.cpp:40-45 */
    SET_VECTOR_ELT(res, 0, Rf_ScalarInteger(hr));
    SET_VECTOR_ELT(res, 1, Rf_ScalarInteger(hc));
    SET_VECTOR_ELT(res, 2, f);
    SET_VECTOR_ELT(res, 3, Rf_ScalarInteger(Rf_asInteger(cell_size)));
    SET_VECTOR_ELT(res, 4, Rf_ScalarInteger(Rf_asInteger(frp)));
    SET_VECTOR_ELT(res, 5, Rf_ScalarInteger(Rf_asInteger(fcp)));
    UNPROTECT(3);
    return res; /* image_fhog.R:46 reshapes $fhog to [hog_height, hog_width, 31] */
}

SEXP _image_dlib_dlib_surf_points(SEXP x, SEXP rows, SEXP cols, SEXP max_points, SEXP detection_threshold)
{
    const int nr = Rf_asInteger(rows), nc = Rf_asInteger(cols);
    SEXP xi = PROTECT(Rf_coerceVector(x, INTSXP));
    if (XLENGTH(xi) < (R_xlen_t)3 * nr * nc) Rf_error("x must hold 3*rows*cols values");
    imgfd_surf_out o;
    imgfd_glue_check(imgfd_surf_i32(imgfd_glue_ctx(), INTEGER(xi), nr, nc, (long)Rf_asReal(max_points),
                                    Rf_asReal(detection_threshold), &o));
    const char *names[] = {"points", "x", "y", "angle", "pyramid_scale", "score", "laplacian", "surf", ""};
    SEXP res = PROTECT(Rf_mkNamed(VECSXP, names)); /* rcpp_surf.cpp:45-52 */
    const double *src[6] = {o.x, o.y, o.angle, o.pyramid_scale, o.score, o.laplacian};
    SET_VECTOR_ELT(res, 0, Rf_ScalarReal((double)o.n)); /* sp.size() */
    for (int k = 0; k < 6; k++) {
        SEXP v = PROTECT(Rf_allocVector(REALSXP, o.n));
        for (int64_t i = 0; i < o.n; i++) REAL(v)[i] = src[k][i];
        SET_VECTOR_ELT(res, 1 + k, v);
        UNPROTECT(1);
    }
    SEXP m = PROTECT(Rf_allocMatrix(REALSXP, (int)o.n, 64)); /* NumericMatrix(n, 64): column-major */
    for (int64_t i = 0; i < o.n; i++)
        for (int j = 0; j < 64; j++) REAL(m)[i + j * o.n] = o.surf[i * 64 + j];
    SET_VECTOR_ELT(res, 7, m);
    if (o.n) imgfd_free(o.data);
    UNPROTECT(3);
    return res; /* image_surf.R:88 then zeroes NaNs */
}

static const R_CallMethodDef CallEntries[] = {{"_image_dlib_dlib_fhog", (DL_FUNC)&_image_dlib_dlib_fhog, 6},
                                              {"_image_dlib_dlib_surf_points", (DL_FUNC)&_image_dlib_dlib_surf_points, 5},
                                              {NULL, NULL, 0}};

void R_init_image_dlib(DllInfo *dll)
{
    R_registerRoutines(dll, NULL, CallEntries, NULL, NULL);
    R_useDynamicSymbols(dll, FALSE);
}

void R_unload_image_dlib(DllInfo *dll)
{
    (void)dll;
    imgfd_glue_unload();
}
