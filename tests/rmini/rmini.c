/* tests/rmini/rmini.c -- TEST INFRASTRUCTURE: the few entry points of R's C API that r/<pkg>/src/glue.c uses
 * (r/stub/Rinternals.h declares them), implemented just well enough to EXECUTE the glue without R: vectors with R's
 * types and lengths, as.integer / as.numeric coercion with R's NA rules, named lists, matrices in column-major order,
 * the PROTECT stack (its balance is checked), Rf_error as a non-local exit back to the harness, Rprintf into a buffer,
 * and the .Call registration table of R_registerRoutines.  tests/test_r_glue_exec.py builds one shared object per
 * package from this file + the package's glue.c + the imgfd library under test and drives it through ctypes the way
 * R's .Call would: look the routine up by its registered name, check the arity, call it with SEXP arguments.
 * R itself is not in this image; this does not replace an `R CMD check`. */
#include <Rinternals.h>
#include <limits.h>
#include <math.h>
#include <setjmp.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define LGLSXP 10
#define NILSXP 0

struct SEXPREC {
    unsigned type;
    R_xlen_t n;
    void *data;         /* int[] (INTSXP, LGLSXP), double[] (REALSXP), SEXP[] (VECSXP) */
    const char **names; /* VECSXP from Rf_mkNamed */
    int nrow, ncol;     /* matrices: dim attribute */
};

static struct SEXPREC nil_value = {NILSXP, 0, NULL, NULL, 0, 0};
SEXP R_NilValue = &nil_value;

#define NA_INT INT_MIN
static double na_real(void) { return NAN; }

static SEXP new_sexp(unsigned type, R_xlen_t n)
{
    SEXP s = (SEXP)calloc(1, sizeof *s);
    const size_t esz = type == REALSXP ? sizeof(double) : type == VECSXP ? sizeof(SEXP) : sizeof(int);
    s->type = type;
    s->n = n;
    s->data = calloc(n > 0 ? (size_t)n : 1, esz);
    if (type == VECSXP)
        for (R_xlen_t i = 0; i < n; i++) ((SEXP *)s->data)[i] = R_NilValue;
    return s;
}

/* ---- what the glue calls */
double *REAL(SEXP s) { return (double *)s->data; }
int *INTEGER(SEXP s) { return (int *)s->data; }
R_xlen_t XLENGTH(SEXP s) { return s->n; }

static char err_buf[512], print_buf[4096];
static jmp_buf err_jmp;
static int err_armed = 0, protect_depth = 0, protect_max = 0;

void Rf_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf, sizeof err_buf, fmt, ap);
    va_end(ap);
    if (!err_armed) { fprintf(stderr, "rmini: Rf_error outside rmini_call: %s\n", err_buf); abort(); }
    longjmp(err_jmp, 1);
}
void Rprintf(const char *fmt, ...)
{
    const size_t used = strlen(print_buf);
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(print_buf + used, sizeof print_buf - used, fmt, ap);
    va_end(ap);
}

int Rf_asInteger(SEXP s)
{
    if (s->n < 1) return NA_INT;
    if (s->type == INTSXP || s->type == LGLSXP) return INTEGER(s)[0];
    if (s->type == REALSXP) {
        const double v = REAL(s)[0];
        return (isnan(v) || v >= 2147483648.0 || v <= -2147483649.0) ? NA_INT : (int)v;
    }
    Rf_error("rmini: asInteger of type %u", s->type);
}
double Rf_asReal(SEXP s)
{
    if (s->n < 1) return na_real();
    if (s->type == REALSXP) return REAL(s)[0];
    if (s->type == INTSXP || s->type == LGLSXP) return INTEGER(s)[0] == NA_INT ? na_real() : (double)INTEGER(s)[0];
    Rf_error("rmini: asReal of type %u", s->type);
}
int Rf_asLogical(SEXP s)
{
    if (s->n < 1) return NA_INT;
    if (s->type == LGLSXP) return INTEGER(s)[0];
    if (s->type == INTSXP) return INTEGER(s)[0] == NA_INT ? NA_INT : INTEGER(s)[0] != 0;
    if (s->type == REALSXP) return isnan(REAL(s)[0]) ? NA_INT : REAL(s)[0] != 0;
    Rf_error("rmini: asLogical of type %u", s->type);
}
SEXP Rf_coerceVector(SEXP s, unsigned type)
{
    if (s->type == type) return s;
    SEXP r = new_sexp(type, s->n);
    r->nrow = s->nrow; r->ncol = s->ncol;
    if (type == INTSXP && s->type == REALSXP) { /* as.integer: toward zero, NA for NaN and what an int cannot hold */
        for (R_xlen_t i = 0; i < s->n; i++) {
            const double v = REAL(s)[i];
            INTEGER(r)[i] = (isnan(v) || v >= 2147483648.0 || v <= -2147483649.0) ? NA_INT : (int)v;
        }
    } else if (type == REALSXP && (s->type == INTSXP || s->type == LGLSXP)) {
        for (R_xlen_t i = 0; i < s->n; i++) REAL(r)[i] = INTEGER(s)[i] == NA_INT ? na_real() : (double)INTEGER(s)[i];
    } else if (type == INTSXP && s->type == LGLSXP) {
        memcpy(r->data, s->data, sizeof(int) * (size_t)s->n);
    } else {
        Rf_error("rmini: coerceVector %u -> %u", s->type, type);
    }
    return r;
}
SEXP Rf_allocVector(unsigned type, R_xlen_t n)
{
    if (n < 0) Rf_error("rmini: negative length vector");
    return new_sexp(type, n);
}
SEXP Rf_allocMatrix(unsigned type, int nrow, int ncol)
{
    if (nrow < 0 || ncol < 0) Rf_error("rmini: negative extents to matrix");
    SEXP s = new_sexp(type, (R_xlen_t)nrow * ncol);
    s->nrow = nrow; s->ncol = ncol;
    return s;
}
SEXP Rf_mkNamed(unsigned type, const char **names)
{
    R_xlen_t n = 0;
    while (names[n][0]) n++;
    SEXP s = new_sexp(type, n);
    s->names = (const char **)calloc((size_t)n + 1, sizeof(char *));
    for (R_xlen_t i = 0; i < n; i++) s->names[i] = strdup(names[i]);
    return s;
}
SEXP Rf_ScalarInteger(int v) { SEXP s = new_sexp(INTSXP, 1); INTEGER(s)[0] = v; return s; }
SEXP Rf_ScalarReal(double v) { SEXP s = new_sexp(REALSXP, 1); REAL(s)[0] = v; return s; }
SEXP Rf_ScalarLogical(int v) { SEXP s = new_sexp(LGLSXP, 1); INTEGER(s)[0] = v == NA_INT ? NA_INT : v != 0; return s; }
SEXP SET_VECTOR_ELT(SEXP s, R_xlen_t i, SEXP v)
{
    if (s->type != VECSXP || i < 0 || i >= s->n) Rf_error("rmini: SET_VECTOR_ELT out of bounds");
    ((SEXP *)s->data)[i] = v;
    return v;
}
SEXP Rf_protect(SEXP s)
{
    if (++protect_depth > protect_max) protect_max = protect_depth;
    return s;
}
void Rf_unprotect(int n)
{
    if (n > protect_depth) Rf_error("rmini: unprotect(): only %d protected item(s)", protect_depth);
    protect_depth -= n;
}
char *R_alloc(size_t n, int size) { return (char *)calloc(n ? n : 1, (size_t)size); }

/* ---- .Call registration */
static const R_CallMethodDef *registered = NULL;
static int dynamic_symbols = 1;
int R_registerRoutines(DllInfo *dll, const void *c, const R_CallMethodDef *call, const void *f, const void *e)
{
    (void)dll; (void)c; (void)f; (void)e;
    registered = call;
    return 1;
}
Rboolean R_useDynamicSymbols(DllInfo *dll, Rboolean v)
{
    (void)dll;
    const Rboolean old = dynamic_symbols;
    dynamic_symbols = v;
    return old;
}

/* ---- the harness side (ctypes) */
#define RMINI_API __attribute__((visibility("default")))
RMINI_API SEXP rmini_int_vector(const int *v, long n) { SEXP s = new_sexp(INTSXP, n); if (n) memcpy(s->data, v, sizeof(int) * (size_t)n); return s; }
RMINI_API SEXP rmini_real_vector(const double *v, long n) { SEXP s = new_sexp(REALSXP, n); if (n) memcpy(s->data, v, sizeof(double) * (size_t)n); return s; }
RMINI_API SEXP rmini_logical(int v) { return Rf_ScalarLogical(v); }
RMINI_API unsigned rmini_type(SEXP s) { return s->type; }
RMINI_API long rmini_length(SEXP s) { return (long)s->n; }
RMINI_API int rmini_nrow(SEXP s) { return s->nrow; }
RMINI_API int rmini_ncol(SEXP s) { return s->ncol; }
RMINI_API void *rmini_data(SEXP s) { return s->data; }
RMINI_API SEXP rmini_elt(SEXP s, long i) { return (s->type == VECSXP && i >= 0 && i < s->n) ? ((SEXP *)s->data)[i] : NULL; }
RMINI_API const char *rmini_name(SEXP s, long i) { return (s->names && i >= 0 && i < s->n) ? s->names[i] : NULL; }
RMINI_API const char *rmini_last_error(void) { return err_buf; }
RMINI_API const char *rmini_printed(void) { return print_buf; }
RMINI_API void rmini_clear_printed(void) { print_buf[0] = 0; }
RMINI_API int rmini_protect_depth(void) { return protect_depth; }
RMINI_API int rmini_protect_max(void) { return protect_max; }
RMINI_API int rmini_dynamic_symbols(void) { return dynamic_symbols; }
/* the registered routine `name`: its address and arity, as R's .Call resolves it (NULL: not registered) */
RMINI_API void *rmini_lookup(const char *name, int *num_args)
{
    for (const R_CallMethodDef *d = registered; d && d->name; d++)
        if (!strcmp(d->name, name)) { *num_args = d->numArgs; return (void *)d->fun; }
    return NULL;
}
/* .Call(fn, args...): NULL if the routine raised an R error (message: rmini_last_error) */
RMINI_API SEXP rmini_call(void *fn, int nargs, SEXP *a)
{
    SEXP res = NULL;
    err_buf[0] = 0;
    protect_depth = 0; protect_max = 0;
    err_armed = 1;
    if (!setjmp(err_jmp)) {
        switch (nargs) {
        case 5: res = ((SEXP(*)(SEXP, SEXP, SEXP, SEXP, SEXP))fn)(a[0], a[1], a[2], a[3], a[4]); break;
        case 6: res = ((SEXP(*)(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP))fn)(a[0], a[1], a[2], a[3], a[4], a[5]); break;
        case 7: res = ((SEXP(*)(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP))fn)(a[0], a[1], a[2], a[3], a[4], a[5], a[6]); break;
        case 16:
            res = ((SEXP(*)(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP))fn)(
                a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8], a[9], a[10], a[11], a[12], a[13], a[14], a[15]);
            break;
        default: snprintf(err_buf, sizeof err_buf, "rmini: no call shape for %d arguments", nargs); break;
        }
    } else {
        res = NULL; /* R unwinds the protect stack on an error */
        protect_depth = 0;
    }
    err_armed = 0;
    return res;
}
