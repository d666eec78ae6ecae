/* r/stub/Rinternals.h -- COMPILE-CHECK STUB, not R.  R is not installed in the build image; this header declares just
 * the part of R's C API the glue files use, with R's own signatures, so that `gcc -fsyntax-only` (tests/test_r_glue.py)
 * catches typos and arity mistakes.  A real build uses R's <Rinternals.h> (R CMD INSTALL puts it on the include path
 * ahead of this directory, which is never shipped in a package). */
#ifndef R_STUB_RINTERNALS_H
#define R_STUB_RINTERNALS_H
#include <stddef.h>
typedef struct SEXPREC *SEXP;
typedef ptrdiff_t R_xlen_t;
typedef void *(*DL_FUNC)(void);
typedef struct _DllInfo DllInfo;
typedef struct { const char *name; DL_FUNC fun; int numArgs; } R_CallMethodDef;
typedef int Rboolean;
#define FALSE 0
#define TRUE 1
#define REALSXP 14
#define INTSXP 13
#define VECSXP 19
extern SEXP R_NilValue;
double *REAL(SEXP);
int *INTEGER(SEXP);
R_xlen_t XLENGTH(SEXP);
int Rf_asInteger(SEXP);
double Rf_asReal(SEXP);
int Rf_asLogical(SEXP);
SEXP Rf_coerceVector(SEXP, unsigned);
SEXP Rf_allocVector(unsigned, R_xlen_t);
SEXP Rf_allocMatrix(unsigned, int, int);
SEXP Rf_mkNamed(unsigned, const char **);
SEXP Rf_ScalarInteger(int);
SEXP Rf_ScalarReal(double);
SEXP Rf_ScalarLogical(int);
SEXP SET_VECTOR_ELT(SEXP, R_xlen_t, SEXP);
SEXP Rf_protect(SEXP);
void Rf_unprotect(int);
#define PROTECT(s) Rf_protect(s)
#define UNPROTECT(n) Rf_unprotect(n)
void Rf_error(const char *, ...) __attribute__((noreturn));
void Rprintf(const char *, ...);
char *R_alloc(size_t, int);
int R_registerRoutines(DllInfo *, const void *, const R_CallMethodDef *, const void *, const void *);
Rboolean R_useDynamicSymbols(DllInfo *, Rboolean);
#endif
