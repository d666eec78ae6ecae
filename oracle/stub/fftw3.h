/* oracle/stub/fftw3.h -- TEST INFRASTRUCTURE: the handful of FFTW3 declarations image.CannyEdges/src/tools.c:89-135 uses
 * (fftw_malloc/free, fftw_plan_dft_2d, fftw_execute, fftw_destroy_plan, fftw_cleanup), so that the reference's Canny
 * sources compile where FFTW3 (a system library the reference does not vendor, CE/src/Makevars:1) is absent.  The
 * transforms behind it are the plain discrete Fourier sums of oracle/fftw_stub.c -- same mathematical result as
 * FFTW, different rounding at the 1e-16 level. */
#ifndef ORACLE_STUB_FFTW3_H
#define ORACLE_STUB_FFTW3_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
typedef double fftw_complex[2];          /* what FFTW declares when C99 complex is not in scope */
#else
typedef double _Complex fftw_complex;    /* tools.c includes <complex.h> first: FFTW then uses the native type */
#endif
typedef struct oracle_fftw_plan *fftw_plan;
#define FFTW_FORWARD (-1)
#define FFTW_BACKWARD (+1)
#define FFTW_ESTIMATE (1U << 6)
void *fftw_malloc(size_t n);
void fftw_free(void *p);
/* n0 x n1 row-major (n1 contiguous); out[k0][k1] = sum in[j0][j1] exp(sign 2 pi i (j0 k0/n0 + j1 k1/n1)), unnormalised */
fftw_plan fftw_plan_dft_2d(int n0, int n1, fftw_complex *in, fftw_complex *out, int sign, unsigned flags);
void fftw_execute(const fftw_plan p);
void fftw_destroy_plan(fftw_plan p);
void fftw_cleanup(void);
#ifdef __cplusplus
}
#endif
#endif
