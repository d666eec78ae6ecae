/*
 * oracle/fftw_stub.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Stand-in for the FFTW3 entry points the reference's Canny blur calls (image.CannyEdges/src/tools.c:89-135), so the
 * reference's own sources can be compiled and run here without the system library.  A 2-D DFT is two passes of 1-D
 * DFTs; each 1-D DFT is evaluated as the defining sum with exact-index twiddles w[(j*k) mod n] and long double
 * accumulation -- O(n^2) per line, slow but transparently correct, accurate to a few 1e-17 relative.  Only the
 * reference-pinning tests and scripts/make_golden.py run it (images up to 640x480).
 */
#include <complex.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "fftw3.h"

struct oracle_fftw_plan {
    int n0, n1, sign;
    double _Complex *in, *out;
};

void *fftw_malloc(size_t n) { return malloc(n ? n : 1); }
void fftw_free(void *p) { free(p); }
void fftw_cleanup(void) {}

fftw_plan fftw_plan_dft_2d(int n0, int n1, fftw_complex *in, fftw_complex *out, int sign, unsigned flags)
{
    (void)flags;
    struct oracle_fftw_plan *p = (struct oracle_fftw_plan *)malloc(sizeof *p);
    p->n0 = n0; p->n1 = n1; p->sign = sign; p->in = in; p->out = out;
    return p;
}

void fftw_destroy_plan(fftw_plan p) { free(p); }

/* y[k] = sum_j x[j*stride] w^(jk), w = exp(sign 2 pi i / n), written to y[k*stride] */
static void dft_line(const long double *wr, const long double *wi, int n, const double _Complex *x, double _Complex *y,
                     size_t stride, long double *tr, long double *ti)
{
    for (int j = 0; j < n; j++) { tr[j] = creal(x[(size_t)j * stride]); ti[j] = cimag(x[(size_t)j * stride]); }
    for (int k = 0; k < n; k++) {
        long double sr = 0, si = 0;
        size_t idx = 0;
        for (int j = 0; j < n; j++) {
            sr += tr[j] * wr[idx] - ti[j] * wi[idx];
            si += tr[j] * wi[idx] + ti[j] * wr[idx];
            idx += (size_t)k;
            if (idx >= (size_t)n) idx -= (size_t)n;
        }
        y[(size_t)k * stride] = (double)sr + (double)si * I;
    }
}

static void twiddles(int n, int sign, long double **wr, long double **wi)
{
    *wr = (long double *)malloc(sizeof(long double) * (size_t)n);
    *wi = (long double *)malloc(sizeof(long double) * (size_t)n);
    const long double two_pi = 6.283185307179586476925286766559005768L;
    for (int m = 0; m < n; m++) {
        const long double a = two_pi * (long double)m / (long double)n;
        (*wr)[m] = cosl(a);
        (*wi)[m] = (long double)sign * sinl(a);
    }
}

void fftw_execute(const fftw_plan p)
{
    const int n0 = p->n0, n1 = p->n1;
    const size_t N = (size_t)n0 * (size_t)n1;
    double _Complex *tmp = (double _Complex *)malloc(sizeof(double _Complex) * N);
    const int nmax = n0 > n1 ? n0 : n1;
    long double *tr = (long double *)malloc(sizeof(long double) * (size_t)nmax);
    long double *ti = (long double *)malloc(sizeof(long double) * (size_t)nmax);
    long double *wr, *wi;
    twiddles(n1, p->sign, &wr, &wi);
    for (int r = 0; r < n0; r++) dft_line(wr, wi, n1, p->in + (size_t)r * n1, tmp + (size_t)r * n1, 1, tr, ti);
    free(wr); free(wi);
    twiddles(n0, p->sign, &wr, &wi);
    for (int c = 0; c < n1; c++) dft_line(wr, wi, n0, tmp + c, p->out + c, (size_t)n1, tr, ti);
    free(wr); free(wi);
    free(tr); free(ti); free(tmp);
}
