/*
 * oracle/ref_shim_harris.cpp -- TEST INFRASTRUCTURE.
 * extern "C" doorways into the UNMODIFIED reference Harris sources, which the
 * Makefile compiles in place from /root/reference (never copied into this
 * repo) into oracle/_ref/libref_harris.so.  Used to pin oracle/harris_oracle.c
 * and (bench.py) as the "reference" CPU baseline.
 */
#include <vector>
#include <string.h>
#include "harris.h"
#include "gaussian.h"
#include "gradient.h"
#ifdef _OPENMP
#include <omp.h>
#endif

/* external-linkage functions the reference defines but does not declare in a header */
void compute_autocorrelation_matrix(float *Ix, float *Iy, float *A, float *B, float *C,
                                    float sigma, int nx, int ny, int gauss);
void compute_corner_response(float *A, float *B, float *C, float *R, int measure, int nx, int ny,
                             float k);

extern "C" {

void ref_set_threads(int n)
{
#ifdef _OPENMP
    omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int ref_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* whole pipeline as rcpp_harris.cpp:19-60 drives it (img is copied: harris() smooths in place) */
long ref_harris(const float *img, int nx, int ny, float k, float sigma_d, float sigma_i,
                float threshold, int gaussian, int gradient, int strategy, int Nselect,
                int measure, int Nscales, int precision, int cells, float *xyR, long cap)
{
    std::vector<harris_corner> corners;
    std::vector<float> I(img, img + (size_t)nx * ny);
    harris_scale(I.data(), corners, Nscales, gaussian, gradient, measure, k, sigma_d, sigma_i,
                 threshold, strategy, cells, Nselect, precision, nx, ny, 0);
    long n = (long)corners.size();
    for (long i = 0; i < n && i < cap; i++) {
        xyR[3 * i] = corners[i].x; xyR[3 * i + 1] = corners[i].y; xyR[3 * i + 2] = corners[i].R;
    }
    return n;
}

/* stage doorways */
void ref_gaussian(const float *I, float *Is, int nx, int ny, float sigma, int type)
{
    if (Is != I) memcpy(Is, I, sizeof(float) * (size_t)nx * ny);
    gaussian(Is, Is, nx, ny, sigma, type);
}
void ref_gradient(const float *I, float *Ix, float *Iy, int nx, int ny, int type)
{
    gradient(const_cast<float *>(I), Ix, Iy, nx, ny, type);
}
void ref_autocorrelation(const float *Ix, const float *Iy, float *A, float *B, float *C,
                         float sigma, int nx, int ny, int gauss)
{
    compute_autocorrelation_matrix(const_cast<float *>(Ix), const_cast<float *>(Iy), A, B, C,
                                   sigma, nx, ny, gauss);
}
void ref_response(const float *A, const float *B, const float *C, float *R, int measure, int nx,
                  int ny, float k)
{
    compute_corner_response(const_cast<float *>(A), const_cast<float *>(B),
                            const_cast<float *>(C), R, measure, nx, ny, k);
}
long ref_nms(const float *R, int nx, int ny, float Th, int radius, float *xyR, long cap)
{
    std::vector<harris_corner> corners;
    non_maximum_suppression(const_cast<float *>(R), corners, Th, radius, nx, ny);
    long n = (long)corners.size();
    for (long i = 0; i < n && i < cap; i++) {
        xyR[3 * i] = corners[i].x; xyR[3 * i + 1] = corners[i].y; xyR[3 * i + 2] = corners[i].R;
    }
    return n;
}

} /* extern "C" */
