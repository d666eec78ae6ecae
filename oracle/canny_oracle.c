/*
 * oracle/canny_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the reference Canny edge detector,
 * image.CannyEdges/src/rcpp_canny.cpp:122-244 (pipeline), :38-62 (clamp),
 * :65-85 (bilin), :88-106 (maxima); src/tools.c:146-202 (Gaussian blur);
 * src/adsf.c (union-find == 8-connected components).  Only tests/, smoke() and
 * bench.py's cpu_baseline leg may call this file.
 *
 * PINNING.  The reference's blur multiplies FFTs computed by FFTW3 (system library, version not pinned by the
 * reference: CannyEdges/DESCRIPTION:21 "SystemRequirements: libpng, fftw3"; src/Makevars:1), which is neither
 * vendored nor installed here, and the reference holds no test or golden vector for this function.  The pin is
 * therefore the reference's OWN sources -- rcpp_canny.cpp, tools.c, adsf.c -- compiled where they lie into
 * oracle/_ref/libref_canny.so (oracle/Makefile, ref_shim_canny.cpp) with two stand-ins: stub/Rcpp.h for the glue
 * types and fftw_stub.c, plain DFT sums in long double, behind the six FFTW calls of tools.c:89-135.  What is pinned:
 * kernel construction and normalisation, the float cast of the blurred image, gradient, hypot/atan2, bilinear
 * non-maximum suppression, integer-truncated thresholds, union-find hysteresis -- everything but FFTW's own rounding
 * (~1e-13 on 0..255 data, far below the float cast).  tests/test_oracle.py compares edge maps pixel for pixel on
 * synthetic frames, noise and chairs.pgm (0 mismatches observed; bound 1e-5 as SURVEY.md 8d); tests/golden/canny_*.npz
 * were written by that reference build (scripts/make_golden.py).
 *
 * The blur is restated as what tools.c:166-185 computes mathematically:
 *     y = float( x (*) g ),  (*) = 2-D circular convolution,
 *     g[j][i] = exp(-(xi^2+yj^2)/s^2) / sum(g),  xi = i<w/2 ? i : i-w  (:146-163)
 * and because g is an outer product, as two 1-D circular convolutions with the
 * FULL-length wrapped kernels (no truncation), accumulated in double, rounded
 * to float once at the end like crealf does (tools.c:129).
 * tests/test_oracle.py also cross-checks this against a literal numpy fft2/ifft2 restatement (pocketfft): at 640x480,
 * 1000x700 and 4K no float of the blur and no edge pixel differs (test_canny_edges_through_an_independent_fft) -- an FFT's
 * ~1e-13 error can flip a float only within that distance of a rounding boundary, ~1e-8 of the pixels.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

/* 1-D wrapped kernel of tools.c:151-156, unnormalised */
static void wrapped_kernel(double *k, int w, double s)
{
    double inv_s = 1 / s;
    for (int i = 0; i < w; i++) {
        double x = i < w / 2 ? i : i - w;
        k[i] = exp(-x * x * inv_s * inv_s);
    }
}

/* gblur(data,in,nx,ny,1,s): tools.c:189-202 -> gblur_gray :166-185, w=nx, h=ny.
 * out values are floats stored in double, like the reference's data[]. */
ORC_API void orc_canny_blur(const unsigned char *in, double *out, int nx, int ny, double s)
{
    double *kx = (double *)malloc(sizeof(double) * nx);
    double *ky = (double *)malloc(sizeof(double) * ny);
    wrapped_kernel(kx, nx, s);
    wrapped_kernel(ky, ny, s);
    double sx = 0, sy = 0;
    for (int i = 0; i < nx; i++) sx += kx[i];
    for (int j = 0; j < ny; j++) sy += ky[j];
    for (int i = 0; i < nx; i++) kx[i] /= sx;
    for (int j = 0; j < ny; j++) ky[j] /= sy;
    /* keep only taps that can move a double sum of 0..255 data (weight >= 1e-22) */
    int *dx = (int *)malloc(sizeof(int) * nx), *dy = (int *)malloc(sizeof(int) * ny);
    int ntx = 0, nty = 0;
    for (int d = 0; d < nx; d++) if (kx[d] >= 1e-22) dx[ntx++] = d;
    for (int d = 0; d < ny; d++) if (ky[d] >= 1e-22) dy[nty++] = d;
    double *tmp = (double *)malloc(sizeof(double) * (size_t)nx * ny);
    for (int y = 0; y < ny; y++)
        for (int x = 0; x < nx; x++) {
            double acc = 0;
            for (int t = 0; t < ntx; t++) {
                int xs = x - dx[t]; if (xs < 0) xs += nx;
                acc += kx[dx[t]] * (double)in[(size_t)y * nx + xs];
            }
            tmp[(size_t)y * nx + x] = acc;
        }
    for (int y = 0; y < ny; y++)
        for (int x = 0; x < nx; x++) {
            double acc = 0;
            for (int t = 0; t < nty; t++) {
                int ys = y - dy[t]; if (ys < 0) ys += ny;
                acc += ky[dy[t]] * tmp[(size_t)ys * nx + x];
            }
            out[(size_t)y * nx + x] = (double)(float)acc; /* crealf, tools.c:129 */
        }
    free(dx); free(dy);
    free(tmp); free(kx); free(ky);
}

/* extend()/value(), rcpp_canny.cpp:38-62 */
static size_t val(long x, long y, long nx, long ny)
{
    long xt = x < 0 ? 0 : (x > nx - 1 ? nx - 1 : x);
    long yt = y < 0 ? 0 : (y > ny - 1 ? ny - 1 : y);
    return (size_t)(xt + nx * yt);
}

/* gradient loop rcpp_canny.cpp:153-175 */
ORC_API void orc_canny_gradient(const double *data, double *grad, double *theta, int nx, int ny,
                                int accGrad)
{
    for (long x = 0; x < nx; x++)
        for (long y = 0; y < ny; y++) {
            double hgrad, vgrad;
            if (accGrad) {
                hgrad = 2 * (data[val(x + 1, y, nx, ny)] - data[val(x - 1, y, nx, ny)]) +
                        data[val(x + 1, y + 1, nx, ny)] - data[val(x - 1, y + 1, nx, ny)] +
                        data[val(x + 1, y - 1, nx, ny)] - data[val(x - 1, y - 1, nx, ny)];
                vgrad = 2 * (data[val(x, y + 1, nx, ny)] - data[val(x, y - 1, nx, ny)]) +
                        data[val(x + 1, y + 1, nx, ny)] - data[val(x + 1, y - 1, nx, ny)] +
                        data[val(x - 1, y + 1, nx, ny)] - data[val(x - 1, y - 1, nx, ny)];
            } else {
                hgrad = data[val(x + 1, y, nx, ny)] - data[val(x - 1, y, nx, ny)];
                vgrad = data[val(x, y + 1, nx, ny)] - data[val(x, y - 1, nx, ny)];
            }
            grad[y * nx + x] = hypot(hgrad, vgrad);
            theta[y * nx + x] = atan2(vgrad, hgrad);
        }
}

/* bilin rcpp_canny.cpp:65-85 */
static double bilin(const double *grad, double t, long x, long y, long nx, long ny, int dir)
{
    double xt = dir * cos(t), yt = dir * sin(t);
    double x1 = floor(xt), x2 = x1 + 1;
    double y1 = floor(yt), y2 = y1 + 1;
    double gradx1 = (x2 - xt) * grad[val(x + (long)x1, y + (long)y1, nx, ny)] +
                    (xt - x1) * grad[val(x + (long)x2, y + (long)y1, nx, ny)];
    double gradx2 = (x2 - xt) * grad[val(x + (long)x1, y + (long)y2, nx, ny)] +
                    (xt - x1) * grad[val(x + (long)x2, y + (long)y2, nx, ny)];
    return (y2 - yt) * gradx1 + (yt - y1) * gradx2;
}

/* maxima rcpp_canny.cpp:88-106; thresholds are int (doubles truncated at the call :180) */
ORC_API void orc_canny_maxima(const double *grad, const double *theta, unsigned char *output,
                              int nx, int ny, int low_thr, int high_thr)
{
    for (long x = 0; x < nx; x++)
        for (long y = 0; y < ny; y++) {
            double t = theta[y * nx + x];
            double prev = bilin(grad, t, x, y, nx, ny, -1);
            double next = bilin(grad, t, x, y, nx, ny, 1);
            double now = grad[y * nx + x];
            if ((now <= prev) || (now <= next) || (now <= low_thr)) output[y * nx + x] = 0;
            else if (now >= high_thr) output[y * nx + x] = 2;
            else output[y * nx + x] = 1;
        }
}

/* hysteresis rcpp_canny.cpp:184-215 with adsf.c:16-50 restated literally */
static int adsf_find(int *t, int a)
{
    int r = a;
    while (t[r] != r) r = t[r];
    while (t[a] != r) { int nx = t[a]; t[a] = r; a = nx; }
    return r;
}
static void adsf_union(int *t, int a, int b)
{
    a = adsf_find(t, a);
    b = adsf_find(t, b);
    if (a != b) { if (a < b) t[b] = a; else t[a] = b; }
}

ORC_API long orc_canny_hysteresis(unsigned char *output, int nx, int ny)
{
    int N = nx * ny;
    int *t = (int *)malloc(sizeof(int) * (size_t)N);
    for (int i = 0; i < N; i++) t[i] = i;
    for (long x = 0; x < nx; x++)
        for (long y = 0; y < ny; y++) {
            int d = x + nx * y;
            if (output[d])
                for (int ex = -1; ex < 2; ex++)
                    for (int ey = -1; ey < 2; ey++) {
                        int ed = (int)val(x + ex, y + ey, nx, ny);
                        if (output[ed]) adsf_union(t, d, ed);
                    }
        }
    for (int d = 0; d < N; d++)
        if (output[d] == 2) output[adsf_find(t, d)] = 2;
    long nonzero = 0;
    for (int d = 0; d < N; d++) {
        if (output[adsf_find(t, d)] < 2) output[d] = 0;
        else { output[d] = (unsigned char)(char)-1; nonzero++; }
    }
    free(t);
    return nonzero;
}

/* canny_edge_detector rcpp_canny.cpp:122-244 minus the Rcpp marshalling.
 * image: nx*ny bytes, index x + nx*y.  edges: nx*ny bytes 0/255.
 * Optional debug outputs (may be NULL): blur (nx*ny doubles), grad, nms. */
ORC_API long orc_canny(const unsigned char *image, int nx, int ny, double s, double low_thr,
                       double high_thr, int accGrad, unsigned char *edges, double *blur_out,
                       double *grad_out, unsigned char *nms_out)
{
    size_t n = (size_t)nx * ny;
    double *data = (double *)malloc(sizeof(double) * n);
    double *grad = (double *)malloc(sizeof(double) * n);
    double *theta = (double *)malloc(sizeof(double) * n);
    orc_canny_blur(image, data, nx, ny, s);
    orc_canny_gradient(data, grad, theta, nx, ny, accGrad);
    orc_canny_maxima(grad, theta, edges, nx, ny, (int)low_thr, (int)high_thr);
    if (blur_out) memcpy(blur_out, data, sizeof(double) * n);
    if (grad_out) memcpy(grad_out, grad, sizeof(double) * n);
    if (nms_out) memcpy(nms_out, edges, n);
    long nonzero = orc_canny_hysteresis(edges, nx, ny);
    free(data); free(grad); free(theta);
    return nonzero;
}
