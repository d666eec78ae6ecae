/*
 * oracle/harris_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, scalar, single thread) of the reference Harris
 * corner detector, image.CornerDetectionHarris/src/ (bnosac/image).  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call it;
 * the product path (image_amd/, libimgfd.so) never links or loads this file.
 *
 * Every function cites the reference lines it restates.  Paths are relative to
 * /root/reference/image.CornerDetectionHarris/src/.
 *
 * Build: gcc -O2 -ffp-contract=off (no -march=native: the reference is built
 * by R with plain -O2, i.e. no FMA contraction on x86-64).
 *
 * Pinning: the reference ships no known-answer test for this path; the
 * restatement is pinned against the reference sources compiled in place
 * (oracle/_ref/libref_harris.so, see oracle/Makefile) on the repository
 * fixtures and on seeded synthetic frames -- tests/test_oracle.py -- and
 * against the committed vectors in tests/golden/ generated from that build.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

/* ---------------------------------------------------------------- H3 ------
 * discrete_gaussian, gaussian.cpp:289-395.
 *  - sigma<=0 or precision<=0: copy (:299-305)
 *  - den = 2*sigma*sigma evaluated in float, widened to double (:307)
 *  - size = (int)(precision*sigma)+1 (:308); size>xdim: return untouched (:312)
 *  - B[i] = 1/(sigma*sqrt(2.0*3.1415926))*exp(-i*i/den), integer -i*i (:316-317)
 *  - norm = 2*sum(B) - B[0] (:319-330)
 *  - rows: double accumulate B[0]*R[i] + sum_j B[j]*(R[i-j]+R[i+j]), stored as
 *    float (:335-361); left border I[size-i], right border I[xdim-i-1] (:345-349)
 *  - columns in place on Is, same borders (:366-392)
 */
ORC_API void orc_gaussian_coeffs(float sigma, int precision, double *B, int *psize)
{
    double den = 2 * sigma * sigma;
    int size = (int)(precision * sigma) + 1;
    *psize = size;
    for (int i = 0; i < size; i++)
        B[i] = 1 / (sigma * sqrt(2.0 * 3.1415926)) * exp(-i * i / den);
    double norm = 0;
    for (int i = 0; i < size; i++) norm += B[i];
    norm *= 2;
    norm -= B[0];
    for (int i = 0; i < size; i++) B[i] /= norm;
}

ORC_API void orc_discrete_gaussian(const float *I, float *Is, int xdim, int ydim,
                                   float sigma, int precision)
{
    if (sigma <= 0 || precision <= 0) {
        if (Is != I) memcpy(Is, I, sizeof(float) * (size_t)xdim * ydim);
        return;
    }
    int size = (int)(precision * sigma) + 1;
    int bdx = xdim + size;
    int bdy = ydim + size;
    if (size > xdim) return;

    double *B = (double *)malloc(sizeof(double) * size);
    orc_gaussian_coeffs(sigma, precision, B, &size);

    double *R = (double *)malloc(sizeof(double) * (size + (xdim > ydim ? xdim : ydim) + size));
    for (int k = 0; k < ydim; k++) {
        for (int i = size; i < bdx; i++) R[i] = I[k * xdim + i - size];
        for (int i = 0, j = bdx; i < size; i++, j++) {
            R[i] = I[k * xdim + size - i];
            R[j] = I[k * xdim + xdim - i - 1];
        }
        for (int i = size; i < bdx; i++) {
            double sum = B[0] * R[i];
            for (int j = 1; j < size; j++) sum += B[j] * (R[i - j] + R[i + j]);
            Is[k * xdim + i - size] = sum;
        }
    }
    for (int k = 0; k < xdim; k++) {
        for (int i = size; i < bdy; i++) R[i] = Is[(i - size) * xdim + k];
        for (int i = 0, j = bdy; i < size; i++, j++) {
            R[i] = Is[(size - i) * xdim + k];
            R[j] = Is[(ydim - i - 1) * xdim + k];
        }
        for (int i = size; i < bdy; i++) {
            double sum = B[0] * R[i];
            for (int j = 1; j < size; j++) sum += B[j] * (R[i - j] + R[i + j]);
            Is[(i - size) * xdim + k] = sum;
        }
    }
    free(R);
    free(B);
}

/* ---------------------------------------------------------------- H4 ------
 * SII "fast Gaussian": sii_precomp gaussian.cpp:61-90, sii_gaussian_conv
 * :179-215, sii_gaussian_conv_image :235-281, extension (clamp) :151-157.
 * K = 3 (gaussian.h:30 default).  All sums are sequential float.
 */
typedef struct { float weights[5]; long radii[5]; int K; } orc_sii;

ORC_API void orc_sii_precomp(double sigma, int K, float *weights, long *radii)
{
    const double sigma0 = 100.0 / 3.14159265358979323846264338327950288;
    static const short radii0[3][5] = {{76, 46, 23, 0, 0}, {82, 56, 37, 19, 0}, {85, 61, 44, 30, 16}};
    static const float weights0[3][5] = {{0.1618f, 0.5502f, 0.9495f, 0, 0},
                                         {0.0976f, 0.3376f, 0.6700f, 0.9649f, 0},
                                         {0.0739f, 0.2534f, 0.5031f, 0.7596f, 0.9738f}};
    const int i = K - 3;
    double sum = 0;
    for (int k = 0; k < K; ++k) {
        radii[k] = (long)(radii0[i][k] * (sigma / sigma0) + 0.5);
        sum += weights0[i][k] * (2 * radii[k] + 1);
    }
    for (int k = 0; k < K; ++k) weights[k] = (float)(weights0[i][k] / sum);
}

static long orc_ext(long N, long n) { return n < 0 ? 0 : (n >= N ? N - 1 : n); }

static void orc_sii_conv(const orc_sii *c, float *dest, float *buffer, const float *src,
                         long N, long stride)
{
    long pad = c->radii[0] + 1;
    float accum = 0;
    buffer += pad;
    for (long n = -pad; n < N + pad; ++n) {
        accum += src[stride * orc_ext(N, n)];
        buffer[n] = accum;
    }
    for (long n = 0; n < N; ++n, dest += stride) {
        accum = c->weights[0] * (buffer[n + c->radii[0]] - buffer[n - c->radii[0] - 1]);
        for (int k = 1; k < c->K; ++k)
            accum += c->weights[k] * (buffer[n + c->radii[k]] - buffer[n - c->radii[k] - 1]);
        *dest = accum;
    }
}

ORC_API void orc_sii_gaussian(const float *src, float *dest, int nx, int ny, float sigma)
{
    orc_sii c;
    c.K = 3;
    orc_sii_precomp(sigma, 3, c.weights, c.radii);
    long m = nx >= ny ? nx : ny;
    float *buffer = (float *)malloc(sizeof(float) * (m + 2 * (c.radii[0] + 1)));
    for (int y = 0; y < ny; ++y) orc_sii_conv(&c, dest + (long)y * nx, buffer, src + (long)y * nx, nx, 1);
    for (int x = 0; x < nx; ++x) orc_sii_conv(&c, dest + x, buffer, dest + x, ny, nx);
    free(buffer);
}

/* gaussian dispatcher, gaussian.cpp:403-430: 0 discrete, 1 SII, else copy. */
ORC_API void orc_gaussian(const float *I, float *Is, int nx, int ny, float sigma, int type)
{
    if (type == 0)
        orc_discrete_gaussian(I, Is, nx, ny, sigma, 3);
    else if (type == 1)
        orc_sii_gaussian(I, Is, nx, ny, sigma);
    else if (Is != I)
        memcpy(Is, I, sizeof(float) * (size_t)nx * ny);
}

/* ------------------------------------------------------------- H5 / H6 ----
 * central_differences gradient.cpp:17-56, sobel_operator :63-106; border rows
 * then border columns are copies of their inner neighbours (:40-55).
 */
ORC_API void orc_gradient(const float *I, float *dx, float *dy, int nx, int ny, int type)
{
    for (int i = 1; i < ny - 1; i++)
        for (int j = 1; j < nx - 1; j++) {
            int p = i * nx + j;
            if (type == 1) {
                dx[p] = 1. / 4. * (I[p + 1] - I[p - 1]) +
                        1. / 8. * (I[p - nx + 1] + I[p + nx + 1] - I[p - nx - 1] - I[p + nx - 1]);
                dy[p] = 1. / 4. * (I[p + nx] - I[p - nx]) +
                        1. / 8. * (I[p + nx + 1] + I[p + nx - 1] - I[p - nx + 1] - I[p - nx - 1]);
            } else {
                dx[p] = 0.5 * (I[p + 1] - I[p - 1]);
                dy[p] = 0.5 * (I[p + nx] - I[p - nx]);
            }
        }
    for (int i = 1; i < nx - 1; i++) {
        dx[i] = dx[i + nx];
        dx[nx * (ny - 1) + i] = dx[nx * (ny - 2) + i];
        dy[i] = dy[i + nx];
        dy[nx * (ny - 1) + i] = dy[nx * (ny - 2) + i];
    }
    for (int i = 0; i < ny; i++) {
        dx[i * nx] = dx[i * nx + 1];
        dx[(i + 1) * nx - 1] = dx[(i + 1) * nx - 2];
        dy[i * nx] = dy[i * nx + 1];
        dy[(i + 1) * nx - 1] = dy[(i + 1) * nx - 2];
    }
}

/* ---------------------------------------------------------------- H7 ------
 * compute_autocorrelation_matrix harris.cpp:44-70: float products, then three
 * in-place gaussians; NO_GAUSSIAN (2) is remapped to FAST (1) (:64-65).
 */
ORC_API void orc_autocorrelation(const float *Ix, const float *Iy, float *A, float *B, float *C,
                                 float sigma, int nx, int ny, int gauss)
{
    for (int i = 0; i < nx * ny; i++) {
        A[i] = Ix[i] * Ix[i];
        B[i] = Ix[i] * Iy[i];
        C[i] = Iy[i] * Iy[i];
    }
    if (gauss == 2) gauss = 1;
    orc_gaussian(A, A, nx, ny, sigma, gauss);
    orc_gaussian(B, B, nx, ny, sigma, gauss);
    orc_gaussian(C, C, nx, ny, sigma, gauss);
}

/* ---------------------------------------------------------------- H8 ------
 * compute_corner_response harris.cpp:78-133.  Harris: all float.  Shi-Tomasi:
 * float expression under a float sqrt (libstdc++ overload picked through
 * `using namespace std`), then 0.5*(A+C)-0.5*D in double.  Harmonic: double
 * divide by (traceA+0.0001).
 */
ORC_API void orc_response(const float *A, const float *B, const float *C, float *R, int measure,
                          int nx, int ny, float k)
{
    int size = nx * ny;
    for (int i = 0; i < size; i++) {
        if (measure == 1) {
            float D = sqrtf(A[i] * A[i] - 2 * A[i] * C[i] + 4 * B[i] * B[i] + C[i] * C[i]);
            float lmin = 0.5 * (A[i] + C[i]) - 0.5 * D;
            R[i] = lmin;
        } else if (measure == 2) {
            float detA = A[i] * C[i] - B[i] * B[i];
            float traceA = A[i] + C[i];
            R[i] = 2 * detA / (traceA + 0.0001);
        } else {
            float detA = A[i] * C[i] - B[i] * B[i];
            float traceA = A[i] + C[i];
            R[i] = detA - k * traceA * traceA;
        }
    }
}

/* ---------------------------------------------------------------- H9 ------
 * non_maximum_suppression harris.cpp:141-255, restated as the reference's own
 * sequential scan-line algorithm (skip[] included), rows in ascending order.
 * Output: xyR triples in raster order.  Returns the corner count (which may
 * exceed cap; only the first cap are stored).
 */
static long nms_scan(const float *R, int nx, int ny, float Th, int radius, float *xyR, long cap, int carry)
{
    if (ny <= 2 * radius + 1 || nx <= 2 * radius + 1) return 0;
    if (radius < 1) radius = 1;
    int *skip = (int *)malloc(sizeof(int) * (size_t)nx * ny);
    for (int i = 0; i < nx * ny; i++) skip[i] = R[i] < Th ? 1 : 0;
    long n = 0;
    for (int i = radius; i < ny - radius; i++) {
        int j = radius;
        while (j < nx - radius && (skip[i * nx + j] || R[i * nx + j - 1] >= R[i * nx + j])) j++;
        while (j < nx - radius) {
            while (j < nx - radius && (skip[i * nx + j] || R[i * nx + j + 1] >= R[i * nx + j])) j++;
            if (j < nx - radius) {
                int p1 = j + 2;
                while (p1 <= j + radius && R[i * nx + p1] < R[i * nx + j]) {
                    skip[i * nx + p1] = 1;
                    p1++;
                }
                if (p1 > j + radius) {
                    int p2 = j - 1;
                    while (p2 >= j - radius && R[i * nx + p2] <= R[i * nx + j]) p2--;
                    if (p2 < j - radius) {
                        int k = i + radius;
                        int found = 0;
                        while (!found && k > i) {
                            int l = j + radius;
                            while (!found && l >= j - radius) {
                                if (R[k * nx + l] > R[i * nx + j]) found = 1;
                                else if (carry) skip[k * nx + l] = 1;
                                l--;
                            }
                            k--;
                        }
                        k = i - radius;
                        while (!found && k < i) {
                            int l = j - radius;
                            while (!found && l <= j + radius) {
                                if (R[k * nx + l] >= R[i * nx + j]) found = 1;
                                l++;
                            }
                            k++;
                        }
                        if (!found) {
                            if (n < cap) {
                                xyR[3 * n + 0] = (float)j;
                                xyR[3 * n + 1] = (float)i;
                                xyR[3 * n + 2] = R[i * nx + j];
                            }
                            n++;
                        }
                    }
                }
                j = p1;
            }
        }
    }
    free(skip);
    return n;
}
ORC_API long orc_nms(const float *R, int nx, int ny, float Th, int radius, float *xyR, long cap)
{
    return nms_scan(R, nx, ny, Th, radius, xyR, cap, 1);
}
/* The row loop of harris.cpp:169-172 is an OpenMP parallel-for in the reference's build (src/Makevars: SHLIB_OPENMP_CXXFLAGS),
 * and the marks of harris.cpp:218 are written into rows BELOW the candidate -- rows another thread may be scanning, or have
 * finished.  Which of them a later row observes depends on the schedule: orc_nms is the one-thread schedule (every mark of
 * the rows above is seen); this is the other extreme, no row sees a mark another row made.  The two differ only when exact
 * ties in adjacent rows meet (tests/test_harris_stages.py). */
ORC_API long orc_nms_rows_independent(const float *R, int nx, int ny, float Th, int radius, float *xyR, long cap)
{
    return nms_scan(R, nx, ny, Th, radius, xyR, cap, 0);
}

/* ---------------------------------------------------------------- H10 -----
 * select_output_corners harris.cpp:263-332.  std::sort is not stable and its
 * order among equal R is implementation-defined; we use a stable merge sort by
 * descending R, so sorted strategies are comparable up to ties only.
 */
typedef struct { float x, y, R; } orc_corner;

static void orc_sort_desc(orc_corner *c, long n)
{
    if (n < 2) return;
    orc_corner *tmp = (orc_corner *)malloc(sizeof(orc_corner) * n);
    for (long w = 1; w < n; w *= 2) {
        for (long lo = 0; lo < n; lo += 2 * w) {
            long mid = lo + w < n ? lo + w : n, hi = lo + 2 * w < n ? lo + 2 * w : n;
            long a = lo, b = mid, o = lo;
            while (a < mid && b < hi) tmp[o++] = (c[b].R > c[a].R) ? c[b++] : c[a++];
            while (a < mid) tmp[o++] = c[a++];
            while (b < hi) tmp[o++] = c[b++];
        }
        memcpy(c, tmp, sizeof(orc_corner) * n);
    }
    free(tmp);
}

ORC_API long orc_select(float *xyR, long n, int strategy, int cells, int N, int nx, int ny)
{
    orc_corner *c = (orc_corner *)xyR;
    if (strategy == 1) {
        orc_sort_desc(c, n);
    } else if (strategy == 2) {
        orc_sort_desc(c, n);
        if (N < n) n = N < 0 ? 0 : N;
    } else if (strategy == 3) {
        int cellx = cells, celly = cells;
        if (cellx > nx) cellx = nx;
        if (celly > ny) celly = ny;
        int size = cellx * celly;
        int Ncell = N / size;
        if (Ncell < 1) Ncell = 1;
        float Dx = (float)nx / cellx;
        float Dy = (float)ny / celly;
        long *cnt = (long *)calloc(size + 1, sizeof(long));
        int *cell = (int *)malloc(sizeof(int) * (n > 0 ? n : 1));
        for (long i = 0; i < n; i++) {
            int px = (float)c[i].x / Dx;
            int py = (float)c[i].y / Dy;
            cell[i] = py * cellx + px;
            cnt[cell[i] + 1]++;
        }
        for (int s = 0; s < size; s++) cnt[s + 1] += cnt[s];
        orc_corner *b = (orc_corner *)malloc(sizeof(orc_corner) * (n > 0 ? n : 1));
        long *pos = (long *)malloc(sizeof(long) * size);
        for (int s = 0; s < size; s++) pos[s] = cnt[s];
        for (long i = 0; i < n; i++) b[pos[cell[i]]++] = c[i];
        long m = 0;
        for (int s = 0; s < size; s++) {
            long len = cnt[s + 1] - cnt[s];
            orc_sort_desc(b + cnt[s], len);
            long take = len > Ncell ? Ncell : len;
            for (long t = 0; t < take; t++) c[m++] = b[cnt[s] + t];
        }
        n = m;
        orc_sort_desc(c, n);
        if (N < n) n = N < 0 ? 0 : N;
        free(cnt); free(cell); free(b); free(pos);
    }
    return n;
}

/* ---------------------------------------------------------------- H11 -----
 * compute_subpixel_precision harris.cpp:340-381; quadratic_approximation
 * interpolation.cpp:27-54; quartic_interpolation :171-212 (Newton, <=20
 * iterations, TOL 1e-10).  float variables, double-promoted constants.
 */
static int orc_quadratic(const float *M, float *x, float *y, float *Mo)
{
    float fx = 0.5 * (M[5] - M[3]);
    float fy = 0.5 * (M[7] - M[1]);
    float fxx = (M[5] - 2 * M[4] + M[3]);
    float fyy = (M[7] - 2 * M[4] + M[1]);
    float fxy = 0.25 * (M[0] - M[2] - M[6] + M[8]);
    float det = fxx * fyy - fxy * fxy;
    if (det * det < 1E-6) return 0;
    float dx = (fyy * fx - fxy * fy) / det;
    float dy = (fxx * fy - fxy * fx) / det;
    *x -= dx;
    *y -= dy;
    *Mo = M[4] + fx * dx + fy * dy + 0.5 * (fxx * dx * dx + 2 * dx * dy * fxy + fyy * dy * dy);
    return 1;
}

static int orc_quartic(const float *M, float *x, float *y, float *Mo)
{
    const float TOL = 1E-10;
    float D[2], b[2], H[3], a[9];
    float dx = 0, dy = 0;
    a[0] = M[4] - 0.5 * (M[1] + M[3] + M[5] + M[7]) + 0.25 * (M[0] + M[2] + M[6] + M[8]);
    a[1] = 0.5 * (M[1] - M[7]) + 0.25 * (-M[0] - M[2] + M[6] + M[8]);
    a[2] = 0.5 * (M[3] - M[5]) + 0.25 * (-M[0] + M[2] - M[6] + M[8]);
    a[3] = 0.5 * (M[3] + M[5]) - M[4];
    a[4] = 0.5 * (M[1] + M[7]) - M[4];
    a[5] = 0.25 * (M[0] - M[2] - M[6] + M[8]);
    a[6] = 0.5 * (M[5] - M[3]);
    a[7] = 0.5 * (M[7] - M[1]);
    a[8] = M[4];
    int i = 0;
    do {
        D[0] = 2 * a[0] * dx * dy * dy + 2 * a[1] * dx * dy + 2 * a[2] * dy * dy + 2 * a[3] * dx + a[5] * dy + a[6];
        D[1] = 2 * a[0] * dx * dx * dy + 2 * a[1] * dx * dx + 2 * a[2] * dx * dy + 2 * a[4] * dy + a[5] * dx + a[7];
        H[0] = 2 * a[0] * dy * dy + 2 * a[1] * dy + 2 * a[3];
        H[1] = 4 * a[0] * dx * dy + 2 * a[1] * dx + 2 * a[2] * dy + a[5];
        H[2] = 2 * a[0] * dx * dx + 2 * a[2] * dx + 2 * a[4];
        float det = H[0] * H[2] - H[1] * H[1];
        if (det * det < 1E-10) return 0;
        b[0] = (D[0] * H[2] - D[1] * H[1]) / det;
        b[1] = (D[1] * H[0] - D[0] * H[1]) / det;
        dx -= b[0];
        dy -= b[1];
        i++;
    } while (D[0] * D[0] + D[1] * D[1] > TOL && i < 20);
    if (dx > 1 || dx < -1 || dy > 1 || dy < -1 || isnan(dx) || isnan(dy)) return 0;
    *x += dx;
    *y += dy;
    *Mo = a[0] * dx * dx * dy * dy + a[1] * dx * dx * dy + a[2] * dx * dy * dy + a[3] * dx * dx +
          a[4] * dy * dy + a[5] * dx * dy + a[6] * dx + a[7] * dy + a[8];
    return 1;
}

ORC_API void orc_subpixel(const float *R, float *xyR, long n, int nx, int type)
{
    orc_corner *c = (orc_corner *)xyR;
    for (long i = 0; i < n; i++) {
        int x = c[i].x, y = c[i].y;
        int mx = x - 1, dx = x + 1, my = y - 1, dy = y + 1;
        float M[9] = {R[my * nx + mx], R[my * nx + x], R[my * nx + dx],
                      R[y * nx + mx],  R[y * nx + x],  R[y * nx + dx],
                      R[dy * nx + mx], R[dy * nx + x], R[dy * nx + dx]};
        if (type == 1) orc_quadratic(M, &c[i].x, &c[i].y, &c[i].R);
        else if (type == 2) orc_quartic(M, &c[i].x, &c[i].y, &c[i].R);
    }
}

/* ---------------------------------------------------------------- H12 -----
 * harris harris.cpp:473-546 (stage order; image smoothed IN PLACE :511; NMS
 * radius = (int)(2*sigma_i+0.5) :523) and harris_scale :554-608 (zoom_out
 * zoom.cpp:121-139 == 2x decimation for integer coordinates; select_corners
 * harris.cpp:443-465 with distance2 :425-435).
 * I is modified (smoothed) like the reference does.  If planes != NULL it must
 * hold 6*nx*ny floats and receives Ix,Iy,A,B,C,R of the finest scale.
 */
static long orc_harris_one(float *I, float **pxyR, int gauss, int grad, int measure, float k,
                           float sigma_d, float sigma_i, float Th, int strategy, int cells, int N,
                           int precision, int nx, int ny, float *planes)
{
    *pxyR = NULL;
    if (nx < 3 || ny < 3) return 0;
    size_t size = (size_t)nx * ny;
    float *buf = planes ? planes : (float *)malloc(sizeof(float) * 6 * size);
    float *Ix = buf, *Iy = buf + size, *A = buf + 2 * size, *B = buf + 3 * size,
          *C = buf + 4 * size, *R = buf + 5 * size;
    orc_gaussian(I, I, nx, ny, sigma_d, gauss);
    orc_gradient(I, Ix, Iy, nx, ny, grad);
    orc_autocorrelation(Ix, Iy, A, B, C, sigma_i, nx, ny, gauss);
    orc_response(A, B, C, R, measure, nx, ny, k);
    int radius = 2 * sigma_i + 0.5;
    long n = orc_nms(R, nx, ny, Th, radius, NULL, 0);
    float *xyR = (float *)malloc(sizeof(float) * 3 * (n > 0 ? n : 1));
    orc_nms(R, nx, ny, Th, radius, xyR, n);
    n = orc_select(xyR, n, strategy, cells, N, nx, ny);
    if (precision == 1 || precision == 2) orc_subpixel(R, xyR, n, nx, precision);
    if (!planes) free(buf);
    *pxyR = xyR;
    return n;
}

static long orc_harris_scale(float *I, float **pxyR, int Nscales, int gauss, int grad, int measure,
                             float k, float sigma_d, float sigma_i, float Th, int strategy,
                             int cells, int N, int precision, int nx, int ny, float *planes)
{
    if (Nscales <= 1 || nx <= 64 || ny <= 64)
        return orc_harris_one(I, pxyR, gauss, grad, measure, k, sigma_d, sigma_i, Th, strategy,
                              cells, N, precision, nx, ny, planes);
    int nxx = nx / 2, nyy = ny / 2;
    float *Iz = (float *)malloc(sizeof(float) * (size_t)nxx * nyy);
    for (int i1 = 0; i1 < nyy; i1++)
        for (int j1 = 0; j1 < nxx; j1++) Iz[i1 * nxx + j1] = I[(2 * i1) * nx + 2 * j1];
    float *cz = NULL;
    long nz = orc_harris_scale(Iz, &cz, Nscales - 1, gauss, grad, measure, k, sigma_d, sigma_i / 2,
                               Th, strategy, cells, N, precision, nxx, nyy, NULL);
    free(Iz);
    float *c1 = NULL;
    long n = orc_harris_one(I, &c1, gauss, grad, measure, k, sigma_d, sigma_i, Th, strategy, cells,
                            N, precision, nx, ny, planes);
    long m = 0;
    for (long i = 0; i < n; i++) {
        long j = 0;
        for (; j < nz; j++) {
            float dx = (cz[3 * j] - c1[3 * i] / 2.);
            float dy = (cz[3 * j + 1] - c1[3 * i + 1] / 2.);
            if (!(dx * dx + dy * dy > sigma_i * sigma_i)) break;
        }
        if (j < nz) {
            c1[3 * m] = c1[3 * i]; c1[3 * m + 1] = c1[3 * i + 1]; c1[3 * m + 2] = c1[3 * i + 2];
            m++;
        }
    }
    free(cz);
    *pxyR = c1;
    return m;
}

/* Entry point mirroring detect_corners(), rcpp_harris.cpp:19-60 (minus Rcpp):
 * img is the float image (already narrowed from double); returns the number of
 * corners, writes up to cap xyR triples.  planes may be NULL. */
ORC_API long orc_harris(const float *img, int nx, int ny, float k, float sigma_d, float sigma_i,
                        float threshold, int gaussian, int gradient, int strategy, int Nselect,
                        int measure, int Nscales, int precision, int cells, float *xyR, long cap,
                        float *planes)
{
    size_t size = (size_t)nx * ny;
    float *I = (float *)malloc(sizeof(float) * (size ? size : 1));
    memcpy(I, img, sizeof(float) * size);
    float *c = NULL;
    long n = orc_harris_scale(I, &c, Nscales, gaussian, gradient, measure, k, sigma_d, sigma_i,
                              threshold, strategy, cells, Nselect, precision, nx, ny, planes);
    if (c) {
        memcpy(xyR, c, sizeof(float) * 3 * (n < cap ? n : cap));
        free(c);
    }
    free(I);
    return n;
}
