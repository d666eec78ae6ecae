// image_amd/csrc/sii.hip -- stacked-integral-images "fast Gaussian" (K6), gaussian code 1 (and the Gaussian of the
// structure tensor for code 2, harris.cpp:64-65).
//
// Replaces sii_precomp / sii_gaussian_conv / sii_gaussian_conv_image, image.CornerDetectionHarris/src/gaussian.cpp:61-281
// with K = 3 boxes (gaussian.h:23-31) and constant boundary extension (:151-157).  The reference forms a running
// FLOAT prefix sum over n = -pad .. N+pad-1 (:193-197) and evaluates  out[n] = sum_k w_k (cum[n+r_k] - cum[n-r_k-1])
// (:202-212).  Float addition is not associative, so the prefix sum must be formed sequentially in the reference's
// order to reproduce its bits (a parallel scan differs by far more than 1e-4 in R):
//   sii_cum_rows : one lane per image row; 64x64 tiles are transposed through LDS so global traffic stays coalesced
//   sii_cum_cols : one lane per image column (naturally coalesced), marching down the rows
//   sii_box_rows / sii_box_cols : fully parallel evaluation of the three boxes from the cumulative plane
// Results are bit-identical to the reference (library built -ffp-contract=off).
#include "common.h"

#include <math.h>

struct SiiCoeffs {
    float w[3];
    int r[3];
    int pad;  // radii[0] + 1
};

// sii_precomp, gaussian.cpp:61-90, K = 3
static SiiCoeffs sii_precomp3(double sigma)
{
    const double sigma0 = 100.0 / 3.14159265358979323846264338327950288;
    static const short radii0[3] = {76, 46, 23};
    static const float weights0[3] = {0.1618f, 0.5502f, 0.9495f};
    SiiCoeffs c;
    double sum = 0;
    for (int k = 0; k < 3; k++) {
        c.r[k] = (int)(long)(radii0[k] * (sigma / sigma0) + 0.5);
        sum += weights0[k] * (2 * c.r[k] + 1);
    }
    for (int k = 0; k < 3; k++) c.w[k] = (float)(weights0[k] / sum);
    c.pad = c.r[0] + 1;
    return c;
}

size_t sii_scratch_floats(int nx, int ny, float sigma)
{
    if (!(sigma > 0)) sigma = 1e-3f;
    const SiiCoeffs c = sii_precomp3(sigma);
    return (size_t)(nx + 2 * c.pad) * (size_t)(ny + 2 * c.pad);
}

// cum[(y)*(nx+2*pad) + n + pad] = sum_{m=-pad..n} src[y][clamp(m)]   (float, sequential)
__global__ void __launch_bounds__(64) sii_cum_rows(const float *__restrict__ src, float *__restrict__ cum, int nx, int ny,
                                                   int pad)
{
    __shared__ float tile[64][65];
    const int lane = threadIdx.x;
    const int y0 = blockIdx.x * 64;
    const size_t fo = (size_t)blockIdx.y * nx * ny, co = (size_t)blockIdx.y * (nx + 2 * pad) * ny;
    const float *s = src + fo;
    float *c = cum + co;
    const int y = y0 + lane;
    const bool ok = y < ny;
    const int cw = nx + 2 * pad;
    float acc = 0.f;
    // left extension: the first sample repeated (extension(), :151-157)
    const float first = ok ? s[(size_t)y * nx] : 0.f;
    for (int n = 0; n < pad; n++) {
        acc += first;
        if (ok) c[(size_t)y * cw + n] = acc;
    }
    for (int x0 = 0; x0 < nx; x0 += 64) {
        // coalesced load of a 64x64 tile: lane = column
        for (int r = 0; r < 64; r++) {
            const int yy = y0 + r, xx = x0 + lane;
            tile[r][lane] = (yy < ny && xx < nx) ? s[(size_t)yy * nx + xx] : 0.f;
        }
        __syncthreads();
        const int lim = min(64, nx - x0);
        for (int k = 0; k < lim; k++) {  // lane = row: sequential float prefix sum
            acc += tile[lane][k];
            tile[lane][k] = acc;
        }
        __syncthreads();
        for (int r = 0; r < 64; r++) {
            const int yy = y0 + r, xx = x0 + lane;
            if (yy < ny && xx < nx) c[(size_t)yy * cw + pad + xx] = tile[r][lane];
        }
        __syncthreads();
    }
    const float last = ok ? s[(size_t)y * nx + nx - 1] : 0.f;
    for (int n = 0; n < pad; n++) {
        acc += last;
        if (ok) c[(size_t)y * cw + pad + nx + n] = acc;
    }
}

// cum[(n + pad)*nx + x] = sum_{m=-pad..n} src[clamp(m)][x]
__global__ void __launch_bounds__(64) sii_cum_cols(const float *__restrict__ src, float *__restrict__ cum, int nx, int ny,
                                                   int pad)
{
    const int x = blockIdx.x * 64 + threadIdx.x;
    if (x >= nx) return;
    const float *s = src + (size_t)blockIdx.y * nx * ny;
    float *c = cum + (size_t)blockIdx.y * nx * (ny + 2 * pad);
    float acc = 0.f;
    const float first = s[x];
    for (int n = 0; n < pad; n++) { acc += first; c[(size_t)n * nx + x] = acc; }
    int y = 0;
    for (; y + 8 <= ny; y += 8) {
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = s[(size_t)(y + k) * nx + x];
#pragma unroll
        for (int k = 0; k < 8; k++) { acc += v[k]; c[(size_t)(pad + y + k) * nx + x] = acc; }
    }
    for (; y < ny; y++) { acc += s[(size_t)y * nx + x]; c[(size_t)(pad + y) * nx + x] = acc; }
    const float last = s[(size_t)(ny - 1) * nx + x];
    for (int n = 0; n < pad; n++) { acc += last; c[(size_t)(pad + ny + n) * nx + x] = acc; }
}

// out[n] = w0*(cum[n+r0] - cum[n-r0-1]); accum += w_k*(...)   (:202-212), along rows / along columns
__global__ void __launch_bounds__(256) sii_box_rows(const float *__restrict__ cum, float *__restrict__ out, int nx, int ny,
                                                    SiiCoeffs c)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= nx) return;
    const float *b = cum + ((size_t)blockIdx.z * ny + y) * (nx + 2 * c.pad) + c.pad + x;
    float accum = c.w[0] * (b[c.r[0]] - b[-c.r[0] - 1]);
    accum += c.w[1] * (b[c.r[1]] - b[-c.r[1] - 1]);
    accum += c.w[2] * (b[c.r[2]] - b[-c.r[2] - 1]);
    out[((size_t)blockIdx.z * ny + y) * nx + x] = accum;
}

__global__ void __launch_bounds__(256) sii_box_cols(const float *__restrict__ cum, float *__restrict__ out, int nx, int ny,
                                                    SiiCoeffs c)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= nx) return;
    const float *b = cum + (size_t)blockIdx.z * nx * (ny + 2 * c.pad) + (size_t)(c.pad + y) * nx + x;
    const ptrdiff_t s = nx;
    float accum = c.w[0] * (b[s * c.r[0]] - b[-s * (c.r[0] + 1)]);
    accum += c.w[1] * (b[s * c.r[1]] - b[-s * (c.r[1] + 1)]);
    accum += c.w[2] * (b[s * c.r[2]] - b[-s * (c.r[2] + 1)]);
    out[((size_t)blockIdx.z * ny + y) * nx + x] = accum;
}

// gaussian(I, Is, nx, ny, sigma, FAST_GAUSSIAN, K=3), gaussian.cpp:419-425.  d_in may equal d_out (the reference works
// in place in compute_autocorrelation_matrix).  d_cum: sii_scratch_floats(nx, ny, sigma) * n_frames floats.
imgfd_status launch_sii_gaussian(imgfd_ctx *ctx, const float *d_in, float *d_out, int nx, int ny, int n_frames, float sigma,
                                 float *d_cum)
{
    if (!d_cum) return imgfd_fail(ctx, IMGFD_ERR_INVALID, "SII gaussian needs its cumulative-sum scratch");
    const SiiCoeffs c = sii_precomp3(sigma);
    hipLaunchKernelGGL(sii_cum_rows, dim3(ceil_div(ny, 64), n_frames), dim3(64), 0, ctx->stream, d_in, d_cum, nx, ny, c.pad);
    hipLaunchKernelGGL(sii_box_rows, dim3(ceil_div(nx, 256), ny, n_frames), dim3(256), 0, ctx->stream, d_cum, d_out, nx, ny, c);
    hipLaunchKernelGGL(sii_cum_cols, dim3(ceil_div(nx, 64), n_frames), dim3(64), 0, ctx->stream, d_out, d_cum, nx, ny, c.pad);
    hipLaunchKernelGGL(sii_box_cols, dim3(ceil_div(nx, 256), ny, n_frames), dim3(256), 0, ctx->stream, d_cum, d_out, nx, ny, c);
    IMGFD_HIP(ctx, hipGetLastError());
    return IMGFD_OK;
}
