// image_amd/csrc/canny.hip -- Canny edge detector (K9-K12) behind imgfd_canny / imgfd_canny_dev.
//
// Replaces canny_edge_detector(), image.CannyEdges/src/rcpp_canny.cpp:122-244, with gblur() from
// src/tools.c:146-202 and the union-find of src/adsf.c.
//
//  K9  blur      tools.c:166-185 multiplies FFTs (FFTW3): y = float(x (*) g) with (*) the 2-D CIRCULAR
//                convolution and g[j][i] = exp(-(xi^2+yj^2)/s^2)/sum, xi = i<w/2 ? i : i-w (:146-163).
//                g is an outer product, so this is two 1-D circular convolutions with the wrapped
//                kernels; taps below 1e-22 cannot move a double sum of 0..255 data and are dropped.
//                Accumulated in double in the oracle's tap order (no contraction), rounded to float
//                once (crealf, tools.c:129).  No FFT needed.
//  K10+K11 grad_nms   rcpp_canny.cpp:153-175 (3x3 gradient, clamp-to-edge, hypot) fused with maxima()
//                :88-106 / bilin() :65-85: one workgroup owns a 32x32 tile, stages the blurred tile
//                (+2 halo) and the gradient-magnitude tile (+1 halo, f64) in LDS.  The reference goes
//                atan2 -> cos/sin; we use the unit vector (h,v)/|g| directly (differs by ~1e-16, only
//                exact ties can flip).  Thresholds are int-truncated (:88,:180).
//  K12 hysteresis     rcpp_canny.cpp:184-215 + adsf.c: a pixel survives iff its 8-connected component of
//                marked pixels contains a strong one.  Monotone label propagation (weak -> strong when a
//                strong 8-neighbour exists): each workgroup iterates its 64x64 tile to a local fixpoint in
//                LDS; launches repeat until no tile changed.  The fixpoint is unique, so the result does
//                not depend on scheduling.
#include "common.h"

#include <math.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#define CANNY_MAX_TAPS 129

struct BlurTaps {
    int n;
    int off[CANNY_MAX_TAPS];     // source offset: out[x] += w * in[(x - off) mod N], off in [0, N)
    double w[CANNY_MAX_TAPS];
};

// rows: u8 -> f64 ;  thread per pixel
__global__ void __launch_bounds__(256) canny_blur_rows(const unsigned char *__restrict__ in, int row_stride,
                                                       size_t frame_stride, double *__restrict__ tmp, int nx, int ny,
                                                       BlurTaps t)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= nx) return;
    const unsigned char *row = in + (size_t)blockIdx.z * frame_stride + (size_t)y * row_stride;
    double acc = 0;
    for (int i = 0; i < t.n; i++) {
        int xs = x - t.off[i];
        if (xs < 0) xs += nx;
        acc += t.w[i] * (double)row[xs];
    }
    tmp[((size_t)blockIdx.z * ny + y) * nx + x] = acc;
}

// columns: f64 -> f32 (the single float rounding of tools.c:129)
__global__ void __launch_bounds__(256) canny_blur_cols(const double *__restrict__ tmp, float *__restrict__ out, int nx,
                                                       int ny, BlurTaps t)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= nx) return;
    const double *pl = tmp + (size_t)blockIdx.z * nx * ny;
    double acc = 0;
    for (int i = 0; i < t.n; i++) {
        int ys = y - t.off[i];
        if (ys < 0) ys += ny;
        acc += t.w[i] * pl[(size_t)ys * nx + x];
    }
    out[((size_t)blockIdx.z * ny + y) * nx + x] = (float)acc;
}

#define GN_T 32  // tile edge

__global__ void __launch_bounds__(256) canny_grad_nms(const float *__restrict__ blur, unsigned char *__restrict__ out,
                                                      int nx, int ny, int accGrad, int low_thr, int high_thr)
{
    __shared__ float sb[GN_T + 4][GN_T + 4 + 1];
    __shared__ double sg[GN_T + 2][GN_T + 2 + 1];
    __shared__ double sh[GN_T][GN_T + 1];
    __shared__ double sv[GN_T][GN_T + 1];
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * GN_T, y0 = blockIdx.y * GN_T;
    const float *pl = blur + (size_t)blockIdx.z * nx * ny;
    // blurred tile with a 2-pixel halo, clamp-to-edge (extend(), rcpp_canny.cpp:38-55)
    for (int i = tid; i < (GN_T + 4) * (GN_T + 4); i += 256) {
        const int r = i / (GN_T + 4), c = i - r * (GN_T + 4);
        const int gx = min(max(x0 + c - 2, 0), nx - 1), gy = min(max(y0 + r - 2, 0), ny - 1);
        sb[r][c] = pl[(size_t)gy * nx + gx];
    }
    __syncthreads();
    // gradient magnitude for the tile + 1 halo.  A halo position outside the image stands for the
    // clamped pixel (value()), whose own neighbourhood is again clamped: evaluate at clamped coords.
    for (int i = tid; i < (GN_T + 2) * (GN_T + 2); i += 256) {
        const int r = i / (GN_T + 2), c = i - r * (GN_T + 2);
        const int gx = min(max(x0 + c - 1, 0), nx - 1), gy = min(max(y0 + r - 1, 0), ny - 1);
        // neighbour coordinates clamped to the image, then mapped into the LDS tile
        const int xm = max(gx - 1, 0) - x0 + 2, xp = min(gx + 1, nx - 1) - x0 + 2, xc = gx - x0 + 2;
        const int ym = max(gy - 1, 0) - y0 + 2, yp = min(gy + 1, ny - 1) - y0 + 2, yc = gy - y0 + 2;
        double h, v;
        if (accGrad) {  // rcpp_canny.cpp:157-163, evaluation order preserved
            h = 2 * ((double)sb[yc][xp] - (double)sb[yc][xm]) + (double)sb[yp][xp] - (double)sb[yp][xm] +
                (double)sb[ym][xp] - (double)sb[ym][xm];
            v = 2 * ((double)sb[yp][xc] - (double)sb[ym][xc]) + (double)sb[yp][xp] - (double)sb[ym][xp] +
                (double)sb[yp][xm] - (double)sb[ym][xm];
        } else {        // :167-169
            h = (double)sb[yc][xp] - (double)sb[yc][xm];
            v = (double)sb[yp][xc] - (double)sb[ym][xc];
        }
        sg[r][c] = hypot(h, v);
        if (r >= 1 && r <= GN_T && c >= 1 && c <= GN_T) { sh[r - 1][c - 1] = h; sv[r - 1][c - 1] = v; }
    }
    __syncthreads();
    for (int i = tid; i < GN_T * GN_T; i += 256) {
        const int r = i / GN_T, c = i - r * GN_T;
        const int gx = x0 + c, gy = y0 + r;
        if (gx >= nx || gy >= ny) continue;
        const double now = sg[r + 1][c + 1];
        // unit direction (cos t, sin t) with t = atan2(v,h); atan2(0,0) = 0 -> (1,0)
        double ux = 1.0, uy = 0.0;
        if (now > 0) { ux = sh[r][c] / now; uy = sv[r][c] / now; }
        double val[2];
#pragma unroll
        for (int d = 0; d < 2; d++) {
            const double xt = d ? ux : -ux, yt = d ? uy : -uy;
            // floor() is -1, 0 or (only when the component is exactly 1) 1; in that last case the far
            // tap has weight 0, so evaluating with x1 = 0 gives the same sum from the 3x3 neighbourhood
            const double x1 = fmin(floor(xt), 0.0), y1 = fmin(floor(yt), 0.0);
            const double x2 = x1 + 1, y2 = y1 + 1;
            const int cx = c + 1 + (int)x1, cy = r + 1 + (int)y1;
            const double gx1 = (x2 - xt) * sg[cy][cx] + (xt - x1) * sg[cy][cx + 1];
            const double gx2 = (x2 - xt) * sg[cy + 1][cx] + (xt - x1) * sg[cy + 1][cx + 1];
            val[d] = (y2 - yt) * gx1 + (yt - y1) * gx2;
        }
        const double prev = val[0], next = val[1];
        unsigned char o;
        if ((now <= prev) || (now <= next) || (now <= (double)low_thr)) o = 0;
        else if (now >= (double)high_thr) o = 2;
        else o = 1;
        out[((size_t)blockIdx.z * ny + gy) * nx + gx] = o;
    }
}

#define HY_T 64

// one propagation sweep: tile-local fixpoint in LDS; *changed is set when any pixel of the tile flipped
__global__ void __launch_bounds__(256) canny_hyst_sweep(unsigned char *__restrict__ st, int nx, int ny,
                                                        unsigned *__restrict__ changed)
{
    __shared__ unsigned char s[HY_T + 2][HY_T + 2 + 2];
    __shared__ int flag[2];
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * HY_T, y0 = blockIdx.y * HY_T;
    unsigned char *pl = st + (size_t)blockIdx.z * nx * ny;
    for (int i = tid; i < (HY_T + 2) * (HY_T + 2); i += 256) {
        const int r = i / (HY_T + 2), c = i - r * (HY_T + 2);
        const int gx = x0 + c - 1, gy = y0 + r - 1;
        s[r][c] = (gx >= 0 && gx < nx && gy >= 0 && gy < ny) ? pl[(size_t)gy * nx + gx] : 0;
    }
    if (tid == 0) { flag[0] = 0; flag[1] = 0; }
    __syncthreads();
    // thread owns a 4x4 patch of the 64x64 tile
    const int pr = (tid >> 4) * 4 + 1, pc = (tid & 15) * 4 + 1;
    bool any = false;
    for (int it = 0;; it++) {
        bool ch = false;
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const int rr = pr + r, cc = pc + c;
                if (s[rr][cc] == 1) {
                    const bool strong = s[rr - 1][cc - 1] == 2 || s[rr - 1][cc] == 2 || s[rr - 1][cc + 1] == 2 ||
                                        s[rr][cc - 1] == 2 || s[rr][cc + 1] == 2 || s[rr + 1][cc - 1] == 2 ||
                                        s[rr + 1][cc] == 2 || s[rr + 1][cc + 1] == 2;
                    if (strong) { s[rr][cc] = 2; ch = true; }
                }
            }
        if (ch) { flag[it & 1] = 1; any = true; }
        __syncthreads();
        const int f = flag[it & 1];
        __syncthreads();
        if (tid == 0) flag[it & 1] = 0;  // reused two iterations later, after the next barrier pair
        if (!f) break;
    }
    if (any) {
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const int gx = x0 + pc + c - 1, gy = y0 + pr + r - 1;
                if (gx < nx && gy < ny && s[pr + r][pc + c] == 2) pl[(size_t)gy * nx + gx] = 2;
            }
        atomicOr(changed, 1u);
    }
}

// edges = 255 where strong, else 0 (rcpp_canny.cpp:210-215); per-frame count of 255s (:226-233)
__global__ void __launch_bounds__(256) canny_finalize(const unsigned char *__restrict__ st, unsigned char *__restrict__ edges,
                                                      size_t n_per_frame, unsigned long long *__restrict__ counts)
{
    __shared__ unsigned wsum[4];
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int f = blockIdx.y;
    bool on = false;
    if (i < n_per_frame) {
        on = st[(size_t)f * n_per_frame + i] == 2;
        edges[(size_t)f * n_per_frame + i] = on ? 255 : 0;
    }
    const unsigned c = (unsigned)__popcll(__ballot(on));
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned t = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        if (t) atomicAdd(&counts[f], (unsigned long long)t);
    }
}

namespace {

// wrapped, normalised 1-D kernel (tools.c:146-163) reduced to the taps >= 1e-22, in the oracle's order
imgfd_status make_taps(imgfd_ctx *ctx, int n, double s, BlurTaps *t)
{
    std::vector<double> k(n);
    const double inv_s = 1 / s;
    double sum = 0;
    for (int i = 0; i < n; i++) {
        const double x = i < n / 2 ? i : i - n;
        k[i] = exp(-x * x * inv_s * inv_s);
        sum += k[i];
    }
    t->n = 0;
    for (int i = 0; i < n; i++) {
        const double w = k[i] / sum;
        if (w >= 1e-22) {
            if (t->n == CANNY_MAX_TAPS) return imgfd_fail(ctx, IMGFD_ERR_UNSUPPORTED, "canny: sigma too large (more than 129 taps)");
            t->off[t->n] = i;
            t->w[t->n] = w;
            t->n++;
        }
    }
    return IMGFD_OK;
}

size_t canny_ws_bytes(int nx, int ny, int nf)
{
    const size_t n = (size_t)nx * ny * nf;
    return align_up(n * sizeof(double), 256) + align_up(n * sizeof(float), 256) + align_up(n, 256) + 4096;
}

// all device work for nf frames; d_edges / d_counts are device buffers
imgfd_status canny_device(imgfd_ctx *ctx, const uint8_t *d_in, int row_stride, size_t frame_stride, int nx, int ny,
                          int nf, double s, double low_thr, double high_thr, int accGrad, uint8_t *d_edges,
                          int64_t *d_counts)
{
    const size_t n = (size_t)nx * ny * nf;
    double *tmp = (double *)ws_alloc(ctx, n * sizeof(double));
    float *blur = (float *)ws_alloc(ctx, n * sizeof(float));
    unsigned char *st = (unsigned char *)ws_alloc(ctx, n);
    unsigned *changed = (unsigned *)ws_alloc(ctx, 256);
    if (!tmp || !blur || !st || !changed) return imgfd_fail(ctx, IMGFD_ERR_OOM, "workspace reservation too small");
    if (!(s > 0)) return imgfd_fail(ctx, IMGFD_ERR_INVALID, "canny: s must be positive");
    BlurTaps tx, ty;
    IMGFD_TRY(make_taps(ctx, nx, s, &tx));
    IMGFD_TRY(make_taps(ctx, ny, s, &ty));
    dim3 g1(ceil_div(nx, 256), ny, nf);
    hipLaunchKernelGGL(canny_blur_rows, g1, dim3(256), 0, ctx->stream, d_in, row_stride, frame_stride, tmp, nx, ny, tx);
    hipLaunchKernelGGL(canny_blur_cols, g1, dim3(256), 0, ctx->stream, tmp, blur, nx, ny, ty);
    dim3 g2(ceil_div(nx, GN_T), ceil_div(ny, GN_T), nf);
    hipLaunchKernelGGL(canny_grad_nms, g2, dim3(256), 0, ctx->stream, blur, st, nx, ny, accGrad, (int)low_thr,
                       (int)high_thr);
    IMGFD_HIP(ctx, hipGetLastError());
    // hysteresis: sweeps until a whole sweep changes nothing; the flag is read back every 2 sweeps
    dim3 g3(ceil_div(nx, HY_T), ceil_div(ny, HY_T), nf);
    for (int round = 0; round < 100000; round++) {
        IMGFD_HIP(ctx, hipMemsetAsync(changed, 0, sizeof(unsigned), ctx->stream));
        hipLaunchKernelGGL(canny_hyst_sweep, g3, dim3(256), 0, ctx->stream, st, nx, ny, changed);
        unsigned h = 0;
        if (round & 1) {
            IMGFD_HIP(ctx, hipMemcpyAsync(&h, changed, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
            IMGFD_HIP(ctx, hipStreamSynchronize(ctx->stream));
            if (!h) break;
        }
    }
    IMGFD_HIP(ctx, hipMemsetAsync(d_counts, 0, sizeof(int64_t) * nf, ctx->stream));
    const size_t npf = (size_t)nx * ny;
    hipLaunchKernelGGL(canny_finalize, dim3((unsigned)((npf + 255) / 256), nf), dim3(256), 0, ctx->stream, st, d_edges,
                       npf, (unsigned long long *)d_counts);
    IMGFD_HIP(ctx, hipGetLastError());
    return IMGFD_OK;
}

}  // namespace

extern "C" {

imgfd_status imgfd_canny(imgfd_ctx *ctx, const uint8_t *img, int nx, int ny, double s, double low_thr,
                         double high_thr, int accGrad, uint8_t *edges, int64_t *pixels_nonzero)
{
    if (!ctx) return IMGFD_ERR_INVALID;
    if (!img || !edges || !pixels_nonzero || nx < 1 || ny < 1)
        return imgfd_fail(ctx, IMGFD_ERR_INVALID, "imgfd_canny: bad argument");
    IMGFD_HIP(ctx, hipSetDevice(ctx->device));
    const size_t n = (size_t)nx * ny;
    IMGFD_TRY(ws_reserve(ctx, canny_ws_bytes(nx, ny, 1) + 2 * align_up(n, 256) + 512));
    uint8_t *d_in = (uint8_t *)ws_alloc(ctx, n);
    uint8_t *d_edges = (uint8_t *)ws_alloc(ctx, n);
    int64_t *d_count = (int64_t *)ws_alloc(ctx, sizeof(int64_t));
    if (!d_in || !d_edges || !d_count) return imgfd_fail(ctx, IMGFD_ERR_OOM, "workspace reservation too small");
    IMGFD_HIP(ctx, hipMemcpyAsync(d_in, img, n, hipMemcpyHostToDevice, ctx->stream));
    IMGFD_TRY(canny_device(ctx, d_in, nx, n, nx, ny, 1, s, low_thr, high_thr, accGrad, d_edges, d_count));
    IMGFD_HIP(ctx, hipMemcpyAsync(edges, d_edges, n, hipMemcpyDeviceToHost, ctx->stream));
    IMGFD_HIP(ctx, hipMemcpyAsync(pixels_nonzero, d_count, sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
    IMGFD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return IMGFD_OK;
}

imgfd_status imgfd_canny_dev(imgfd_ctx *ctx, const imgfd_frames *fr, double s, double low_thr,
                             double high_thr, int accGrad, uint8_t *d_edges, int64_t *d_counts)
{
    if (!ctx || !fr || !fr->d_frames || !d_edges || !d_counts || fr->n_frames < 0 || fr->dtype != 0 || fr->nx < 1 || fr->ny < 1)
        return imgfd_fail(ctx, IMGFD_ERR_INVALID, "imgfd_canny_dev: bad argument (frames must be u8)");
    if (!fr->n_frames) return IMGFD_OK;
    IMGFD_HIP(ctx, hipSetDevice(ctx->device));
    const int nx = fr->nx, ny = fr->ny;
    const size_t per_frame = canny_ws_bytes(nx, ny, 1);
    const int chunk = (int)std::max<size_t>(1, std::min<size_t>((size_t)fr->n_frames, ((size_t)3 << 30) / per_frame));
    IMGFD_TRY(ws_reserve(ctx, canny_ws_bytes(nx, ny, chunk) + 512));
    for (int f0 = 0; f0 < fr->n_frames; f0 += chunk) {
        const int nf = std::min(chunk, fr->n_frames - f0);
        ctx->ws_used = 0;
        IMGFD_TRY(canny_device(ctx, (const uint8_t *)fr->d_frames + (size_t)f0 * fr->frame_stride_bytes,
                               fr->row_stride_bytes, fr->frame_stride_bytes, nx, ny, nf, s, low_thr, high_thr, accGrad,
                               d_edges + (size_t)f0 * nx * ny, d_counts + f0));
    }
    return IMGFD_OK;
}

}  // extern "C"
