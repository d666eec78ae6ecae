// image_amd/csrc/canny.hip -- Canny edge detector (K9-K12) behind imgfd_canny / imgfd_canny_dev.
//
// Replaces canny_edge_detector(), image.CannyEdges/src/rcpp_canny.cpp:122-244, with gblur() from
// src/tools.c:146-202 and the union-find of src/adsf.c.
//
//  K9  blur      tools.c:166-185 multiplies FFTs (FFTW3): y = float(x (*) g) with (*) the 2-D CIRCULAR
//                convolution and g[j][i] = exp(-(xi^2+yj^2)/s^2)/sum, xi = i<w/2 ? i : i-w (:146-163).
//                g is an outer product, so this is two 1-D circular convolutions with the wrapped
//                kernels; taps below 1e-17 are dropped: together they move a blurred value by < 1e-16 (a unit in the
//                last place of the double sum; the reference's FFT itself carries ~1e-13), see make_taps.
//                Accumulated in double in the oracle's tap order (no contraction), rounded to float
//                once (crealf, tools.c:129).  No FFT needed.
//  K10+K11 grad_nms   rcpp_canny.cpp:153-175 (3x3 gradient, clamp-to-edge, hypot) fused with maxima()
//                :88-106 / bilin() :65-85: one workgroup owns a 32x32 tile, stages the blurred tile
//                (+2 halo) and the gradient-magnitude tile (+1 halo, f64) in LDS.  The reference goes
//                atan2 -> cos/sin; we use the unit vector (h,v)/|g| directly (differs by ~1e-16, only
//                exact ties can flip).  Thresholds are int-truncated (:88,:180).
//  K12 hysteresis     rcpp_canny.cpp:184-215 + adsf.c: a pixel survives iff its 8-connected component of
//                marked pixels contains a strong one.  The NMS kernel publishes two bit planes per frame
//                (S = strong, W = marked; one 64-bit __ballot word per 64 pixels of a row).  Propagation
//                S |= W & dilate3x3(S) is monotone with a unique fixpoint, so scheduling cannot change the
//                result.  canny_hyst_block: a wave per tile, a block of tiles per workgroup iterated to the
//                block's fixpoint through LDS, a fixed number of launches queued (no host read-back);
//                canny_finish: union-find over whatever the launches left, then the 0/255 bytes and the count.
// the blurred plane and the edge map go out with streaming stores (imgfd_canny_dev on 32 4K frames: 1.916 -> 1.904 ms)
#define IMGFD_NT_OUT 1
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


// the same two passes for kernels of more than CANNY_MAX_TAPS taps (s beyond ~10): taps in global memory
__global__ void __launch_bounds__(256) canny_blur_rows_g(const unsigned char *__restrict__ in, int row_stride, size_t frame_stride,
                                                         double *__restrict__ tmp, int nx, int ny, const int *__restrict__ off,
                                                         const double *__restrict__ w, int nt)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= nx) return;
    const unsigned char *row = in + (size_t)blockIdx.z * frame_stride + (size_t)y * row_stride;
    double acc = 0;
    for (int i = 0; i < nt; i++) {
        int xs = x - off[i];
        if (xs < 0) xs += nx;
        acc += w[i] * (double)row[xs];
    }
    tmp[((size_t)blockIdx.z * ny + y) * nx + x] = acc;
}
__global__ void __launch_bounds__(256) canny_blur_cols_g(const double *__restrict__ tmp, float *__restrict__ out, int nx, int ny,
                                                         const int *__restrict__ off, const double *__restrict__ w, int nt)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= nx) return;
    const double *pl = tmp + (size_t)blockIdx.z * nx * ny;
    double acc = 0;
    for (int i = 0; i < nt; i++) {
        int ys = y - off[i];
        if (ys < 0) ys += ny;
        acc += w[i] * pl[(size_t)ys * nx + x];
    }
    out[((size_t)blockIdx.z * ny + y) * nx + x] = (float)acc;
}

// ------------------------------------------------------------------ K9 fast path: fused separable blur
// One 256-thread workgroup owns a 64-column strip and marches down a segment of rows in chunks of 32:
//   load   : u8 rows (64 + 2*HL columns, wrap-around addressing) -> LDS, next chunk prefetched in registers
//   rows   : a thread owns 8 consecutive pixels of one row; pair sums of the u8 taps are exact integers,
//            one v_cvt_f64_u32 + one f64 FMA per pair; results stay f64 (tools.c multiplies spectra: there
//            is no rounding between the two passes) in an LDS ring of row-filtered rows
//   columns: a thread owns one column and 8 consecutive output rows, streams the 8+2R ring rows once
//            (lane <-> column: conflict-free ds_read_b64), 8 accumulators; the single float rounding
//            (crealf, tools.c:129) happens at the 256-byte coalesced store.
// HBM traffic is the algorithmic 1 B read + 4 B written per pixel (+ halo, served by L2).
#define BM_TW 64
// Ring row = BM_TW doubles stored transposed by 8 (column c at (c % 8) * 8 + c / 8) and BM_RP apart.  The row pass hands
// each thread 8 adjacent columns (strip s, output o -> position o * 8 + s): the 8 strips of a row write 8 consecutive
// doubles per ds_write_b64 and, 72 = 8 mod 16, the two rows of a 16-lane group fall into the other half of the banks --
// no conflicts (straight [row][column] storage put all 64 lanes of a write on two bank pairs).  The column pass gives
// lane l the position l (consecutive doubles) and works out which column that is.
#define BM_RP (BM_TW + 8)
#define BM_CH 32
#define BM_NT 256
#define BM_PX 8

struct BlurMarchParams {
    const unsigned char *in;
    float *out;
    int nx, ny;
    int row_stride;       // bytes between input rows
    long frame_stride;    // bytes between input frames
    int seg_rows;
    int strips, xcd_order;  // strips per frame row (grid.x = strips x segments); 1: XCD-aware order of the (segment, strip) tiles
    int aligned4;         // base, strides and nx are multiples of 4: dword tile loads
    double wx[33];        // wx[j], j = 0..R: normalised taps along x (zero padded up to the template radius)
    double wy[33];
};

__device__ __forceinline__ int wrap_idx(int i, int n)
{
    i %= n;
    return i < 0 ? i + n : i;
}

template <int R>
struct BlurGeom {
    static constexpr int HL = (R + 3) / 4 * 4;          // halo columns each side, whole dwords
    static constexpr int WT = BM_TW + 2 * HL;           // tile width in bytes
    static constexpr int WD = WT / 4;                   // ... in dwords
    static constexpr int PITCH = WD + 1 - (WD & 1);     // odd dword pitch: two rows cover all 32 banks
    static constexpr int NQ = (HL - R + BM_PX + 2 * R + 3) / 4;  // dwords of one thread's byte window
    static constexpr int RING = (BM_CH + 2 * R) <= 64 ? 64 : 128;
    static constexpr int NLD = (BM_CH * WD + BM_NT - 1) / BM_NT;
    static constexpr size_t LDS_BYTES = sizeof(unsigned) * BM_CH * PITCH + sizeof(double) * RING * BM_RP;
};

template <int R>
__global__ void __launch_bounds__(BM_NT) canny_blur_march(BlurMarchParams p)
{
    using G = BlurGeom<R>;
    constexpr int HL = G::HL, WD = G::WD, PITCH = G::PITCH, NQ = G::NQ, RING = G::RING, NLD = G::NLD;
    HIP_DYNAMIC_SHARED(double, smem_d)
    double *ring = smem_d;                                               // [RING][BM_RP], columns transposed by 8
    unsigned *raw = reinterpret_cast<unsigned *>(smem_d + RING * BM_RP);  // [BM_CH][PITCH]

    const int tid = threadIdx.x;
    // grid = (strips x segments of a frame, frames); XCD-aware order inside the frame (imgfd_xcd_tile): the strips left and right
    // of a strip, which share the 128-byte lines at its edges and its 2R halo columns, march on the same XCD (the kernel
    // fetched 4.1 B/px for 1 algorithmic when they were dealt round-robin)
    const int in_frame = (int)(p.xcd_order ? imgfd_xcd_tile(blockIdx.x, gridDim.x) : blockIdx.x);
    const int seg_i = in_frame / p.strips, strip_i = in_frame - seg_i * p.strips;
    const int x0 = strip_i * BM_TW;
    const int y0 = seg_i * p.seg_rows;
    const int nrows = min(p.ny, y0 + p.seg_rows) - y0;
    const int nchunks = (nrows + 2 * R + BM_CH - 1) / BM_CH;
    const int ybase = y0 - R;
    const unsigned char *in = p.in + (size_t)blockIdx.y * p.frame_stride;
    float *out = p.out + (size_t)blockIdx.y * p.nx * p.ny;

    // tile loads: slot l of a thread is always dword q[l] of tile row r[l]; only the image row moves (by BM_CH rows per
    // chunk, wrapping around the image), so the column offset and the LDS address are fixed up front and the row index
    // is advanced incrementally -- no division in the loop
    unsigned pre[NLD];
    int ld_row[NLD], ld_col[NLD], ld_lds[NLD];
#pragma unroll
    for (int l = 0; l < NLD; l++) {
        const int i = tid + l * BM_NT;
        const int r = i / WD, q = i - r * WD;
        ld_lds[l] = i < BM_CH * WD ? r * PITCH + q : -1;
        ld_row[l] = wrap_idx(ybase + r, p.ny);
        ld_col[l] = wrap_idx(x0 - HL + 4 * q, p.nx);  // aligned4: nx % 4 == 0, so a dword never straddles the wrap
    }
    const int row_step = BM_CH % p.ny;
    auto prefetch = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int l = 0; l < NLD; l++) {
            pre[l] = 0;
            if (ld_lds[l] >= 0) {
                const unsigned char *row = in + (size_t)ld_row[l] * p.row_stride;
                if (p.aligned4) {
                    pre[l] = *reinterpret_cast<const unsigned *>(row + ld_col[l]);
                } else {
                    unsigned v = 0;
#pragma unroll
                    for (int b = 0; b < 4; b++) {
                        v |= (unsigned)row[(ld_col[l] + b) % p.nx] << (8 * b);
                    }
                    pre[l] = v;
                }
                ld_row[l] += row_step;
                ld_row[l] = ld_row[l] >= p.ny ? ld_row[l] - p.ny : ld_row[l];
            }
        }
    };
    auto commit = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int l = 0; l < NLD; l++)
            if (ld_lds[l] >= 0) raw[ld_lds[l]] = pre[l];
    };

    prefetch();
    for (int chunk = 0; chunk < nchunks; chunk++) {
        commit();
        __syncthreads();
        if (chunk + 1 < nchunks) prefetch();

        // ---- row pass: thread = (row r, 8-pixel strip s)
        {
            const int r = tid >> 3, s = tid & 7;
            unsigned dw[NQ];
#pragma unroll
            for (int q = 0; q < NQ; q++) dw[q] = raw[r * PITCH + 2 * s + q];
            // byte k of the window = column x0 - HL + 8s + k; output o is centred on byte HL + o.  The 8 outputs are
            // independent chains of R+1 dependent operations (integer pair sum -> v_cvt_f64_u32 -> f64 fma): groups of
            // 4 advance together, tap by tap (one chain at a time was latency-bound: 15 cycles per dependent f64 op).
            double *dst = ring + ((chunk * BM_CH + r) & (RING - 1)) * BM_RP + s;  // output o of the strip at dst[8 * o]
            // every byte of the window becomes a double once (v_cvt_f32_ubyteN + v_cvt_f64_f32, both exact); the pair sums
            // byte[o-j] + byte[o+j] are then f64 adds of two integers <= 255 -- exact, the value an integer add followed by a
            // conversion gives, for 16 instead of 24 instructions per output
            double wd[NQ * 4];
#pragma unroll
            for (int k = 0; k < NQ * 4; k++) {
                wd[k] = (double)(float)((dw[k >> 2] >> (8 * (k & 3))) & 0xffu);
                IMGFD_OPAQUE(wd[k]);  // or the compiler folds the f64 adds back into integer adds + one conversion per pair
            }
#pragma unroll
            for (int o0 = 0; o0 < BM_PX; o0 += 4) {
                double acc[4];
#pragma unroll
                for (int g = 0; g < 4; g++) acc[g] = p.wx[0] * wd[HL + o0 + g];
#pragma unroll
                for (int j = 1; j <= R; j++) {
                    double pair[4];
#pragma unroll
                    for (int g = 0; g < 4; g++) pair[g] = wd[HL + o0 + g - j] + wd[HL + o0 + g + j];
#pragma unroll
                    for (int g = 0; g < 4; g++) acc[g] = __builtin_fma(p.wx[j], pair[g], acc[g]);
                }
#pragma unroll
                for (int g = 0; g < 4; g++) dst[8 * (o0 + g)] = acc[g];
            }
        }
        __syncthreads();

        // ---- column pass: thread = (8-row group g, column col)
        {
            const int g = __builtin_amdgcn_readfirstlane(tid >> 6), pos = tid & 63;  // the wave index: ring rows in SGPRs
            const int col = (pos & 7) * 8 + (pos >> 3);  // the column stored at ring position `pos`
            const int oi0 = chunk * BM_CH - 2 * R + BM_PX * g;  // first output row of the group, relative to y0
            if (oi0 + BM_PX > 0 && oi0 < nrows) {
                double acc[BM_PX];
#pragma unroll
                for (int o = 0; o < BM_PX; o++) acc[o] = 0.0;
#pragma unroll
                for (int k = 0; k < BM_PX + 2 * R; k++) {
                    const double v = ring[((oi0 + k) & (RING - 1)) * BM_RP + pos];
#pragma unroll
                    for (int o = 0; o < BM_PX; o++) {
                        const int j = k - R - o;
                        if (j >= -R && j <= R) acc[o] = __builtin_fma(p.wy[j < 0 ? -j : j], v, acc[o]);
                    }
                }
                const int gx = x0 + col;
                if (gx < p.nx) {
#pragma unroll
                    for (int o = 0; o < BM_PX; o++) {
                        const int oi = oi0 + o;
                        if (oi >= 0 && oi < nrows) IMGFD_OUT_STORE((float)acc[o], &out[(size_t)(y0 + oi) * p.nx + gx]);
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------ K10+K11
#define GN_TX 64  // tile width = one __ballot word
#ifndef GN_TY
#define GN_TY 16
#endif

#define GN_XO 4  // the LDS tile of the blurred image starts at x0-4 (16-byte aligned in the plane), y0-2

// gradient of the blurred image at (gx, gy), rcpp_canny.cpp:153-170, from the LDS tile `sb`.
// value() clamps coordinates (:38-62), so a neighbour outside the image stands for the border pixel; INSIDE = the
// whole tile neighbourhood lies in the image (no clamping needed).
struct CannyGrad { double h, v; };
template <bool INSIDE>
__device__ __forceinline__ CannyGrad canny_gradient(const double (*sb)[GN_TX + 2 * GN_XO + 4], int gx, int gy, int x0, int y0,
                                                    int nx, int ny, int accGrad)
{
    int xm, xp, xc, ym, yp, yc;
    if (INSIDE) {
        xc = gx - x0 + GN_XO; xm = xc - 1; xp = xc + 1;
        yc = gy - y0 + 2; ym = yc - 1; yp = yc + 1;
    } else {
        xm = max(gx - 1, 0) - x0 + GN_XO; xp = min(gx + 1, nx - 1) - x0 + GN_XO; xc = gx - x0 + GN_XO;
        ym = max(gy - 1, 0) - y0 + 2; yp = min(gy + 1, ny - 1) - y0 + 2; yc = gy - y0 + 2;
    }
    CannyGrad g;
    if (accGrad) {  // :157-163, evaluation order preserved
        g.h = 2 * (sb[yc][xp] - sb[yc][xm]) + sb[yp][xp] - sb[yp][xm] + sb[ym][xp] - sb[ym][xm];
        g.v = 2 * (sb[yp][xc] - sb[ym][xc]) + sb[yp][xp] - sb[ym][xp] + sb[yp][xm] - sb[ym][xm];
    } else {        // :167-169
        g.h = sb[yc][xp] - sb[yc][xm];
        g.v = sb[yp][xc] - sb[ym][xc];
    }
    return g;
}
// |g|: h and v are exact in f64 (sums of a few floats); sqrt(fma(h,h,v*v)) is within one ulp of hypot(h,v) (:172)
__device__ __forceinline__ double canny_mag(const CannyGrad &g) { return sqrt(__builtin_fma(g.h, g.h, g.v * g.v)); }
// |g| and 1/|g| together: one v_rsq_f64 seed and two coupled Newton steps (Goldschmidt) give sqrt(s) and 1/(2 sqrt(s)) to
// full double precision, a last residual step rounds sqrt(s) -- 11 double operations instead of a separate IEEE sqrt
// (~15) and division (~11).  The magnitude agrees with sqrt() to the last bit except for rare 1-ulp cases, the reciprocal
// to within an ulp: the same size of perturbation as dropping atan2/cos/sin already is (only exact ties can flip).
__device__ __forceinline__ double canny_mag_rcp(const CannyGrad &g, double *rcp)
{
    // a zero gradient is given the magnitude 1e-150 (s raised to the smallest value that keeps every step below in the
    // normal range) instead of branching around the iteration: it is below every threshold, adds nothing to a neighbour's
    // interpolation (x + 1e-150 == x for every other magnitude a float image can produce) and h * rcp = v * rcp = 0
    const double s = fmax(__builtin_fma(g.h, g.h, g.v * g.v), 1e-300);
    const double y0 = __builtin_amdgcn_rsq(s);
    double gq = s * y0, hq = 0.5 * y0;
    double r = __builtin_fma(-hq, gq, 0.5);
    gq = __builtin_fma(gq, r, gq);
    hq = __builtin_fma(hq, r, hq);
    r = __builtin_fma(-hq, gq, 0.5);
    gq = __builtin_fma(gq, r, gq);
    hq = __builtin_fma(hq, r, hq);
    const double d = __builtin_fma(-gq, gq, s);
    gq = __builtin_fma(d, hq, gq);
    *rcp = hq + hq;
    return gq;
}

// strong / marked bit planes: word (y, bx) covers pixels x = 64*bx .. 64*bx+63 of row y
template <bool INSIDE>
__device__ __forceinline__ void canny_grad_nms_tile(double (*sb)[GN_TX + 2 * GN_XO + 4], double (*sg)[GN_TX + 2 + 1],
                                                    const float *__restrict__ blur, unsigned long long *__restrict__ S,
                                                    unsigned long long *__restrict__ Wm, int nx, int ny, int words_per_row,
                                                    int accGrad, int low_thr, int high_thr, int bx, int by, int bz)
{
    constexpr int LW = GN_TX + 2 * GN_XO;  // 72 columns: x0-4 .. x0+67
    const int tid = threadIdx.x;
    const int x0 = bx * GN_TX, y0 = by * GN_TY;
    const float *pl = blur + (size_t)bz * nx * ny;
    // INSIDE tiles see x0-4 >= 0, x0+68 <= nx, y0-2 >= 0, y0+18 <= ny and a 16-byte aligned plane
    if (INSIDE) {
        for (int i = tid; i < (GN_TY + 4) * (LW / 4); i += 256) {
            const int r = i / (LW / 4), q = i - r * (LW / 4);
            // the blurred values are widened once here (the reference keeps them in doubles, rcpp_canny.cpp:145-146)
            const float4 v = *reinterpret_cast<const float4 *>(pl + (size_t)(y0 - 2 + r) * nx + (x0 - GN_XO + 4 * q));
            double *d = &sb[r][4 * q];
            d[0] = (double)v.x; d[1] = (double)v.y; d[2] = (double)v.z; d[3] = (double)v.w;
        }
    } else {
        // clamp-to-edge (extend(), rcpp_canny.cpp:38-55)
        for (int i = tid; i < (GN_TY + 4) * LW; i += 256) {
            const int r = i / LW, c = i - r * LW;
            const int gx = min(max(x0 + c - GN_XO, 0), nx - 1), gy = min(max(y0 + r - 2, 0), ny - 1);
            sb[r][c] = (double)pl[(size_t)gy * nx + gx];
        }
    }
    __syncthreads();
    // gradient magnitude: a thread owns one column of NQ consecutive tile rows (lane = column, wave wv = rows NQ*wv ..) and
    // keeps h, v and 1/|g| in registers; the one-pixel ring around the tile is shared out over the first threads.  A ring
    // position outside the image stands for the clamped pixel, whose own neighbourhood is clamped again: evaluate at
    // clamped coordinates.
    constexpr int NQ = GN_TY / 4;
    // the wave index as a scalar: row numbers, LDS row offsets and the mask-word addresses stay in SGPRs
    const int c = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), R0 = NQ * wv;
    CannyGrad own[NQ];
    double own_rcp[NQ];
    if (INSIDE) {
        // consecutive rows share their row terms: d(y) = b(y,x+1) - b(y,x-1) and e(y) = b(y,x-1) + 2 b(y,x) + b(y,x+1) give
        // h = 2 d(y) + d(y+1) + d(y-1), v = e(y+1) - e(y-1) (rcpp_canny.cpp:157-163).  The blurred values are floats a few
        // binades apart, every partial sum is exact in f64: the regrouping has the bits of the reference's left-to-right sum.
        const int xc = c + GN_XO;
        double d[NQ + 2], e[NQ + 2];
        if (accGrad) {  // workgroup-uniform: a branch, not selects
#pragma unroll
            for (int j = 0; j < NQ + 2; j++) {
                const double a = sb[R0 + 1 + j][xc - 1], b = sb[R0 + 1 + j][xc], cc = sb[R0 + 1 + j][xc + 1];
                d[j] = cc - a;
                e[j] = (a + cc) + 2 * b;
            }
#pragma unroll
            for (int q = 0; q < NQ; q++) {
                own[q].h = 2 * d[q + 1] + d[q + 2] + d[q];  // :157-159
                own[q].v = e[q + 2] - e[q];                 // :160-163
            }
        } else {
#pragma unroll
            for (int j = 0; j < NQ + 2; j++) {
                d[j] = sb[R0 + 1 + j][xc + 1] - sb[R0 + 1 + j][xc - 1];
                e[j] = sb[R0 + 1 + j][xc];
            }
#pragma unroll
            for (int q = 0; q < NQ; q++) {
                own[q].h = d[q + 1];         // :167
                own[q].v = e[q + 2] - e[q];  // :168-169
            }
        }
#pragma unroll
        for (int q = 0; q < NQ; q++) sg[R0 + q + 1][c + 1] = canny_mag_rcp(own[q], &own_rcp[q]);
    } else {
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            const int r = R0 + q;
            own[q] = canny_gradient<false>(sb, min(x0 + c, nx - 1), min(y0 + r, ny - 1), x0, y0, nx, ny, accGrad);
            sg[r + 1][c + 1] = canny_mag_rcp(own[q], &own_rcp[q]);
        }
    }
    constexpr int RING = 2 * (GN_TX + 2) + 2 * GN_TY;
    if (tid < RING) {
        int r, cc;
        if (tid < GN_TX + 2) { r = 0; cc = tid; }
        else if (tid < 2 * (GN_TX + 2)) { r = GN_TY + 1; cc = tid - (GN_TX + 2); }
        else { const int k = tid - 2 * (GN_TX + 2); r = 1 + (k >> 1); cc = (k & 1) ? GN_TX + 1 : 0; }
        const int gx = INSIDE ? x0 + cc - 1 : min(max(x0 + cc - 1, 0), nx - 1), gy = INSIDE ? y0 + r - 1 : min(max(y0 + r - 1, 0), ny - 1);
        double unused;
        sg[r][cc] = canny_mag_rcp(canny_gradient<INSIDE>(sb, gx, gy, x0, y0, nx, ny, accGrad), &unused);
    }
    __syncthreads();
    // a zero gradient carries the magnitude 1e-150 (canny_mag_rcp): never a maximum, whatever the threshold -- like the
    // reference's 0 <= prev.  (A nonzero gradient of a float image is above 1e-45.)
    const double low = fmax((double)low_thr, 1e-100);
#pragma unroll
    for (int q = 0; q < NQ; q++) {  // one wave per tile row: the ballot is the mask word
        const int r = R0 + q;
        const int gx = x0 + c, gy = y0 + r;
        int o = 0;
        if (INSIDE || (gx < nx && gy < ny)) {
            const double now = sg[r + 1][c + 1];
            // unit direction (cos t, sin t) with t = atan2(v,h) (:69-70,173); atan2(0,0) = 0 -> (1,0)
            // (a zero gradient has own_rcp = 0: the direction comes out as (0,0) instead of (1,0), both taps then read the
            // pixel itself and 0 <= 0 rejects it -- as atan2(0,0)'s direction does, since no magnitude is negative)
            const double ux = own[q].h * own_rcp[q], uy = own[q].v * own_rcp[q];
            // bilin(), :65-85, at (c,r) -/+ (ux,uy): x1 = floor(xt) is -1 or 0 (for xt == 1 exactly the far tap has weight
            // 0, so x1 = 0 gives the same sum), hence the weights (x2 - xt, xt - x1) are (1 - |xt|, |xt|) for xt >= 0 and
            // (|xt|, 1 - |xt|) for xt < 0 -- the same two numbers for both taps: the pixel's own column/row always
            // weighs 1 - |.|, the neighbour on the side of sign(xt) weighs |.|.  Every product and two-term sum below
            // is one of bilin()'s own (at most with its two terms swapped), so the value has bilin()'s bits for this
            // (ux, uy) -- with 19 instead of 26 double operations for the two taps.
            const double ax = fabs(ux), ay = fabs(uy), bx = 1.0 - ax, by = 1.0 - ay;
            // the tap at +(ux,uy) looks towards sign(ux), sign(uy); the tap at -(ux,uy) the other way.  (For a zero
            // component bilin() takes the +1 side in both taps; its weight is 0 there and every magnitude is finite,
            // so reading the opposite neighbour instead adds the same +0.)
            constexpr int SGP = GN_TX + 2 + 1;  // row pitch of sg
            const double *ctr = &sg[r + 1][c + 1];
            const int sx = ux < 0 ? -1 : 1, sy = uy < 0 ? -SGP : SGP;
            const double own_term = bx * now;
            double val[2];
#pragma unroll
            for (int d = 0; d < 2; d++) {
                const int ox = d ? sx : -sx, oy = d ? sy : -sy;
                const double g_own = own_term + ax * ctr[ox];
                const double g_nb = bx * ctr[oy] + ax * ctr[oy + ox];
                val[d] = by * g_own + ay * g_nb;
            }
            const double prev = val[0], next = val[1];
            if ((now <= prev) || (now <= next) || (now <= low)) o = 0;  // maxima(), :88-106
            else if (now >= (double)high_thr) o = 2;
            else o = 1;
        }
        const unsigned long long strong = __ballot(o == 2), marked = __ballot(o >= 1);
        if (c == 0 && (INSIDE || gy < ny)) {
            const size_t w = ((size_t)bz * ny + gy) * words_per_row + bx;
            S[w] = strong;
            Wm[w] = marked;
        }
    }
}

__global__ void __launch_bounds__(256) canny_grad_nms(const float *__restrict__ blur, unsigned long long *__restrict__ S,
                                                      unsigned long long *__restrict__ Wm, int nx, int ny,
                                                      int words_per_row, int accGrad, int low_thr, int high_thr, int vec4,
                                                      unsigned *__restrict__ sweep_flags, int n_sweep_flags, int tiles_x, int xcd_order,
                                                      unsigned long long *__restrict__ counts)
{
    __shared__ __attribute__((aligned(16))) double sb[GN_TY + 4][GN_TX + 2 * GN_XO + 4];
    __shared__ double sg[GN_TY + 2][GN_TX + 2 + 1];
    // grid = (tiles of a frame, frames), XCD-aware tile order inside the frame (imgfd_xcd_tile; tile = row * tiles_x + column).
    // The 72 floats a tile row needs straddle three or four 128-byte lines, two of them shared with the neighbours left and
    // right: dealt round-robin to the XCDs, every tile fetched them from HBM itself -- 9.9 B/px for 4 algorithmic; 4.00 now
    // (round 4, profiles/r04/xcd_tile_order.txt)
    const int in_frame = (int)(xcd_order ? imgfd_xcd_tile(blockIdx.x, gridDim.x) : blockIdx.x), bz = blockIdx.y;
    const int by = in_frame / tiles_x, bx = in_frame - by * tiles_x;
    const int x0 = bx * GN_TX, y0 = by * GN_TY;
    // workgroup-uniform: interior tiles skip every clamp and fetch the blurred tile as float4s
    const bool inside = vec4 && x0 - GN_XO >= 0 && x0 + GN_TX + GN_XO <= nx && y0 - 2 >= 0 && y0 + GN_TY + 2 <= ny;
    if (inside) canny_grad_nms_tile<true>(sb, sg, blur, S, Wm, nx, ny, words_per_row, accGrad, low_thr, high_thr, bx, by, bz);
    else canny_grad_nms_tile<false>(sb, sg, blur, S, Wm, nx, ny, words_per_row, accGrad, low_thr, high_thr, bx, by, bz);
    // The launches behind this kernel start from cleared words: the sweeps' "changed" flags, per frame the last sweep that changed
    // it, the arrival counter of canny_finish's frame barriers and pixels_nonzero (a memset of their own was 5 us of a single
    // frame's critical path).  At the END of the kernel: in front of the tile code the 64-bit store through `counts` cost the
    // mask-word addresses their scalar registers (68 vector instructions more per workgroup, 882 against 830 us per 32 frames).
    if (in_frame == 0 && bz == 0 && (int)threadIdx.x < n_sweep_flags) sweep_flags[threadIdx.x] = 0;
    if (in_frame == 0 && threadIdx.x == 0) {
        sweep_flags[n_sweep_flags + bz] = 0;              // per frame: the last sweep that changed it
        sweep_flags[n_sweep_flags + gridDim.y + bz] = 0;  // ... and the arrival counter of canny_finish's frame barriers
        if (bz == 0) sweep_flags[n_sweep_flags + 2 * gridDim.y] = 0;  // ... and its ticket counter
        counts[bz] = 0;                                   // pixels_nonzero: the expansion adds to it
    }
}

// ------------------------------------------------------------------ K12
__device__ __forceinline__ unsigned long long brev64(unsigned long long x)
{
    return ((unsigned long long)__brev((unsigned)x) << 32) | (unsigned long long)__brev((unsigned)(x >> 32));
}
// all bits of m reachable from the seeds p (a subset of m) through runs of consecutive ones in m
__device__ __forceinline__ unsigned long long flood_runs(unsigned long long m, unsigned long long p)
{
    const unsigned long long up = m & ~(m + p);
    const unsigned long long rm = brev64(m), rp = brev64(p);
    const unsigned long long dn = brev64(rm & ~(rm + rp));
    return up | dn | p;
}
__device__ __forceinline__ unsigned long long dilate_h(unsigned long long s, unsigned long long left, unsigned long long right)
{
    return s | (s << 1) | (s >> 1) | (left >> 63) | (right << 63);
}

// value of the lane above / below (row y-1 / y+1 of the tile): one DPP wavefront shift per dword on gfx950 (no LDS
// round trip like ds_bpermute); lanes 0 / 63 receive 0 and are overridden with the halo rows by the caller
__device__ __forceinline__ unsigned long long lane_above(unsigned long long v)
{
    // bound_ctrl: a lane without a source reads 0 -- no "old" value has to be put into the destination first
    const unsigned lo = __builtin_amdgcn_mov_dpp((unsigned)v, 0x138 /* wave_shr:1 */, 0xf, 0xf, true);
    const unsigned hi = __builtin_amdgcn_mov_dpp((unsigned)(v >> 32), 0x138, 0xf, 0xf, true);
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned long long lane_below(unsigned long long v)
{
    const unsigned lo = __builtin_amdgcn_mov_dpp((unsigned)v, 0x130 /* wave_shl:1 */, 0xf, 0xf, true);
    const unsigned hi = __builtin_amdgcn_mov_dpp((unsigned)(v >> 32), 0x130, 0xf, 0xf, true);
    return ((unsigned long long)hi << 32) | lo;
}

#define HY_SWEEPS_MAX 32  // flags[]: one word per sweep, one word per frame ("the last sweep that changed this frame" + 1), one more per frame (arrivals at canny_finish's frame barriers)
// a wave turns its own LDS patch round (written lane = (row, word), read lane = row): program order inside the wave is the
// only synchronisation; the fences keep the compiler from moving one lane's read over another lane's write
__device__ __forceinline__ void hyst_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- sweeps.  A wave owns a tile of HW words x 64 rows (lane = row, HW words of S and W per lane + the halo words left and
// right) and iterates it to its fixpoint in registers: the 3x3 dilation is shifts + two DPP wave shifts, runs of W inside a
// word are flooded in O(1) by carry propagation (flood_runs), only words whose neighbourhood moved in the previous pass are
// visited.  A WORKGROUP owns BX x BY such tiles (a wave each) and iterates them to the fixpoint of the whole block inside one
// launch: what crosses a tile outline inside the block travels through LDS -- a "board" on which a wave that changed posts
// its first and last row and its first and last column -- and costs one workgroup barrier per round; what crosses the block's
// outline waits for the next launch.  Launches alternate between two groupings of the same tiles (the second starts half a
// block up and left), so every tile outline lies inside a block in one of them: a chain that wobbles along an outline of one
// grouping is an LDS matter in the other (the single 4K bench frame: 8 working launches with a tile per wave and no exchange,
// round 4 -> 3-5; profiles/r05/canny_block_sweeps.txt).  S only grows and every value on the board is a lower bound of its
// owner's state, so a reader that meets a newer posting than the barrier promises is only better informed; the block stops when
// a whole round posted nothing (then every wave has read a complete board and found nothing to add: the block's fixpoint under
// its outer halo).  flags[sweep] is raised when any tile changed; a launch whose predecessor changed nothing returns at once,
// so the host queues a fixed number of launches without ever reading a flag back.  act[] holds one byte per tile and sweep
// parity, "this tile changed in that sweep": a tile can only change if itself or one of its 8 neighbours changed in the
// previous sweep, so a block none of whose tiles (and the ring of tiles around them) changed leaves at once.
// The tile comes in through LDS: in registers a lane owns a ROW, and rows lie wpr * 8 bytes apart -- loaded lane = row, every
// 8-byte load of a wave touched 64 different 128-byte lines (PMC, round 4: 69 % of a sweep's wave cycles waited for them);
// loaded lane = (row, word) an instruction touches one line per row, and the wave's own LDS patch turns the tile round.
template <int HW>
struct HystBoard {
    unsigned long long top[HW], bot[HW];  // rows 0 and 63 of the tile
    unsigned long long lm, rm;            // bit r: pixel (row r, first column) / (row r, last column)
};
template <int HW, int BX, int BY>
__global__ void __launch_bounds__(64 * BX * BY) canny_hyst_block(unsigned long long *__restrict__ S, const unsigned long long *__restrict__ Wm, int wpr,
                                                                 int ny, int tiles_x, int tiles_y, int blocks_x, int shift_x, int shift_y,
                                                                 unsigned *__restrict__ flags, int sweep, unsigned char *__restrict__ act, int prio)
{
    if (sweep > 0 && flags[sweep - 1] == 0) return;  // the sweep before changed nothing anywhere
    // a sweep is a chain of dependent instructions in a few waves: where it shares a SIMD with another kernel's waves (the
    // Harris chain on the other stream) it goes first -- it asks for a fraction of the issue slots and is the frame's critical path
    if (prio) __builtin_amdgcn_s_setprio(3);
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int ntiles = tiles_x * tiles_y;
    // The block grid of this launch starts (shift_x, shift_y) tiles up and left of the frame's corner: launches alternate
    // between two groupings of the same tiles, so that every tile outline lies INSIDE a block in one of them -- a chain that
    // wobbles along an outline of one grouping (one launch per crossing) is an LDS matter in the other.
    const int bx0 = (int)(blockIdx.x % blocks_x) * BX - shift_x, by0 = (int)(blockIdx.x / blocks_x) * BY - shift_y;
    const int wx = wv % BX, wy = wv / BX;
    const int tx = bx0 + wx, ty = by0 + wy, tile = ty * tiles_x + tx;
    const bool live = tx >= 0 && tx < tiles_x && ty >= 0 && ty < tiles_y;  // wave-uniform; a wave without a tile still keeps the barriers
    unsigned char *act_w = act + ((size_t)(sweep & 1) * gridDim.y + blockIdx.y) * ntiles;            // written by this sweep
    const unsigned char *act_r = act + ((size_t)((sweep + 1) & 1) * gridDim.y + blockIdx.y) * ntiles;  // previous sweep
    if (sweep > 0) {  // a tile can only change if itself or one of its 8 neighbours changed in the previous sweep: the block's tiles and the ring around them
        static_assert((BX + 2) * (BY + 2) <= 64, "one lane per tile of the neighbourhood");
        bool near = false;
        if (lane < (BX + 2) * (BY + 2)) {
            const int nx_ = bx0 - 1 + lane % (BX + 2), ny_ = by0 - 1 + lane / (BX + 2);
            near = nx_ >= 0 && nx_ < tiles_x && ny_ >= 0 && ny_ < tiles_y && act_r[ny_ * tiles_x + nx_] != 0;
        }
        if (!__any(near)) {  // the same answer in every wave of the workgroup
            if (live && lane == 0) act_w[tile] = 0;
            return;
        }
    }
    const int w0 = tx * HW;
    unsigned long long *Sf = S + (size_t)blockIdx.y * ny * wpr;
    const unsigned long long *Wf = Wm + (size_t)blockIdx.y * ny * wpr;
    constexpr int LPR = HW + 2 <= 4 ? 4 : 8, SP = (HW + 2) | 1, WP = HW | 1;  // lanes per row of a load instruction; odd pitches: the lane = row reads hit distinct banks
    HIP_DYNAMIC_SHARED(unsigned long long, hb_dyn)
    unsigned long long *ls = hb_dyn + (size_t)wv * 64 * (SP + WP), *lw = ls + 64 * SP;
    __shared__ HystBoard<HW> board[BX * BY];
    __shared__ unsigned rflag[3];
    unsigned long long s[HW + 2], w[HW];
    if (live) {
        const int c = lane % LPR, rsub = lane / LPR;
#pragma unroll
        for (int i = 0; i < LPR; i++) {
            const int r = i * (64 / LPR) + rsub, yy = ty * 64 + r, wi = w0 - 1 + c;
            if (c < HW + 2) ls[r * SP + c] = (yy < ny && wi >= 0 && wi < wpr) ? Sf[(size_t)yy * wpr + wi] : 0ull;
        }
        constexpr int LPW = HW <= 2 ? 2 : 4;
        const int cw_ = lane % LPW, rw_ = lane / LPW;
#pragma unroll
        for (int i = 0; i < LPW; i++) {
            const int r = i * (64 / LPW) + rw_, yy = ty * 64 + r, wi = w0 + cw_;
            if (cw_ < HW) lw[r * WP + cw_] = (yy < ny && wi < wpr) ? Wf[(size_t)yy * wpr + wi] : 0ull;
        }
    }
    if (threadIdx.x < 3) rflag[threadIdx.x] = 0;
    if (lane == 0) {
#pragma unroll
        for (int q = 0; q < HW; q++) board[wv].top[q] = board[wv].bot[q] = 0ull;
        board[wv].lm = board[wv].rm = 0ull;
    }
    hyst_wave_sync();
    bool todo = false;
#pragma unroll
    for (int q = 0; q < HW + 2; q++) s[q] = live ? ls[lane * SP + q] : 0ull;
#pragma unroll
    for (int q = 0; q < HW; q++) {
        w[q] = live ? lw[lane * WP + q] : 0ull;
        todo = todo || (w[q] & ~s[q + 1]) != 0ull;
    }
    todo = __any(todo) != 0;  // a tile without a marked-but-unlit pixel never changes: its outline is what memory holds
    // halo rows above / below as the launch finds them in memory: lanes 0..HW+1 fetch the words, kept wave-uniform
    unsigned long long t[HW + 2], b[HW + 2];
    {
        unsigned long long trow = 0ull, brow = 0ull;
        const int wi = w0 - 1 + lane;
        const int yt = ty * 64 - 1, yb = ty * 64 + 64;
        if (todo && lane < HW + 2 && wi >= 0 && wi < wpr) {
            if (yt >= 0) trow = Sf[(size_t)yt * wpr + wi];
            if (yb < ny) brow = Sf[(size_t)yb * wpr + wi];
        }
#pragma unroll
        for (int q = 0; q < HW + 2; q++) {
            t[q] = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(trow >> 32), q) << 32) |
                   (unsigned)__builtin_amdgcn_readlane((int)(unsigned)trow, q);
            b[q] = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(brow >> 32), q) << 32) |
                   (unsigned)__builtin_amdgcn_readlane((int)(unsigned)brow, q);
        }
    }
    const bool first = lane == 0, last = lane == 63;
    bool any = false, run = todo;
    int round = 0;
    __syncthreads();  // board and round flags are cleared
    for (;; round++) {
        bool changed = false;
        if (run) {  // wave-uniform
            unsigned long long top_d[HW], bot_d[HW];
#pragma unroll
            for (int q = 0; q < HW; q++) {
                top_d[q] = dilate_h(t[q + 1], t[q], t[q + 2]);
                bot_d[q] = dilate_h(b[q + 1], b[q], b[q + 2]);
            }
            unsigned prev = (1u << HW) - 1u;
            // the tile's own fixpoint under its present halo.  Bit q of `prev` / `cur` = "word q of some row changed in the previous /
            // in this pass" (wave-uniform): a chain that climbs through one word does not drag the others through every pass.  A
            // word's dilation is taken when the word is visited (words to its left already hold this pass's additions: the same
            // fixpoint, reached no later); lane_above / lane_below give 0 to lanes 0 / 63, whose neighbours are the halo rows
            for (;;) {
                unsigned cur = 0;
#pragma unroll
                for (int q = 0; q < HW; q++) {
                    const unsigned around = ((7u << q) >> 1) & ((1u << HW) - 1u);
                    if (!((prev & around) || (q > 0 && (cur & (1u << (q - 1)))))) continue;
                    const unsigned long long d = dilate_h(s[q + 1], s[q], s[q + 2]);
                    const unsigned long long halo = first ? top_d[q] : (last ? bot_d[q] : 0ull);
                    const unsigned long long cand = w[q] & ~s[q + 1] & (d | lane_above(d) | lane_below(d) | halo);
                    if (__any(cand != 0ull)) {
                        s[q + 1] |= flood_runs(w[q], cand);
                        cur |= 1u << q;
                    }
                }
                if (!cur) break;
                changed = true;
                prev = cur;
            }
        }
        if (changed) {  // post the outline
            any = true;
            const unsigned long long lm = __ballot((s[1] & 1ull) != 0ull), rm = __ballot((s[HW] >> 63) != 0ull);
            if (first) {
#pragma unroll
                for (int q = 0; q < HW; q++) board[wv].top[q] = s[q + 1];
                board[wv].lm = lm;
                board[wv].rm = rm;
                rflag[round % 3] = 1u;
            }
            if (last) {
#pragma unroll
                for (int q = 0; q < HW; q++) board[wv].bot[q] = s[q + 1];
            }
        }
        if (threadIdx.x == 0) rflag[(round + 1) % 3] = 0u;  // read last behind the barrier of round - 2: nobody is still there
        __syncthreads();
        if (rflag[round % 3] == 0u) break;  // a whole round without a posting
        run = false;
        if (todo) {
            // what the neighbours inside the block have posted, OR-ed onto the halo
            bool grew = false;
            auto merge = [&grew](unsigned long long &dst, unsigned long long v) {
                v |= dst;
                grew = grew || v != dst;
                dst = v;
            };
            auto uni = [](unsigned long long v) {  // every lane read the same LDS word: keep it in scalar registers
                return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32)) << 32) |
                       (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
            };
            if (wy > 0) {
                const HystBoard<HW> *nb = &board[wv - BX];
#pragma unroll
                for (int q = 0; q < HW; q++) merge(t[q + 1], uni(nb->bot[q]));
                if (wx > 0) merge(t[0], uni(nb[-1].bot[HW - 1]) & (1ull << 63));
                if (wx < BX - 1) merge(t[HW + 1], uni(nb[1].bot[0]) & 1ull);
            }
            if (wy < BY - 1) {
                const HystBoard<HW> *nb = &board[wv + BX];
#pragma unroll
                for (int q = 0; q < HW; q++) merge(b[q + 1], uni(nb->top[q]));
                if (wx > 0) merge(b[0], uni(nb[-1].top[HW - 1]) & (1ull << 63));
                if (wx < BX - 1) merge(b[HW + 1], uni(nb[1].top[0]) & 1ull);
            }
            bool side = false;
            if (wx > 0) {
                const unsigned long long add = ((uni(board[wv - 1].rm) >> lane) & 1ull) << 63;
                side = side || (add & ~s[0]) != 0ull;
                s[0] |= add;
            }
            if (wx < BX - 1) {
                const unsigned long long add = (uni(board[wv + 1].lm) >> lane) & 1ull;
                side = side || (add & ~s[HW + 1]) != 0ull;
                s[HW + 1] |= add;
            }
            run = grew || __any(side);
        }
    }
    if (any) {  // wave-uniform: rows back to the LDS patch, then lane = (row, word) stores
        hyst_wave_sync();
#pragma unroll
        for (int q = 0; q < HW; q++) lw[lane * WP + q] = s[q + 1];
        hyst_wave_sync();
        constexpr int LPW = HW <= 2 ? 2 : 4;
        const int cw_ = lane % LPW, rw_ = lane / LPW;
#pragma unroll
        for (int i = 0; i < LPW; i++) {
            const int r = i * (64 / LPW) + rw_, yy = ty * 64 + r, wi = w0 + cw_;
            if (cw_ < HW && yy < ny && wi < wpr) Sf[(size_t)yy * wpr + wi] = lw[r * WP + cw_];
        }
    }
    if (live && lane == 0) act_w[tile] = any ? 1 : 0;
    if (threadIdx.x == 0 && round > 0) {  // the loop left in round r: rounds 0 .. r-1 posted
        flags[sweep] = 1u;  // plain stores: every writer stores the same value
        flags[HY_SWEEPS_MAX + blockIdx.y] = (unsigned)sweep + 1u;
    }
}

// ---- what the queued sweeps leave: union-find
// A sweep carries the strong label across one tile outline, so a chain of weak pixels that winds through the frame can outlast
// any fixed number of sweeps (one edge snaking through a 4K frame crosses sixteen thousand outlines in a row).  When the last
// queued sweep still changed a frame, two kernels finish the frame WITHOUT walking chains: the runs of still-unlit marked
// pixels (a run = consecutive bits of one 64-bit word of W & ~S) are the nodes of a forest in the dead blur plane (4 bytes per
// pixel, a node's parent at its first pixel).  canny_uf_init makes every run its own root; canny_uf_merge unites runs that
// touch (union by smaller label, lock-free: atomicMin, path halving) and unites a run that touches a strong pixel with the
// label LIT = 0; canny_expand_count, which writes the edge map anyway, counts the runs of such a frame whose root is LIT as
// strong.  Work and depth depend on the number of runs, not on how they are chained; all workgroups of a frame that the last
// sweep left alone (the case in practice) leave at once.
#define UF_NT 256
__device__ __forceinline__ unsigned uf_load(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }  // never a stale L1 line
__device__ __forceinline__ unsigned uf_find(const unsigned *P, unsigned x)
{
    while (x != 0u) {
        const unsigned up = uf_load(P + (x - 1u));
        if (up == x) break;
        x = up;
    }
    return x;
}
// the same walk with path halving: every node passed is re-hung below its grandparent (labels only ever fall along a path, so
// an atomicMin keeps the forest a forest whatever the other threads do meanwhile)
__device__ __forceinline__ unsigned uf_find_halving(unsigned *P, unsigned x)
{
    while (x != 0u) {
        const unsigned up = uf_load(P + (x - 1u));
        if (up == x) break;
        if (up != 0u) {
            const unsigned upup = uf_load(P + (up - 1u));
            if (upup != up) atomicMin(P + (x - 1u), upup);
            x = upup == up ? up : upup;
        } else {
            x = 0u;
        }
    }
    return x;
}
__device__ __forceinline__ void uf_union(unsigned *P, unsigned a, unsigned b)
{
    for (;;) {
        a = uf_find_halving(P, a);
        b = uf_find_halving(P, b);
        if (a == b) return;
        if (a < b) { const unsigned t = a; a = b; b = t; }  // hang the larger root below the smaller label
        const unsigned old = atomicMin(P + (a - 1u), b);
        if (old == a) return;  // a was still a root: linked
        a = old;               // somebody linked a meanwhile: go on from there
    }
}
// the run of `m` that holds bit `bit` (set in m): its lowest bit
__device__ __forceinline__ int uf_run_start(unsigned long long m, int bit)
{
    const unsigned long long below = ~m & ((bit == 63 ? ~0ull : ((2ull << bit) - 1ull)));  // zeros of m at or below `bit`
    return below ? 64 - __clzll((long long)below) : 0;
}
// the run of `m` that starts at bit b (set in m, bit b-1 clear): m's ones from b up to the first zero
__device__ __forceinline__ unsigned long long uf_run_from(unsigned long long m, int b)
{
    const unsigned long long zeros_above = b == 63 ? 0ull : (~m >> (b + 1)) << (b + 1);
    return zeros_above ? (((zeros_above & (0ull - zeros_above)) - 1ull) & ~((1ull << b) - 1ull)) : (~0ull << b);
}
struct UfFrame {  // the planes of one frame
    const unsigned long long *Sf;
    const unsigned long long *Wf;
    unsigned *P;
    int wpr, nx, ny;
    __device__ __forceinline__ unsigned label(int y, int wi, int bit) const { return (unsigned)(y * nx + wi * 64 + bit) + 1u; }
    __device__ __forceinline__ unsigned long long lit(int y, int wi) const
    {
        return (y < 0 || y >= ny || wi < 0 || wi >= wpr) ? 0ull : Sf[(size_t)y * wpr + wi];
    }
    __device__ __forceinline__ unsigned long long rem(int y, int wi) const  // marked and not (yet) strong
    {
        if (y < 0 || y >= ny || wi < 0 || wi >= wpr) return 0ull;
        const size_t i = (size_t)y * wpr + wi;
        return Wf[i] & ~Sf[i];
    }
};
// flags layout (see HY_SWEEPS_MAX): [sweep flags][per frame: last sweep that changed it + 1]
// every run of W & ~S of frame `frame` becomes its own root (workgroup `bid` of `nblk` shares the frame's words)
__device__ __forceinline__ void uf_init_frame(const unsigned long long *__restrict__ S, const unsigned long long *__restrict__ Wm, int wpr, int nx, int ny,
                                              unsigned *__restrict__ parents, int frame, int bid, int nblk)
{
    const size_t plane = (size_t)ny * wpr;
    const unsigned long long *Sf = S + frame * plane, *Wf = Wm + frame * plane;
    unsigned *P = parents + (size_t)frame * nx * ny;
    for (long i = (long)bid * UF_NT + threadIdx.x; i < (long)plane; i += (long)nblk * UF_NT) {
        const unsigned long long m = Wf[i] & ~Sf[i];
        if (!m) continue;
        const int y = (int)(i / wpr), wi = (int)(i - (long)y * wpr);
        unsigned long long starts = m & ~(m << 1);
        while (starts) {  // every run is its own root
            const int b = __ffsll((long long)starts) - 1;
            starts &= starts - 1ull;
            const unsigned l = (unsigned)(y * nx + wi * 64 + b) + 1u;
            P[l - 1u] = l;
        }
    }
}
__global__ void __launch_bounds__(UF_NT) canny_uf_init(const unsigned long long *__restrict__ S, const unsigned long long *__restrict__ Wm,
                                                      int wpr, int nx, int ny, unsigned *__restrict__ parents, const unsigned *__restrict__ flags,
                                                      unsigned last_sweep)
{
    if (flags[HY_SWEEPS_MAX + blockIdx.y] != last_sweep) return;  // the last queued sweep left this frame alone: converged
    uf_init_frame(S, Wm, wpr, nx, ny, parents, blockIdx.y, blockIdx.x, gridDim.x);
}
// unite what touches (8-neighbourhood): the run to the left in the row, the runs of the row above, and LIT
__device__ __forceinline__ void uf_merge_frame(const unsigned long long *__restrict__ S, const unsigned long long *__restrict__ Wm, int wpr, int nx, int ny,
                                               unsigned *__restrict__ parents, int frame, int bid, int nblk)
{
    const size_t plane = (size_t)ny * wpr;
    UfFrame f{S + frame * plane, Wm + frame * plane, parents + (size_t)frame * nx * ny, wpr, nx, ny};
    unsigned *P = f.P;
    for (long i = (long)bid * UF_NT + threadIdx.x; i < (long)plane; i += (long)nblk * UF_NT) {
        const unsigned long long m = f.Wf[i] & ~f.Sf[i];
        if (!m) continue;
        const int y = (int)(i / wpr), wi = (int)(i - (long)y * wpr);
        const unsigned long long up_m = f.rem(y - 1, wi), up_l = f.rem(y - 1, wi - 1), up_r = f.rem(y - 1, wi + 1), left_m = f.rem(y, wi - 1);
        const unsigned long long s_any = f.lit(y - 1, wi) | f.lit(y, wi) | f.lit(y + 1, wi);
        const bool sl = ((f.lit(y - 1, wi - 1) | f.lit(y, wi - 1) | f.lit(y + 1, wi - 1)) >> 63) != 0ull;  // a strong pixel left of column 0
        const bool sr = ((f.lit(y - 1, wi + 1) | f.lit(y, wi + 1) | f.lit(y + 1, wi + 1)) & 1ull) != 0ull;  // right of column 63
        unsigned long long starts = m & ~(m << 1);
        while (starts) {
            const int b = __ffsll((long long)starts) - 1;
            starts &= starts - 1ull;
            const unsigned long long run = uf_run_from(m, b);
            const unsigned long long wide = run | (run << 1) | (run >> 1);
            const bool at0 = (run & 1ull) != 0ull, at63 = (run >> 63) != 0ull;
            const unsigned me = f.label(y, wi, b);
            if ((s_any & wide) || (at0 && sl) || (at63 && sr)) uf_union(P, me, 0u);
            if (at0 && (left_m >> 63)) uf_union(P, me, f.label(y, wi - 1, uf_run_start(left_m, 63)));
            unsigned long long hit = up_m & wide;
            while (hit) {  // one union per run of the row above that the dilated run meets
                const int st = uf_run_start(up_m, __ffsll((long long)hit) - 1);
                uf_union(P, me, f.label(y - 1, wi, st));
                hit &= ~uf_run_from(up_m, st);
            }
            if (at0 && (up_l >> 63)) uf_union(P, me, f.label(y - 1, wi - 1, uf_run_start(up_l, 63)));
            if (at63 && (up_r & 1ull)) uf_union(P, me, f.label(y - 1, wi + 1, 0));
        }
    }
}

__global__ void __launch_bounds__(UF_NT) canny_uf_merge(const unsigned long long *__restrict__ S, const unsigned long long *__restrict__ Wm,
                                                       int wpr, int nx, int ny, unsigned *__restrict__ parents, const unsigned *__restrict__ flags,
                                                       unsigned last_sweep)
{
    if (flags[HY_SWEEPS_MAX + blockIdx.y] != last_sweep) return;
    uf_merge_frame(S, Wm, wpr, nx, ny, parents, blockIdx.y, blockIdx.x, gridDim.x);
}

#define EXP_BLOCKS 128
#define CANNY_REPORTS 4  /* reports of working sweeps kept per context (pinned: sequence number, value) */
static_assert(CANNY_REPORTS == sizeof(imgfd_ctx::canny_report_queued) / sizeof(unsigned), "one host slot per report slot");
// rcpp_canny.cpp:226-243: the edge map as 0/255 bytes and its number of non-zero pixels; a workgroup walks every
// EXP_BLOCKS-th row and adds its count once
// A frame that went through the union-find kernels (uf != nullptr and its flag says so): the runs of W & ~S whose root is LIT
// count as strong (plain loads of the forest: it was finished by the previous kernel).
__device__ __forceinline__ void expand_count_frame(const unsigned long long *__restrict__ S, int wpr, unsigned char *__restrict__ edges, int nx, int ny,
                                                   unsigned long long *__restrict__ counts, const unsigned long long *__restrict__ Wm,
                                                   const unsigned *__restrict__ parents, bool united, int frame, int bid, int nblk)
{
    __shared__ unsigned wsum[4];
    const bool vec = (nx & 15) == 0 && (reinterpret_cast<size_t>(edges) & 15) == 0;
    const unsigned *P = parents + (size_t)frame * nx * ny;
    unsigned c = 0;
    for (int y = bid; y < ny; y += nblk) {
        for (int x = 16 * (int)threadIdx.x; x < nx; x += 16 * 256) {
            unsigned long long word = S[((size_t)frame * ny + y) * wpr + (x >> 6)];
            if (united) {
                const unsigned long long m = Wm[((size_t)frame * ny + y) * wpr + (x >> 6)] & ~word;
                unsigned long long starts = m & ~(m << 1);
                while (starts) {
                    const int b = __ffsll((long long)starts) - 1;
                    starts &= starts - 1ull;
                    unsigned r = (unsigned)(y * nx + (x >> 6) * 64 + b) + 1u;
                    while (r != 0u && P[r - 1u] != r) r = P[r - 1u];
                    if (r == 0u) word |= uf_run_from(m, b);
                }
            }
            unsigned bits = (unsigned)(word >> (x & 63)) & 0xffffu;
            if (x + 16 > nx) bits &= (1u << (nx - x)) - 1u;
            c += (unsigned)__popc(bits);
            unsigned v[4];
#pragma unroll
            for (int k = 0; k < 4; k++) v[k] = ((((bits >> (4 * k)) & 0xfu) * 0x00204081u) & 0x01010101u) * 0xffu;
            unsigned char *dst = edges + ((size_t)frame * ny + y) * nx + x;
            if (vec) {
                typedef unsigned v4u __attribute__((vector_size(16)));
                IMGFD_OUT_STORE((v4u{v[0], v[1], v[2], v[3]}), reinterpret_cast<v4u *>(dst));
            } else {
                for (int k = 0; k < 16 && x + k < nx; k++) dst[k] = (unsigned char)(v[k >> 2] >> (8 * (k & 3)));
            }
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned t = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        if (t) atomicAdd(&counts[frame], (unsigned long long)t);
    }
}
__global__ void __launch_bounds__(256) canny_expand_count(const unsigned long long *__restrict__ S, int wpr, unsigned char *__restrict__ edges,
                                                          int nx, int ny, unsigned long long *__restrict__ counts,
                                                          const unsigned long long *__restrict__ Wm, const unsigned *__restrict__ parents,
                                                          const unsigned *__restrict__ flags, unsigned last_sweep)
{
    const bool united = flags[HY_SWEEPS_MAX + blockIdx.y] == last_sweep;  // workgroup-uniform
    expand_count_frame(S, wpr, edges, nx, ny, counts, Wm, parents, united, blockIdx.y, blockIdx.x, gridDim.x);
}
// The three kernels above as ONE launch (round 5; a single 4K frame spent 10 us of its critical chain in the two union-find
// launches that find nothing to do).  A frame the last sweep left alone -- the case in practice -- is expanded at once.  A frame
// it did not finish takes the same three steps with two barriers between them, each across the workgroups of THAT FRAME only
// (a counter in the flags block, cleared by the gradient/NMS kernel).
// Forward progress, by construction (round 6; round 5 relied on workgroups being dispatched in grid order): a workgroup does not
// work on the (frame, slice) its grid position names but on the one its TICKET names -- tickets are handed out by an atomic counter
// as workgroups START RUNNING, frame by frame (ticket / blocks per frame).  The workgroups that are resident at any time therefore
// hold a gap-free prefix of the tickets: every frame but the last one they have reached is complete in tickets, so its barriers open
// with no further dispatch, its workgroups leave and make room; and the last frame completes as soon as the blocks of ONE frame are
// resident together -- which the host guarantees by launching no more blocks per frame than the stream's compute units hold at
// once (hipOccupancyMaxActiveBlocksPerMultiprocessor x the device's compute units; a stream confined to some of them by a CU mask
// takes the three launches instead: canny_finish_blocks).  Whatever else runs on the device only delays that.  Needs workgroups that can wait for each other (not the one-block-at-a-time emulator
// of the tests: the host asks the device).
__device__ __forceinline__ void frame_barrier(unsigned *counter, unsigned target)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(8);
    }
    __syncthreads();
    __threadfence();  // what the other workgroups wrote before they arrived
}
__global__ void __launch_bounds__(256) canny_finish(const unsigned long long *__restrict__ S, int wpr, unsigned char *__restrict__ edges, int nx, int ny,
                                                    unsigned long long *__restrict__ counts, const unsigned long long *__restrict__ Wm,
                                                    unsigned *__restrict__ parents, unsigned *__restrict__ flags, unsigned last_sweep, int n_frames,
                                                    unsigned *__restrict__ report, unsigned report_seq)
{
    __shared__ unsigned ticket_s;
    if (threadIdx.x == 0) ticket_s = __hip_atomic_fetch_add(flags + HY_SWEEPS_MAX + 2 * n_frames, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    // small batches: the first workgroup tells the HOST (pinned memory, no copy, nobody waits) how many sweeps this call's frames
    // needed -- the next calls on this context queue that many and one more (canny_device)
    if (report && ticket_s == 0) {
        unsigned most = 0;
        if ((int)threadIdx.x < n_frames) most = flags[HY_SWEEPS_MAX + threadIdx.x];
        if (threadIdx.x < 64) {
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) most = max(most, (unsigned)__shfl_xor((int)most, d));
            if (threadIdx.x == 0) {
                __hip_atomic_store(report + 1, most, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(report, report_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
    const int frame = (int)(ticket_s / gridDim.x), bid = (int)(ticket_s - (unsigned)frame * gridDim.x);  // grid: (blocks per frame, frames)
    const bool united = flags[HY_SWEEPS_MAX + frame] == last_sweep;  // workgroup-uniform
    if (united) {
        unsigned *bar = flags + HY_SWEEPS_MAX + n_frames + frame;
        uf_init_frame(S, Wm, wpr, nx, ny, parents, frame, bid, gridDim.x);
        __threadfence();
        frame_barrier(bar, gridDim.x);
        uf_merge_frame(S, Wm, wpr, nx, ny, parents, frame, bid, gridDim.x);
        __threadfence();
        frame_barrier(bar, 2u * gridDim.x);
    }
    expand_count_frame(S, wpr, edges, nx, ny, counts, Wm, parents, united, frame, bid, gridDim.x);
}
namespace {

// Wrapped, normalised 1-D kernel (tools.c:146-163) reduced to the taps >= 1e-17, in the oracle's order.  The oracle keeps
// taps down to 1e-22 (for s = 2: radius 14 instead of 12); the two extra pairs weigh 1.3e-19 and 1.5e-22 and contribute
// at most 510 * 1.3e-19 = 7e-17 to a value of order 1..255 -- below half a unit in the last place of the double
// accumulator except for near-black neighbourhoods, and thirteen orders below the float the result is rounded to.
#define CANNY_TAP_MIN 1e-17
// off / w: every kept tap; t: the same as a kernel argument when they are at most CANNY_MAX_TAPS (t->n = 0 otherwise: the
// caller takes the kernels that read the taps from memory -- tools.c:146-185 blurs with any s)
void make_taps(int n, double s, std::vector<int> &off, std::vector<double> &w, BlurTaps *t)
{
    std::vector<double> k(n);
    const double inv_s = 1 / s;
    double sum = 0;
    for (int i = 0; i < n; i++) {
        const double x = i < n / 2 ? i : i - n;
        k[i] = exp(-x * x * inv_s * inv_s);
        sum += k[i];
    }
    off.clear(); w.clear();
    for (int i = 0; i < n; i++) {
        const double v = k[i] / sum;
        if (v >= CANNY_TAP_MIN) { off.push_back(i); w.push_back(v); }
    }
    t->n = 0;
    if (off.size() <= CANNY_MAX_TAPS) {
        t->n = (int)off.size();
        for (int i = 0; i < t->n; i++) { t->off[i] = off[i]; t->w[i] = w[i]; }
    }
}

// radius of the taps make_taps keeps, when they form the symmetric set {0, 1..R, n-R..n-1} (n > 2R+1);
// -1 when the kernel wraps onto itself (tiny images) and only the generic kernels apply
int symmetric_radius(const BlurTaps &t, int n)
{
    if ((t.n & 1) == 0) return -1;
    const int R = t.n / 2;
    if (n < 2 * R + 2) return -1;
    for (int j = 0; j <= R; j++)
        if (t.off[j] != j) return -1;
    for (int j = 1; j <= R; j++)
        if (t.off[t.n - j] != n - j || t.w[t.n - j] != t.w[j]) return -1;
    return R;
}

template <int R>
imgfd_status launch_blur_march(imgfd_ctx *ctx, BlurMarchParams &p, int nf)
{
    using G = BlurGeom<R>;
    const int strips = ceil_div(p.nx, BM_TW);
    // Segment length: a workgroup walks (rows + 2R) rows in chunks of BM_CH.  With `slots` workgroups resident on the chip
    // the pass takes ceil(workgroups / slots) rounds of (chunks per segment) steps: the segment count that minimises that
    // product (ties: fewer, longer segments = less halo work).  (Round 2 aimed at 6 workgroups per CU whatever the chip
    // holds: a single 4K frame got 1260 workgroups of 4 chunks for 1024 slots -- two rounds, the second a quarter full:
    // 69 us instead of the 21 us per frame of a batch.)
    const long slots = (long)std::max(1, std::min(8, (int)((size_t)160 * 1024 / G::LDS_BYTES))) * ctx->num_cu;
    long best_cost = -1;
    int seg = p.ny;
    for (int nseg = 1; nseg <= ceil_div(p.ny, BM_CH); nseg++) {
        int m = ceil_div(ceil_div(p.ny, nseg) + 2 * R, BM_CH);
        if (m < 2) m = 2;
        const int sr = m * BM_CH - 2 * R;  // (rows + 2R) fills whole chunks
        if (sr < 1) continue;
        const long wgs = (long)strips * ceil_div(p.ny, sr) * nf;
        const long cost = ((wgs + slots - 1) / slots) * m;
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; seg = sr; }
    }
    p.seg_rows = seg;
    p.strips = strips;
    p.xcd_order = 1;
    dim3 grid((unsigned)strips * (unsigned)ceil_div(p.ny, seg), nf);
    IMGFD_HIP(ctx, hipFuncSetAttribute((const void *)canny_blur_march<R>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)G::LDS_BYTES));
    hipLaunchKernelGGL((canny_blur_march<R>), grid, dim3(BM_NT), G::LDS_BYTES, ctx->stream, p);
    IMGFD_HIP(ctx, hipGetLastError());
    return IMGFD_OK;
}

template <int HW, int BX, int BY>
imgfd_status launch_hyst_block(imgfd_ctx *ctx, unsigned long long *S, const unsigned long long *Wm, int wpr, int ny, int tiles_x, int tiles_y,
                               int nf, unsigned *flags, unsigned char *act, int sweeps)
{
    constexpr int SP = (HW + 2) | 1, WP = HW | 1;
    constexpr size_t lds = (size_t)BX * BY * 64 * (SP + WP) * sizeof(unsigned long long);
    if (lds > 48 * 1024)
        IMGFD_HIP(ctx, hipFuncSetAttribute((const void *)canny_hyst_block<HW, BX, BY>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    for (int i = 0; i < sweeps; i++) {
        // odd launches group the tiles half a block up and left
        const int sx = (i & 1) ? BX / 2 : 0, sy = (i & 1) ? BY / 2 : 0;
        const int blocks_x = ceil_div(tiles_x + sx, BX), blocks_y = ceil_div(tiles_y + sy, BY);
        hipLaunchKernelGGL((canny_hyst_block<HW, BX, BY>), dim3(blocks_x * blocks_y, nf), dim3(64 * BX * BY), lds, ctx->stream, S, Wm, wpr, ny, tiles_x,
                           tiles_y, blocks_x, sx, sy, flags, i, act, 1 /* wave priority 3: the sweeps run beside chip-filling kernels */);
    }
    IMGFD_HIP(ctx, hipGetLastError());
    return IMGFD_OK;
}
imgfd_status launch_hyst_blocks(imgfd_ctx *ctx, int hw, int shape, unsigned long long *S, const unsigned long long *Wm, int wpr, int ny, int tiles_x,
                                int tiles_y, int nf, unsigned *flags, unsigned char *act, int sweeps)
{
#define HY_CASE(W, X, Y) \
    if (hw == W && shape == 10 * X + Y) return launch_hyst_block<W, X, Y>(ctx, S, Wm, wpr, ny, tiles_x, tiles_y, nf, flags, act, sweeps);
    HY_CASE(1, 2, 4) HY_CASE(4, 2, 2)
#undef HY_CASE
    return imgfd_fail(ctx, IMGFD_ERR_INVALID, "canny: no sweep kernel for this tile width / block shape");
}

size_t canny_ws_bytes(int nx, int ny, int nf)
{
    const size_t n = (size_t)nx * ny * nf;
    const size_t words = (size_t)ceil_div(nx, 64) * ny * nf;
    return align_up(n * sizeof(double), 256) + align_up(n * sizeof(float), 256) + 2 * align_up(words * 8, 256) +
           align_up(12 * ((size_t)nx + ny), 256) + 512 +  // taps in memory (kernels of more than CANNY_MAX_TAPS taps)
           align_up(4 * ((size_t)HY_SWEEPS_MAX + 2 * (size_t)nf + 1), 256) +  // sweep flags, per-frame flags and counters, the finishing kernel's tickets
           align_up(2 * (size_t)nf * ceil_div(nx, 64) * ceil_div(ny, 64), 256) + 4096;
}

// Workgroups of canny_finish that can be resident together on this context's stream: the occupancy API's answer per compute unit x
// the device's units -- or 0 ("use the three launches") when the runtime does not say, or when the stream may NOT use every unit
// (hipExtStreamCreateWithCUMask): how a CU mask maps onto the XCDs a queue's workgroups are dealt to is not documented, so the
// units such a stream really has cannot be counted (a first cut multiplied by the mask's bits: 48 unfinished frames on a stream
// masked to 32 of 256 units never came back).  Computed once per context.
int canny_finish_blocks(imgfd_ctx *ctx)
{
    if (ctx->canny_finish_fit >= 0) return ctx->canny_finish_fit;
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)canny_finish, 256, 0) != hipSuccess || per_cu < 1) return ctx->canny_finish_fit = 0;
    uint32_t mask[32] = {0};
    const uint32_t words = (uint32_t)std::min(32, (ctx->num_cu + 31) / 32);
    if (ctx->stream && hipExtStreamGetCUMask(ctx->stream, words, mask) == hipSuccess) {
        int n = 0;
        for (int i = 0; i < ctx->num_cu; i++) n += (mask[i >> 5] >> (i & 31)) & 1u;
        if (n < ctx->num_cu) { (void)hipGetLastError(); return ctx->canny_finish_fit = 0; }
    }
    (void)hipGetLastError();
    return ctx->canny_finish_fit = per_cu * ctx->num_cu;
}

// all device work for nf frames; d_edges / d_counts are device buffers
imgfd_status canny_device(imgfd_ctx *ctx, const uint8_t *d_in, int row_stride, size_t frame_stride, int nx, int ny,
                          int nf, double s, double low_thr, double high_thr, int accGrad, uint8_t *d_edges,
                          int64_t *d_counts, const std::function<imgfd_status(int)> *hook = nullptr, bool first = true,
                          bool last = true)
{
    if (!(s > 0)) return imgfd_fail(ctx, IMGFD_ERR_INVALID, "canny: s must be positive");
    const size_t n = (size_t)nx * ny * nf;
    const int wpr = ceil_div(nx, 64);
    const size_t words = (size_t)wpr * ny * nf;
    // the taps of this (nx, ny, s): kept on the context between calls
    struct TapsCache {
        int nx = -1, ny = -1;
        double s = 0;
        BlurTaps tx, ty;
        std::vector<int> offx, offy;
        std::vector<double> wtx, wty;
    };
    if (!ctx->canny_taps) {
        ctx->canny_taps = new TapsCache();
        ctx->canny_taps_free = [](void *p) { delete static_cast<TapsCache *>(p); };
    }
    TapsCache &tc = *static_cast<TapsCache *>(ctx->canny_taps);
    if (tc.nx != nx || tc.ny != ny || tc.s != s) {
        tc.nx = -1;  // stays invalid if an allocation below throws
        make_taps(nx, s, tc.offx, tc.wtx, &tc.tx);
        make_taps(ny, s, tc.offy, tc.wty, &tc.ty);
        tc.nx = nx; tc.ny = ny; tc.s = s;
    }
    const BlurTaps &tx = tc.tx, &ty = tc.ty;
    const std::vector<int> &offx = tc.offx, &offy = tc.offy;
    const std::vector<double> &wtx = tc.wtx, &wty = tc.wty;
    const bool big = tx.n == 0 || ty.n == 0;  // more than CANNY_MAX_TAPS taps along an axis
    const int Rx = big ? -1 : symmetric_radius(tx, nx), Ry = big ? -1 : symmetric_radius(ty, ny);
    const int R = std::max(Rx, Ry);
    const bool fast = Rx >= 0 && Ry >= 0 && R <= 32;
    double *tmp = fast ? nullptr : (double *)ws_alloc(ctx, n * sizeof(double));
    float *blur = (float *)ws_alloc(ctx, n * sizeof(float));
    unsigned long long *S = (unsigned long long *)ws_alloc(ctx, words * 8);
    unsigned long long *Wm = (unsigned long long *)ws_alloc(ctx, words * 8);
    unsigned *flags = (unsigned *)ws_alloc(ctx, 4 * ((size_t)HY_SWEEPS_MAX + 2 * (size_t)nf + 1));
    const size_t act_bytes = 2 * (size_t)nf * wpr * ceil_div(ny, 64);  // tile activity of the sweeps, two parities (tiles of one word at the least)
    unsigned char *act = (unsigned char *)ws_alloc(ctx, act_bytes);
    const size_t tap_bytes = big ? align_up(12 * (offx.size() + offy.size()), 256) + 256 : 0;
    char *taps_dev = big ? (char *)ws_alloc(ctx, tap_bytes) : nullptr;
    if ((!fast && !tmp) || !blur || !S || !Wm || !flags || !act || (big && !taps_dev)) return imgfd_fail(ctx, IMGFD_ERR_OOM, "workspace reservation too small");
    // the caller's hook (imgfd_detect_dev releases FAST-9 and the Harris chain on its other stream through it) is called
    // with the position reached: 0 before the blur, 1 behind it, 2 behind the gradient/NMS kernel
    auto at = [&](int pos) -> imgfd_status {
        if (!hook || (pos < 2 ? !first : !last)) return IMGFD_OK;
        return (*hook)(pos);
    };
    IMGFD_TRY(at(0));
    if (fast) {
        BlurMarchParams p;
        memset(&p, 0, sizeof p);
        p.in = d_in; p.out = blur; p.nx = nx; p.ny = ny; p.row_stride = row_stride; p.frame_stride = (long)frame_stride;
        p.aligned4 = (nx % 4 == 0) && (row_stride % 4 == 0) && (frame_stride % 4 == 0) && ((size_t)d_in % 4 == 0);
        for (int j = 0; j <= Rx; j++) p.wx[j] = tx.w[j];
        for (int j = 0; j <= Ry; j++) p.wy[j] = ty.w[j];
        // zero taps up to the template radius add +0.0 to the sums: exact
        if (R <= 4) IMGFD_TRY(launch_blur_march<4>(ctx, p, nf));
        else if (R <= 7) IMGFD_TRY(launch_blur_march<7>(ctx, p, nf));
        else if (R <= 10) IMGFD_TRY(launch_blur_march<10>(ctx, p, nf));
        else if (R <= 12) IMGFD_TRY(launch_blur_march<12>(ctx, p, nf));
        else if (R <= 14) IMGFD_TRY(launch_blur_march<14>(ctx, p, nf));
        else if (R <= 18) IMGFD_TRY(launch_blur_march<18>(ctx, p, nf));
        else if (R <= 24) IMGFD_TRY(launch_blur_march<24>(ctx, p, nf));
        else IMGFD_TRY(launch_blur_march<32>(ctx, p, nf));
    } else if (big) {
        // doubles first (8-byte aligned), then the offsets; the host vectors die with this call: wait for the copies
        double *d_wx = (double *)taps_dev, *d_wy = d_wx + wtx.size();
        int *d_ox = (int *)(d_wy + wty.size()), *d_oy = d_ox + offx.size();
        IMGFD_HIP(ctx, hipMemcpyAsync(d_wx, wtx.data(), 8 * wtx.size(), hipMemcpyHostToDevice, ctx->stream));
        IMGFD_HIP(ctx, hipMemcpyAsync(d_wy, wty.data(), 8 * wty.size(), hipMemcpyHostToDevice, ctx->stream));
        IMGFD_HIP(ctx, hipMemcpyAsync(d_ox, offx.data(), 4 * offx.size(), hipMemcpyHostToDevice, ctx->stream));
        IMGFD_HIP(ctx, hipMemcpyAsync(d_oy, offy.data(), 4 * offy.size(), hipMemcpyHostToDevice, ctx->stream));
        IMGFD_HIP(ctx, hipStreamSynchronize(ctx->stream));
        dim3 g1(ceil_div(nx, 256), ny, nf);
        hipLaunchKernelGGL(canny_blur_rows_g, g1, dim3(256), 0, ctx->stream, d_in, row_stride, frame_stride, tmp, nx, ny, (const int *)d_ox,
                           (const double *)d_wx, (int)offx.size());
        hipLaunchKernelGGL(canny_blur_cols_g, g1, dim3(256), 0, ctx->stream, (const double *)tmp, blur, nx, ny, (const int *)d_oy,
                           (const double *)d_wy, (int)offy.size());
    } else {
        dim3 g1(ceil_div(nx, 256), ny, nf);
        hipLaunchKernelGGL(canny_blur_rows, g1, dim3(256), 0, ctx->stream, d_in, row_stride, frame_stride, tmp, nx, ny, tx);
        hipLaunchKernelGGL(canny_blur_cols, g1, dim3(256), 0, ctx->stream, tmp, blur, nx, ny, ty);
    }
    IMGFD_TRY(at(1));
    dim3 g2((unsigned)wpr * (unsigned)ceil_div(ny, GN_TY), nf);
    hipLaunchKernelGGL(canny_grad_nms, g2, dim3(256), 0, ctx->stream, blur, S, Wm, nx, ny, wpr, accGrad, (int)low_thr,
                       (int)high_thr, (int)(nx % 4 == 0 && (size_t)blur % 16 == 0), flags, HY_SWEEPS_MAX, wpr, 1,
                       (unsigned long long *)d_counts);
    IMGFD_HIP(ctx, hipGetLastError());
    IMGFD_TRY(at(2));
    unsigned uf_last_sweep = 0;
    unsigned *report = nullptr;
    unsigned report_seq = 0;
    // Hysteresis, terminated on the device -- no host read-back anywhere.  A fixed number of sweeps is queued (a sweep whose
    // predecessor changed nothing returns at once: an idle launch costs a few microseconds), then the union-find kernel, which
    // leaves at once for every frame the last sweep left alone and otherwise completes the frame whatever its chains look like.
    {
        // Tile width and tiles per workgroup by batch size (imgfd_canny_dev alone on 4K frames, us per frame, scripts/canny_shapes_probe.py,
        // profiles/r05/canny_block_sweeps.txt): up to a dozen frames one-word tiles in blocks of 2 x 4 (one frame 134, against 154 / 172
        // with two- / four-word tiles: a sweep's length is its slowest wave's fixpoint loop, and a pass over a tile costs its words);
        // batches four-word tiles in blocks of 2 x 2 (56.5 per frame at 32 frames, one-word tiles 59).
        const bool few = nf <= 12;
        const int hw = few ? 1 : 4, shape = few ? 24 : 22;  // (the other widths and block shapes measured: profiles/r05/canny_block_sweeps.txt)
        // The bench frames need 4 (one frame) to 6 launches (a batch) and one more that finds nothing; a launch that returns at
        // once costs ~1.5 us, a frame the queued launches do not finish 0.3-0.9 ms in the union-find part of canny_finish: the
        // margin is three launches.
        int sweeps = few ? 8 : 9;
        // Small batches (a single frame is ONE chain of dependent launches, and an idle sweep launch is 5-6 us of it: four of the
        // eight queued for a bench frame, round 5): as many as the recent calls on this context needed, and one more that finds
        // nothing.  The numbers come back through pinned memory (canny_finish's report), read here without waiting for anything:
        // a call may see the report of a call a few before it.  A frame that needs more than were queued is finished by
        // canny_finish's union-find part -- correct, 0.3-0.9 ms slower for that frame.
        report = nullptr;
        if (few && ctx->coop && canny_finish_blocks(ctx) >= 16 && nf <= 64) {
            if (!ctx->canny_report) {
                void *p = nullptr;
                if (hipHostMalloc(&p, sizeof(unsigned) * 2 * CANNY_REPORTS, hipHostMallocDefault) == hipSuccess) {
                    memset(p, 0, sizeof(unsigned) * 2 * CANNY_REPORTS);
                    ctx->canny_report = (unsigned *)p;
                }
                (void)hipGetLastError();
            }
            if (ctx->canny_report) {
                if (ctx->canny_report_nx != nx || ctx->canny_report_ny != ny) {  // another frame size: what its frames need is not known yet
                    ctx->canny_report_nx = nx; ctx->canny_report_ny = ny;
                    ctx->canny_report_base = ctx->canny_report_seq;
                }
                // the reports that have arrived (the host may be many calls ahead of the device: whatever the slots hold of calls on
                // this frame size counts): the largest number of working sweeps among them; a call whose LAST queued sweep still changed
                // something (its frames went to the union-find part) asks for all eight
                unsigned known = 0;
                bool any = false;
                for (int k = 0; k < CANNY_REPORTS; k++) {
                    const unsigned seq = __atomic_load_n(ctx->canny_report + 2 * k, __ATOMIC_ACQUIRE), val = __atomic_load_n(ctx->canny_report + 2 * k + 1, __ATOMIC_RELAXED);
                    if (seq > ctx->canny_report_base && seq <= ctx->canny_report_seq) {
                        any = true;
                        known = std::max(known, val >= ctx->canny_report_queued[seq % CANNY_REPORTS] && seq + CANNY_REPORTS > ctx->canny_report_seq ? 8u : val);
                    }
                }
                if (any) sweeps = std::max(2, std::min(8, (int)known + 1));
                report_seq = ++ctx->canny_report_seq;
                report = ctx->canny_report + 2 * (report_seq % CANNY_REPORTS);
            }
        }
        if (ctx->tune.hyst_sweeps >= 1 && ctx->tune.hyst_sweeps <= HY_SWEEPS_MAX) sweeps = ctx->tune.hyst_sweeps;  // tests: leave the work to the union-find part
        if (report) ctx->canny_report_queued[report_seq % CANNY_REPORTS] = (unsigned)sweeps;
        const int tiles_x = ceil_div(wpr, hw), tiles_y = ceil_div(ny, 64);
        IMGFD_TRY(launch_hyst_blocks(ctx, hw, shape, S, Wm, wpr, ny, tiles_x, tiles_y, nf, flags, act, sweeps));
        uf_last_sweep = (unsigned)sweeps;
        ctx->canny_flags = flags; ctx->canny_sweeps = sweeps; ctx->canny_frames = nf;
        IMGFD_HIP(ctx, hipGetLastError());
    }
    // the blur plane is dead behind the gradient/NMS kernel: 4 bytes per pixel for the union-find forest
    int exp_blocks = std::min(EXP_BLOCKS, ny);
    const int fit = canny_finish_blocks(ctx);  // blocks of canny_finish the stream's compute units hold at once (0: unknown)
    if (ctx->coop && fit >= 16) {  // (fewer: a stream confined to one or two compute units -- the three launches below)
        exp_blocks = std::min(exp_blocks, fit);  // a frame's blocks wait for each other: no more of them than can be resident together
        // union-find (frames the sweeps did not finish only), 0/255 bytes and pixels_nonzero in ONE launch
        hipLaunchKernelGGL(canny_finish, dim3(exp_blocks, nf), dim3(256), 0, ctx->stream, (const unsigned long long *)S, wpr, d_edges, nx, ny,
                           (unsigned long long *)d_counts, (const unsigned long long *)Wm, reinterpret_cast<unsigned *>(blur), flags, uf_last_sweep, nf, report, report_seq);
    } else {
        // Workgroups per frame of the union-find kernels: enough to spread a frame that needs them over the chip, few enough
        // that the idle case stays one short launch each
        const int uf_blocks = std::max(1, std::min({2048 / nf, ceil_div(wpr * ny, 2 * UF_NT), 256}));
        hipLaunchKernelGGL(canny_uf_init, dim3(uf_blocks, nf), dim3(UF_NT), 0, ctx->stream, (const unsigned long long *)S, (const unsigned long long *)Wm, wpr, nx, ny,
                           reinterpret_cast<unsigned *>(blur), (const unsigned *)flags, uf_last_sweep);
        hipLaunchKernelGGL(canny_uf_merge, dim3(uf_blocks, nf), dim3(UF_NT), 0, ctx->stream, (const unsigned long long *)S, (const unsigned long long *)Wm, wpr, nx, ny,
                           reinterpret_cast<unsigned *>(blur), (const unsigned *)flags, uf_last_sweep);
        // 0/255 bytes and pixels_nonzero in one kernel (the gradient/NMS kernel zeroed the counts)
        hipLaunchKernelGGL(canny_expand_count, dim3(exp_blocks, nf), dim3(256), 0, ctx->stream, S, wpr, d_edges, nx, ny,
                           (unsigned long long *)d_counts, (const unsigned long long *)Wm, (const unsigned *)blur, (const unsigned *)flags, uf_last_sweep);
    }
    IMGFD_HIP(ctx, hipGetLastError());
    return IMGFD_OK;
}

}  // namespace

extern "C" {

// edges: nx*ny bytes (0/255), or -- edges_f64 -- nx*ny doubles (0.0/255.0: the NumericMatrix of rcpp_canny.cpp:226-233)
static imgfd_status canny_host(imgfd_ctx *ctx, const void *img, int kind, int nx, int ny, double s, double low_thr,
                               double high_thr, int accGrad, uint8_t *edges, double *edges_f64, int64_t *pixels_nonzero)
{
    if (!ctx) return IMGFD_ERR_INVALID;
    if (!img || (!edges && !edges_f64) || !pixels_nonzero || nx < 1 || ny < 1 || !frame_fits(nx, ny))
        return imgfd_fail(ctx, IMGFD_ERR_INVALID, "imgfd_canny: bad argument");
    IMGFD_HIP(ctx, hipSetDevice(ctx->device));
    const size_t n = (size_t)nx * ny;
    IMGFD_TRY(ws_reserve(ctx, canny_ws_bytes(nx, ny, 1) + 2 * align_up(n, 256) + upload_stage_bytes(kind, n) + (edges_f64 ? align_up(n * sizeof(double), 256) : 0) + 512));
    uint8_t *d_in = (uint8_t *)ws_alloc(ctx, n);
    uint8_t *d_edges = (uint8_t *)ws_alloc(ctx, n);
    int64_t *d_count = (int64_t *)ws_alloc(ctx, sizeof(int64_t));
    double *d_wide = edges_f64 ? (double *)ws_alloc(ctx, n * sizeof(double)) : nullptr;
    if (!d_in || !d_edges || !d_count || (edges_f64 && !d_wide)) return imgfd_fail(ctx, IMGFD_ERR_OOM, "workspace reservation too small");
    IMGFD_TRY(upload_image(ctx, img, kind, n, d_in));
    IMGFD_TRY(canny_device(ctx, d_in, nx, n, nx, ny, 1, s, low_thr, high_thr, accGrad, d_edges, d_count));
    if (edges_f64) {
        IMGFD_TRY(download_widened(ctx, d_edges, true, n, d_wide, edges_f64));
    } else
    IMGFD_HIP(ctx, hipMemcpyAsync(edges, d_edges, n, hipMemcpyDeviceToHost, ctx->stream));
    IMGFD_HIP(ctx, hipMemcpyAsync(pixels_nonzero, d_count, sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
    IMGFD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return IMGFD_OK;
}

imgfd_status imgfd_canny(imgfd_ctx *ctx, const uint8_t *img, int nx, int ny, double s, double low_thr,
                         double high_thr, int accGrad, uint8_t *edges, int64_t *pixels_nonzero)
{
    return imgfd_guard(ctx, [&] { return canny_host(ctx, img, IMGFD_SRC_U8, nx, ny, s, low_thr, high_thr, accGrad, edges, nullptr, pixels_nonzero); });
}

imgfd_status imgfd_canny_i32(imgfd_ctx *ctx, const int32_t *image, int nx, int ny, double s, double low_thr,
                             double high_thr, int accGrad, uint8_t *edges, int64_t *pixels_nonzero)
{
    return imgfd_guard(ctx, [&] { return canny_host(ctx, image, IMGFD_SRC_I32, nx, ny, s, low_thr, high_thr, accGrad, edges, nullptr, pixels_nonzero); });
}

imgfd_status imgfd_canny_f64out(imgfd_ctx *ctx, const int32_t *image, int nx, int ny, double s, double low_thr,
                                double high_thr, int accGrad, double *edges, int64_t *pixels_nonzero)
{
    return imgfd_guard(ctx, [&] { return canny_host(ctx, image, IMGFD_SRC_I32, nx, ny, s, low_thr, high_thr, accGrad, nullptr, edges, pixels_nonzero); });
}

imgfd_status imgfd_canny_dev(imgfd_ctx *ctx, const imgfd_frames *fr, double s, double low_thr,
                             double high_thr, int accGrad, uint8_t *d_edges, int64_t *d_counts)
{
    return canny_dev_hooked(ctx, fr, s, low_thr, high_thr, accGrad, d_edges, d_counts, nullptr);
}

}  // extern "C"

imgfd_status canny_dev_hooked(imgfd_ctx *ctx, const imgfd_frames *fr, double s, double low_thr, double high_thr, int accGrad,
                              uint8_t *d_edges, int64_t *d_counts, const std::function<imgfd_status(int)> *hook)
try {
    if (!ctx || !fr || !fr->d_frames || !d_edges || !d_counts || fr->n_frames < 0 || fr->dtype != 0 || fr->nx < 1 || fr->ny < 1 ||
        !frame_fits(fr->nx, fr->ny))
        return imgfd_fail(ctx, IMGFD_ERR_INVALID, "imgfd_canny_dev: bad argument (frames must be u8)");
    if (!fr->n_frames) {
        for (int pos = 0; hook && pos < 3; pos++) IMGFD_TRY((*hook)(pos));
        return IMGFD_OK;
    }
    IMGFD_HIP(ctx, hipSetDevice(ctx->device));
    const int nx = fr->nx, ny = fr->ny;
    const size_t per_frame = canny_ws_bytes(nx, ny, 1);
    const int chunk = sub_batch_frames(ctx, fr->n_frames, per_frame, (size_t)12 << 30);
    IMGFD_TRY(ws_reserve(ctx, canny_ws_bytes(nx, ny, chunk) + 512));
    for (int f0 = 0; f0 < fr->n_frames; f0 += chunk) {
        const int nf = std::min(chunk, fr->n_frames - f0);
        ctx->ws_used = 0;
        IMGFD_TRY(canny_device(ctx, (const uint8_t *)fr->d_frames + (size_t)f0 * fr->frame_stride_bytes,
                               fr->row_stride_bytes, fr->frame_stride_bytes, nx, ny, nf, s, low_thr, high_thr, accGrad,
                               d_edges + (size_t)f0 * nx * ny, d_counts + f0, hook, f0 == 0, f0 + chunk >= fr->n_frames));
    }
    return IMGFD_OK;
} catch (const std::bad_alloc &) {
    return imgfd_fail(ctx, IMGFD_ERR_OOM, "canny_dev_hooked: out of host memory");
} catch (...) {
    return imgfd_fail(ctx, IMGFD_ERR_HIP, "canny_dev_hooked: unexpected C++ exception");
}
