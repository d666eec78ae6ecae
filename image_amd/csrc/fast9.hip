// image_amd/csrc/fast9.hip -- FAST-9 corner detector (K7, K8).
//
// Replaces image.CornerDetectionF9/src/f9.cpp: detectAllCorners (:2953-5714, a generated decision
// tree), cornerScore (:171-2941, the same tree inside a binary search) and nonMaxSuppression
// (:84-169).  What the tree decides: pixel p (3 <= x < w-3, 3 <= y < h-3) is a corner iff 9 contiguous
// pixels of the 16-pixel radius-3 ring (makeOffsets, :42-59) are all > cb = min(255, p+b) or all
// < c_b = max(0, p-b) (:2962-2963, saturating in unsigned char).  The binary search of cornerScore
// returns the largest b in [threshold, 254] for which p is still a corner, which has the closed form
//     score = max( max_arcs(min_9 ring) - p - 1 ,  p - min_arcs(max_9 ring) - 1 ).
// nonMaxSuppression keeps a corner iff none of its 8 neighbours is a corner with score >= its own.
// All integer arithmetic: results are bit-exact, and compact.hip emits them in the reference's raster
// order.  One workgroup per 64x16 tile staged in LDS; a wave covers 64 pixels of a row (the __ballot word is the
// mask word).
#include "common.h"

// ring offsets in the order of makeOffsets(): (dx,dy)
#define F9_RING(X)                                                                                  \
    X(0, 0, 3) X(1, 1, 3) X(2, 2, 2) X(3, 3, 1) X(4, 3, 0) X(5, 3, -1) X(6, 2, -2) X(7, 1, -3)       \
    X(8, 0, -3) X(9, -1, -3) X(10, -2, -2) X(11, -3, -1) X(12, -3, 0) X(13, -3, 1) X(14, -2, 2) X(15, -1, 3)

__device__ __forceinline__ bool f9_has_arc9(unsigned m16)
{
    const unsigned M = m16 | (m16 << 16);
    unsigned t = M & (M >> 1);  // runs of 2
    t &= t >> 2;                // runs of 4
    t &= t >> 4;                // runs of 8
    t &= M >> 8;                // runs of 9
    return (t & 0xFFFFu) != 0;
}

__device__ __forceinline__ int f9_score(const int (&v)[16], int p)
{
    // sliding min/max over every 9-arc of the circular ring
    int mn2[16], mx2[16], mn4[16], mx4[16], mn8[16], mx8[16];
#pragma unroll
    for (int i = 0; i < 16; i++) { mn2[i] = min(v[i], v[(i + 1) & 15]); mx2[i] = max(v[i], v[(i + 1) & 15]); }
#pragma unroll
    for (int i = 0; i < 16; i++) { mn4[i] = min(mn2[i], mn2[(i + 2) & 15]); mx4[i] = max(mx2[i], mx2[(i + 2) & 15]); }
#pragma unroll
    for (int i = 0; i < 16; i++) { mn8[i] = min(mn4[i], mn4[(i + 4) & 15]); mx8[i] = max(mx4[i], mx4[(i + 4) & 15]); }
    int best_min = 0, best_max = 255;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        best_min = max(best_min, min(mn8[i], v[(i + 8) & 15]));
        best_max = min(best_max, max(mx8[i], v[(i + 8) & 15]));
    }
    return max(best_min - p - 1, p - best_max - 1);
}

#define F9_TX 64   // tile width = one __ballot word
#define F9_TY 16
#define F9_HALO 4  // ring radius 3 + 1 (the 3x3 non-max neighbourhood needs the scores of the neighbours)

// One workgroup per 64x16 tile.  The u8 tile (+4 halo) is staged in LDS once (HBM traffic = the algorithmic
// 1 B/px + halo); the ring test reads LDS; with non-max suppression the scores of the tile (+1 halo) go to a
// second LDS tile and the 3x3 test runs on it, so neither a score plane nor a second kernel touch HBM.
// Wave w handles tile rows w, w+4, ...: lane <-> column, so one __ballot is the 64-bit mask word.
template <int NONMAX>
__global__ void __launch_bounds__(256) fast9_tile(const unsigned char *__restrict__ img, int w, int h, int stride,
                                                  size_t frame_stride, int b, int aligned4,
                                                  unsigned long long *__restrict__ mask,
                                                  unsigned *__restrict__ rowcount, int words_per_row)
{
    constexpr int LW = F9_TX + 2 * F9_HALO, LH = F9_TY + 2 * F9_HALO;  // 72 x 24
    __shared__ unsigned tile32[LH][LW / 4];
    __shared__ unsigned char sc[F9_TY + 2][F9_TX + 2 + 2];
    unsigned char(*tile)[LW] = reinterpret_cast<unsigned char(*)[LW]>(tile32);
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * F9_TX, y0 = blockIdx.y * F9_TY;
    const unsigned char *fr = img + (size_t)blockIdx.z * frame_stride;
    // ---- stage the tile; out-of-image positions repeat the border pixel (they are never tested, only loaded)
    for (int i = tid; i < LH * (LW / 4); i += 256) {
        const int r = i / (LW / 4), q = i - r * (LW / 4);
        const int gy = min(max(y0 - F9_HALO + r, 0), h - 1);
        const int gx = x0 - F9_HALO + 4 * q;
        const unsigned char *row = fr + (size_t)gy * stride;
        unsigned v;
        if (aligned4 && gx >= 0 && gx + 3 < w) {
            v = *reinterpret_cast<const unsigned *>(row + gx);
        } else {
            v = 0;
#pragma unroll
            for (int e = 0; e < 4; e++) v |= (unsigned)row[min(max(gx + e, 0), w - 1)] << (8 * e);
        }
        tile32[r][q] = v;
    }
    __syncthreads();
    // corner test (+ score) of the pixel at tile position (ty, tx), image position (gx, gy)
    auto corner_score = [&](int ty, int tx, int gx, int gy) -> int {
        if (gx < 3 || gx >= w - 3 || gy < 3 || gy >= h - 3) return 0;
        const unsigned char *c = &tile[ty][tx];
        const int p = *c;
        const int cb = min(255, p + b), c_b = max(0, p - b);
        // any 9 contiguous ring pixels contain at least two of the four compass points
        const int n0 = c[3 * LW], n4 = c[3], n8 = c[-3 * LW], n12 = c[-3];
        const int nb = (n0 > cb) + (n4 > cb) + (n8 > cb) + (n12 > cb);
        const int nd = (n0 < c_b) + (n4 < c_b) + (n8 < c_b) + (n12 < c_b);
        if (nb < 2 && nd < 2) return 0;
        int v[16];
#define F9_LOAD(i, dx, dy) v[i] = c[(dx) + LW * (dy)];
        F9_RING(F9_LOAD)
#undef F9_LOAD
        unsigned brighter = 0, darker = 0;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            brighter |= (unsigned)(v[i] > cb) << i;
            darker |= (unsigned)(v[i] < c_b) << i;
        }
        if (!(f9_has_arc9(brighter) || f9_has_arc9(darker))) return 0;
        return NONMAX ? f9_score(v, p) + 1 : 1;  // score + 1 in 1..255; 0 = not a corner
    };
    const int lane = tid & 63, wv = tid >> 6;
    if (NONMAX) {
        for (int i = tid; i < (F9_TY + 2) * (F9_TX + 2); i += 256) {
            const int r = i / (F9_TX + 2), cx = i - r * (F9_TX + 2);
            sc[r][cx] = (unsigned char)corner_score(r + F9_HALO - 1, cx + F9_HALO - 1, x0 + cx - 1, y0 + r - 1);
        }
        __syncthreads();
    }
    for (int r = wv; r < F9_TY; r += 4) {
        const int gy = y0 + r;
        bool keep;
        if (NONMAX) {
            const int s = sc[r + 1][lane + 1];
            keep = s && sc[r][lane] < s && sc[r][lane + 1] < s && sc[r][lane + 2] < s && sc[r + 1][lane] < s &&
                   sc[r + 1][lane + 2] < s && sc[r + 2][lane] < s && sc[r + 2][lane + 1] < s && sc[r + 2][lane + 2] < s;
        } else {
            keep = corner_score(r + F9_HALO, lane + F9_HALO, x0 + lane, gy) != 0;
        }
        const unsigned long long word = __ballot(keep);
        if (lane == 0 && gy < h) {
            mask[((size_t)blockIdx.z * h + gy) * words_per_row + blockIdx.x] = word;
            if (word) atomicAdd(&rowcount[(size_t)blockIdx.z * h + gy], (unsigned)__popcll(word));
        }
    }
}

imgfd_status launch_fast9(imgfd_ctx *ctx, const uint8_t *d_img, int w, int h, int stride,
                          size_t frame_stride, int n_frames, int threshold, int nonmax, const CompactBuffers &cb)
{
    dim3 grid(cb.words_per_row, ceil_div(h, F9_TY), n_frames);
    const int aligned4 = ((size_t)d_img % 4 == 0) && stride % 4 == 0 && frame_stride % 4 == 0;
    if (!nonmax)
        hipLaunchKernelGGL(fast9_tile<0>, grid, dim3(256), 0, ctx->stream, d_img, w, h, stride, frame_stride, threshold,
                           aligned4, cb.mask, cb.rowcount, cb.words_per_row);
    else
        hipLaunchKernelGGL(fast9_tile<1>, grid, dim3(256), 0, ctx->stream, d_img, w, h, stride, frame_stride, threshold,
                           aligned4, cb.mask, cb.rowcount, cb.words_per_row);
    IMGFD_HIP(ctx, hipGetLastError());
    return IMGFD_OK;
}
