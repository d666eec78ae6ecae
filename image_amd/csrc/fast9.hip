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
// order.  One thread per pixel, one wave per 64 pixels of a row (the __ballot word is the mask word).
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

template <int NONMAX>
__global__ void __launch_bounds__(256) fast9_detect(const unsigned char *__restrict__ img, int w, int h, int stride,
                                                    size_t frame_stride, int b,
                                                    unsigned char *__restrict__ score,
                                                    unsigned long long *__restrict__ mask,
                                                    unsigned *__restrict__ rowcount, int words_per_row)
{
    const int lane = threadIdx.x & 63;
    const int x = blockIdx.x * 64 + lane;
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int frame = blockIdx.z;
    bool corner = false;
    int sc = 0;
    if (x >= 3 && x < w - 3 && y >= 3 && y < h - 3) {
        const unsigned char *c = img + (size_t)frame * frame_stride + (size_t)y * stride + x;
        const int p = *c;
        const int cb = min(255, p + b), c_b = max(0, p - b);
        int v[16];
#define F9_LOAD(i, dx, dy) v[i] = c[(dx) + stride * (dy)];
        F9_RING(F9_LOAD)
#undef F9_LOAD
        unsigned brighter = 0, darker = 0;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            brighter |= (unsigned)(v[i] > cb) << i;
            darker |= (unsigned)(v[i] < c_b) << i;
        }
        corner = f9_has_arc9(brighter) || f9_has_arc9(darker);
        if (NONMAX && corner) sc = f9_score(v, p) + 1;  // 1..255; 0 = not a corner
    }
    if (NONMAX) {
        if (x < w && y < h) score[((size_t)frame * h + y) * w + x] = (unsigned char)sc;
    } else {
        const unsigned long long word = __ballot(corner);
        if (lane == 0 && y < h && (int)blockIdx.x < words_per_row) {
            mask[((size_t)frame * h + y) * words_per_row + blockIdx.x] = word;
            if (word) atomicAdd(&rowcount[(size_t)frame * h + y], (unsigned)__popcll(word));
        }
    }
}

__global__ void __launch_bounds__(256) fast9_nms(const unsigned char *__restrict__ score, int w, int h,
                                                 unsigned long long *__restrict__ mask,
                                                 unsigned *__restrict__ rowcount, int words_per_row)
{
    const int lane = threadIdx.x & 63;
    const int x = blockIdx.x * 64 + lane;
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int frame = blockIdx.z;
    bool keep = false;
    // corners only exist for 3 <= x < w-3, 3 <= y < h-3, so the 3x3 neighbourhood is in range
    if (x >= 3 && x < w - 3 && y >= 3 && y < h - 3) {
        const unsigned char *c = score + ((size_t)frame * h + y) * w + x;
        const int s = *c;
        if (s) {
            keep = c[-w - 1] < s && c[-w] < s && c[-w + 1] < s && c[-1] < s && c[1] < s &&
                   c[w - 1] < s && c[w] < s && c[w + 1] < s;
        }
    }
    const unsigned long long word = __ballot(keep);
    if (lane == 0 && y < h && (int)blockIdx.x < words_per_row) {
        mask[((size_t)frame * h + y) * words_per_row + blockIdx.x] = word;
        if (word) atomicAdd(&rowcount[(size_t)frame * h + y], (unsigned)__popcll(word));
    }
}

// d_score: n_frames*w*h bytes of scratch, needed when nonmax != 0
imgfd_status launch_fast9(imgfd_ctx *ctx, const uint8_t *d_img, int w, int h, int stride,
                          size_t frame_stride, int n_frames, int threshold, int nonmax,
                          uint8_t *d_score, const CompactBuffers &cb)
{
    dim3 grid(cb.words_per_row, ceil_div(h, 4), n_frames);
    if (!nonmax) {
        hipLaunchKernelGGL(fast9_detect<0>, grid, dim3(256), 0, ctx->stream, d_img, w, h, stride, frame_stride,
                           threshold, (unsigned char *)nullptr, cb.mask, cb.rowcount, cb.words_per_row);
    } else {
        if (!d_score) return imgfd_fail(ctx, IMGFD_ERR_INVALID, "fast9 non-max suppression needs a score plane");
        hipLaunchKernelGGL(fast9_detect<1>, grid, dim3(256), 0, ctx->stream, d_img, w, h, stride, frame_stride,
                           threshold, d_score, cb.mask, cb.rowcount, cb.words_per_row);
        hipLaunchKernelGGL(fast9_nms, grid, dim3(256), 0, ctx->stream, d_score, w, h, cb.mask, cb.rowcount,
                           cb.words_per_row);
    }
    IMGFD_HIP(ctx, hipGetLastError());
    return IMGFD_OK;
}
