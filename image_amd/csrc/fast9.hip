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
// order.  One workgroup per 64x64 tile staged in LDS; a tile row is 64 pixels wide = one 64-bit word of the
// compaction mask.
#include "common.h"

#include <algorithm>

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

// packed arithmetic on two unsigned 16-bit lanes of a dword (v_pk_sub_u16 clamp / v_pk_max_u16 / v_pk_min_u16)
__device__ __forceinline__ unsigned f9_pk_subs(unsigned a, unsigned b)
{
    return __builtin_bit_cast(unsigned, __builtin_elementwise_sub_sat(__builtin_bit_cast(imgfd_u16x2, a), __builtin_bit_cast(imgfd_u16x2, b)));
}
__device__ __forceinline__ unsigned f9_pk_max(unsigned a, unsigned b)
{
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(imgfd_u16x2, a), __builtin_bit_cast(imgfd_u16x2, b)));
}
__device__ __forceinline__ unsigned f9_pk_min(unsigned a, unsigned b)
{
    return __builtin_bit_cast(unsigned, __builtin_elementwise_min(__builtin_bit_cast(imgfd_u16x2, a), __builtin_bit_cast(imgfd_u16x2, b)));
}
// bytes 1 and 3 of a dword in the low bytes of its two 16-bit lanes (v_perm_b32: one instruction for shift + mask)
__device__ __forceinline__ unsigned f9_odd_bytes(unsigned v) { return __builtin_amdgcn_perm(0u, v, 0x0c030c01u); }

// Compass pre-test, two pixels at a time (16-bit lanes).  A contiguous arc of 9 of the 16 ring pixels leaves out 7
// consecutive ones, which cannot hold two opposite ring pixels (8 apart): the arc contains the north or the south pixel,
// and the east or the west pixel.  So a corner needs max(N,S) and max(E,W) above p + b, or min(N,S) and min(E,W) below
// p - b (f9.cpp:2962-2963: cb = min(255, p + b), c_b = max(0, p - b); a ring pixel is never above 255 or below 0, so the
// unsaturated sum and the saturating difference decide the same).  Non-zero lane <=> the pixel passes.  11 packed
// instructions per pair of pixels (the "two of the four compass pixels" form of this test took 23 and let more through).
__device__ __forceinline__ unsigned f9_compass(unsigned p, unsigned n, unsigned e, unsigned s, unsigned w, unsigned bb)
{
    const unsigned hi = f9_pk_min(f9_pk_max(n, s), f9_pk_max(e, w));
    const unsigned lo = f9_pk_max(f9_pk_min(n, s), f9_pk_min(e, w));
    return f9_pk_subs(hi, p + bb) | f9_pk_subs(f9_pk_subs(p, bb), lo);
}

#define F9_TX 64   // tile width = one __ballot word
#ifndef F9_TY
#define F9_TY 64  // sweep on MI355X, 4K frames: 16 -> 780 us, 32 -> 631, 64 -> 555, 96 -> 583, 128 -> 660 (per 32 frames)
#endif
#define F9_HALO 4  // ring radius 3 + 1 (the 3x3 non-max neighbourhood needs the scores of the neighbours)

// One workgroup per 64 x F9_TY tile.  The u8 tile (+4 halo) is staged in LDS once (HBM traffic = the algorithmic
// 1 B/px + halo).  Three phases:
//   1. compass pre-test on packed dwords: a thread takes 4 consecutive pixels (5 LDS dword reads instead of 20 byte
//      reads) and evaluates them two at a time in the 16-bit lanes of packed instructions (f9_compass: a 9-arc
//      holds one pixel of each opposite compass pair).  Survivors (a few percent of a natural frame) are appended to a
//      candidate list in LDS.
//   2. the full ring test (+ score) runs over the candidate list with all lanes busy and writes the score tile
//      (+1 halo for the 3x3 test) in LDS.
//   3. the 3x3 non-max test reads the score tile, again for the candidates only; survivors set their bit in the
//      row's 64-bit mask word (tile width = word width).  Neither a score plane nor a second kernel touch HBM.
template <int NONMAX>
__global__ void __launch_bounds__(256) fast9_tile(const unsigned char *__restrict__ img, int w, int h, int stride,
                                                  size_t frame_stride, int b, int aligned4, int aligned16,
                                                  unsigned long long *__restrict__ mask,
                                                  unsigned *__restrict__ rowcount, int words_per_row, TileRuns runs, int xcd_order)
{
    // LDS tile: rows y0-4 .. y0+TY+3; columns x0-16 .. x0+79 (the left margin of 16 keeps the first column 16-byte
    // aligned in the frame, so interior tiles are staged with 16-byte loads; only x0-4 .. x0+67 are ever looked at)
    constexpr int XL = 16, LW = F9_TX + 2 * XL, LH = F9_TY + 2 * F9_HALO;
    constexpr int LQ = LW / 4, QL = (XL - F9_HALO) / 4, NQ = (F9_TX + 2 * F9_HALO) / 4;  // dwords per row; the 18 in use
    constexpr int SR = F9_TY + 2, SCW = F9_TX + 4;                     // score tile: 18 rows of 68 bytes (66 used)
    __shared__ __attribute__((aligned(16))) unsigned tile32[LH][LQ];
    __shared__ unsigned sc32[SR][SCW / 4];
    __shared__ unsigned short cand[SR * (F9_TX + 2)];
    __shared__ unsigned ncand;
    __shared__ unsigned long long rowmask[F9_TY];
    unsigned char(*tile)[LW] = reinterpret_cast<unsigned char(*)[LW]>(tile32);
    unsigned char(*sc)[SCW] = reinterpret_cast<unsigned char(*)[SCW]>(sc32);
    const int tid = threadIdx.x;
    // the score tile is cleared once; every tile puts back to 0 the cells its candidates wrote (far fewer than the tile)
    for (int i = tid; i < SR * (SCW / 4); i += 256) (&sc32[0][0])[i] = 0u;
    {  // one run of tiles per workgroup (TileRuns, common.h)
    const unsigned run_id = xcd_order ? imgfd_xcd_tile(blockIdx.x, gridDim.x) : blockIdx.x;  // an XCD owns neighbouring runs and bands: their shared lines are L2 hits
    const int frame = (int)(run_id / (unsigned)(runs.runs_per_band * runs.bands));
    const int in_frame = (int)(run_id - (unsigned)frame * (unsigned)(runs.runs_per_band * runs.bands));
    const int band = in_frame / runs.runs_per_band, tile0 = (in_frame - band * runs.runs_per_band) * runs.run;
    for (int tile_x = tile0; tile_x < min(tile0 + runs.run, runs.tiles_x); tile_x++) {
    const int x0 = tile_x * F9_TX, y0 = band * F9_TY;
    const unsigned char *fr = img + (size_t)frame * frame_stride;
    // ---- stage the tile; out-of-image positions repeat the border pixel (they are never tested, only loaded)
    if (aligned16 && x0 - XL >= 0 && x0 - XL + LW <= w) {  // workgroup-uniform
        for (int i = tid; i < LH * (LW / 16); i += 256) {
            const int r = i / (LW / 16), q = i - r * (LW / 16);
            const int gy = min(max(y0 - F9_HALO + r, 0), h - 1);
            *reinterpret_cast<uint4 *>(&tile32[r][4 * q]) = *reinterpret_cast<const uint4 *>(fr + (size_t)gy * stride + (x0 - XL + 16 * q));
        }
    } else {
        for (int i = tid; i < LH * NQ; i += 256) {
            const int r = i / NQ, q = QL + i - r * NQ;
            const int gy = min(max(y0 - F9_HALO + r, 0), h - 1);
            const int gx = x0 - XL + 4 * q;
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
    }
    for (int i = tid; i < F9_TY; i += 256) rowmask[i] = 0ull;
    if (tid == 0) ncand = 0u;
    __syncthreads();
    // ---- phase 1: compass pre-test; score-tile row r <-> tile row r + 3, score column cx <-> tile column cx + 3.
    // A thread keeps its dword column q and walks down the rows, P1_ROWS at a time: what depends on the column alone
    // (the mask of tested pixels) is worked out once per tile, and no item index has to be divided into row and column.
    constexpr int P1_ROWS = 256 / NQ;
    const int p1_row = tid / NQ, q = QL + tid - p1_row * NQ;
    unsigned colmask;  // pixels of this dword inside the tested range: tile columns [lo, hi) and image columns [3, w - 3)
    {
        const int lo = max(NONMAX ? XL - 1 : XL, 3 - (x0 - XL)), hi = min(NONMAX ? XL + F9_TX + 1 : XL + F9_TX, w - 3 - (x0 - XL));
        const int a = lo - 4 * q, e_end = hi - 4 * q;  // valid e in [a, e_end)
        const unsigned vlo = a <= 0 ? 0xfu : (a >= 4 ? 0u : (0xfu << a) & 0xfu);
        const unsigned vhi = e_end >= 4 ? 0xfu : (e_end <= 0 ? 0u : (1u << e_end) - 1u);
        colmask = p1_row < P1_ROWS ? vlo & vhi : 0u;
    }
    // rows with a tested pixel: image rows [3, h - 3); without non-max suppression only the tile's own rows
    const int r_first = max(NONMAX ? 0 : 1, 3 - (y0 - 1)), r_last = min(NONMAX ? SR - 1 : F9_TY, h - 4 - (y0 - 1));
    for (int r = r_first + p1_row; r <= r_last && colmask; r += P1_ROWS) {
        const int ty = r + F9_HALO - 1;
        const unsigned cur = tile32[ty][q], prev = tile32[ty][q - 1], next = tile32[ty][q + 1];
        const unsigned up = tile32[ty - 3][q], dn = tile32[ty + 3][q];
        const unsigned lf = (prev >> 8) | (cur << 24);  // byte e = pixel (4q + e) - 3
        const unsigned rt = (cur >> 24) | (next << 8);  // byte e = pixel (4q + e) + 3
        // pixels 0, 2 of the dword in the 16-bit lanes of the "even" words, pixels 1, 3 in the "odd" ones
        const unsigned M = 0x00ff00ffu, bb = (unsigned)b * 0x00010001u;
        const unsigned pe = cur & M, po = f9_odd_bytes(cur);
        const unsigned fe = f9_compass(pe, up & M, rt & M, dn & M, lf & M, bb);
        const unsigned fo = f9_compass(po, f9_odd_bytes(up), f9_odd_bytes(rt), f9_odd_bytes(dn), f9_odd_bytes(lf), bb);
        // append the survivors (order inside the list is irrelevant: every candidate writes its own score cell); only
        // a few percent of the threads get here, so a returning LDS atomic is cheaper than a wave-wide prefix sum -- and
        // the four lane tests are only made for a dword that has a survivor at all
        if (fe | fo) {
            const unsigned pass = (((fe & 0xffffu) ? 1u : 0u) | ((fo & 0xffffu) ? 2u : 0u) | ((fe >> 16) ? 4u : 0u) | ((fo >> 16) ? 8u : 0u)) & colmask;
            unsigned at = atomicAdd(&ncand, (unsigned)__popc(pass));
#pragma unroll
            for (int e = 0; e < 4; e++)
                if (pass & (1u << e)) cand[at++] = (unsigned short)(r * LW + 4 * q + e);
        }
    }
    __syncthreads();
    // ---- phase 2: ring test and score of the candidates
    const unsigned nc = ncand;
    for (unsigned i = tid; i < nc; i += 256) {
        const int pos = cand[i];
        const int r = pos / LW, tx = pos - r * LW;
        const unsigned char *c = &tile[r + F9_HALO - 1][tx];
        const int p = *c;
        const int cb = min(255, p + b), c_b = max(0, p - b);
        int v[16];
#define F9_LOAD(i, dx, dy) v[i] = c[(dx) + LW * (dy)];
        F9_RING(F9_LOAD)
#undef F9_LOAD
        unsigned brighter = 0, darker = 0;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            brighter |= (unsigned)(v[k] > cb) << k;
            darker |= (unsigned)(v[k] < c_b) << k;
        }
        if (f9_has_arc9(brighter) || f9_has_arc9(darker))
            sc[r][tx - (XL - 1)] = (unsigned char)(NONMAX ? f9_score(v, p) + 1 : 1);  // score + 1 in 1..255; 0 = no corner
    }
    __syncthreads();
    // ---- phase 3: 3x3 non-max test, again over the candidates only; survivors set their bit in the row's mask word
    for (unsigned i = tid; i < nc; i += 256) {
        const int pos = cand[i];
        const int r = pos / LW, cx = pos - r * LW - (XL - 1);
        if (r < 1 || r > F9_TY || cx < 1 || cx > F9_TX) continue;  // the ring of the score tile only serves its neighbours
        const int s = sc[r][cx];
        bool keep = s != 0;
        if (NONMAX && keep)
            keep = sc[r - 1][cx - 1] < s && sc[r - 1][cx] < s && sc[r - 1][cx + 1] < s && sc[r][cx - 1] < s &&
                   sc[r][cx + 1] < s && sc[r + 1][cx - 1] < s && sc[r + 1][cx] < s && sc[r + 1][cx + 1] < s;
        if (keep) atomicOr(&rowmask[r - 1], 1ull << (cx - 1));
    }
    __syncthreads();
    for (int r = tid; r < F9_TY; r += 256) {
        const int gy = y0 + r;
        if (gy >= h) continue;
        const unsigned long long word = rowmask[r];
        mask[((size_t)frame * h + gy) * words_per_row + tile_x] = word;
        if (word) atomicAdd(&rowcount[(size_t)frame * h + gy], (unsigned)__popcll(word));
    }
    for (unsigned i = tid; i < nc; i += 256) {  // phase 3 is over (barrier above): the score cells go back to 0
        const int pos = cand[i];
        const int r = pos / LW;
        sc[r][pos - r * LW - (XL - 1)] = 0;
    }
    __syncthreads();  // the next tile's staging clears rowmask; its phases write the candidate list and the score tile
    }
    }
}

imgfd_status launch_fast9(imgfd_ctx *ctx, const uint8_t *d_img, int w, int h, int stride,
                          size_t frame_stride, int n_frames, int threshold, int nonmax, const CompactBuffers &cb)
{
    const int bands = ceil_div(h, F9_TY);
    const TileRuns runs = tile_runs(cb.words_per_row, bands, n_frames, tile_run_length(ctx, cb.words_per_row, bands, n_frames));
    dim3 grid(runs.total);
    const int aligned4 = ((size_t)d_img % 4 == 0) && stride % 4 == 0 && frame_stride % 4 == 0;
    const int aligned16 = ((size_t)d_img % 16 == 0) && stride % 16 == 0 && frame_stride % 16 == 0;
    if (!nonmax)
        hipLaunchKernelGGL(fast9_tile<0>, grid, dim3(256), 0, ctx->stream, d_img, w, h, stride, frame_stride, threshold,
                           aligned4, aligned16, cb.mask, cb.rowcount, cb.words_per_row, runs, 1);
    else
        hipLaunchKernelGGL(fast9_tile<1>, grid, dim3(256), 0, ctx->stream, d_img, w, h, stride, frame_stride, threshold,
                           aligned4, aligned16, cb.mask, cb.rowcount, cb.words_per_row, runs, 1);
    IMGFD_HIP(ctx, hipGetLastError());
    return IMGFD_OK;
}
