// image_amd/csrc/surf.hip -- SURF keypoints (dlib) behind imgfd_surf (K16-K18 on the device, K19 on the host).
//
// Replaces dlib_surf_points(), image.dlib/src/rcpp_surf.cpp:10-53, i.e. dlib's get_surf_points
// (image.dlib/inst/dlib-19.20/dlib/image_keypoint/surf.h:237-288):
//   K16 integral image   integral_image_generic<int32>::load, dlib/image_transforms/integral_image.h:33-62, with
//                        gray = (r+g+b)/3 (dlib/pixel.h:775-783).  int32 sums wrap (4096^2 bright tiles overflow);
//                        wrap-around addition is associative, so a parallel scan gives the reference's bits:
//                        surf_int_sums / surf_int_carry / surf_int_apply (bands of rows cut into strips; the table is written ONCE, by
//                        column residue mod 4: SurfTable); surf_gray_rowscan / surf_colscan serve unaligned or small images.
//   K17 Hessian pyramid  hessian_pyramid::build_pyramid(img, 4, 6, 2), dlib/image_keypoint/hessian_pyramid.h:87-178:
//                        box-filter determinants in f64, 32 integral-image look-ups per level pixel, intervals 1-4 of every octave
//                        (0 and 5 only exist as neighbours in the maximum test, which computes what it needs of them):
//                        surf_pyramid_lds<0> (the table window of a block of first-octave level pixels in LDS; intervals 1 and 2 of
//                        octave 1 come out of the same window), surf_pyramid_taps (the rest of octaves 1-3 in one launch ordered by
//                        image band, a buffer load per look-up; surf_pyramid_plain for tables that are not laid out by residue).
//                        All publish one threshold bit per level pixel (|det| >= threshold).
//   K18 interest points  get_interest_points :453-506: surf_nms_list turns the threshold bits into a dense list, surf_nms_screen
//                        runs the 3x3x3 maximum test (:324-356) against the stored intervals, surf_nms_finish against the
//                        unbuilt ones + the quadratic interpolation with
//                        the closed-form 3x3 inverse (:411-446, dlib/matrix/matrix_la.h:922-962), in f64 with the
//                        reference's operation order (no contraction).  Survivors are appended with a sort key
//                        (octave, interval, row, column); the host orders them by that key = the order in which the
//                        reference pushes them, then applies surf.h:268's std::sort over reverse iterators.
//   K19 orientation + 64-d descriptor (surf.h:75-232): surf_describe.hip, one workgroup per point.  imgfd_surf keeps
//                        the handful of libm calls (atan2 of the 109 samples, sin/cos of the winner) on the host's
//                        glibc (surf_host.cpp) so the result stays bit-identical; imgfd_surf_dev runs them on the
//                        device as well.
// the Hessian pyramid (12 B of doubles per tile pixel) goes out with streaming stores: imgfd_surf_dev 0.306-0.311 -> 0.287 ms per tile
#define IMGFD_NT_OUT 1
#include "common.h"

#include <type_traits>
#include "surf_describe.h"

#include <math.h>
#include <stdlib.h>

#include <algorithm>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#define SURF_OCT 4
#define SURF_INT 6

struct SurfLevel {
    int step, border_px, lobe, off;  // hessian_pyramid.h:119-128
    double area_inv;
    size_t plane;                    // offset (doubles) of the level inside the pyramid buffer
    size_t mask;                     // offset (64-bit words) of the level's threshold mask: nr rows of ceil(nc/64) words
};
struct SurfGeom {
    int rows, cols;
    size_t mask_words;               // 64-bit words of all threshold masks
    int nr[SURF_OCT], nc[SURF_OCT];
    SurfLevel lev[SURF_OCT * SURF_INT];
};

static long surf_border_of(long i) { return (long)ceil((3 * (2.0 * (i + 1) + 1)) / 2.0); }  // get_border_size :180-196
static long surf_step_of(long o) { return 2 * (long)(pow(2.0, (double)o) + 0.5); }          // get_step_size :198-211

static size_t surf_geometry(int rows, int cols, SurfGeom *g)
{
    g->rows = rows; g->cols = cols;
    size_t total = 0, mask_words = 0;
    for (int o = 0; o < SURF_OCT; o++) {
        const long step = surf_step_of(o);
        g->nr[o] = (int)(rows / step); g->nc[o] = (int)(cols / step);
        for (int i = 0; i < SURF_INT; i++) {
            SurfLevel &L = g->lev[o * SURF_INT + i];
            L.step = (int)step;
            L.border_px = (int)(surf_border_of(i) * step);
            L.lobe = (int)((long)(pow(2.0, o + 1.0) + 0.5) * (i + 1) + 1);
            L.off = L.lobe / 2 + 1;
            L.area_inv = 1.0 / pow(3.0 * L.lobe, 2.0);
            // Intervals 0 and 5 of an octave are never maxima themselves (get_interest_points runs i = 1 .. 4,
            // hessian_pyramid.h:461): they only lend their 3x3 neighbourhoods to the survivors of intervals 1 and 4, which the
            // maximum test computes from the integral image on the spot.  They have no plane and no mask in the buffers.
            L.plane = total;
            L.mask = mask_words;
            if (i >= 1 && i <= SURF_INT - 2) {
                total += (size_t)g->nr[o] * g->nc[o];
                mask_words += (size_t)g->nr[o] * ((g->nc[o] + 63) / 64);
            }
        }
    }
    g->mask_words = mask_words;
    return total;
}

// ---- K16
__global__ void __launch_bounds__(256) surf_gray_rowscan(const unsigned char *__restrict__ rgb, unsigned *__restrict__ out,
                                                         int cols)
{
    __shared__ unsigned part[256];
    const int tid = threadIdx.x;
    const size_t r = blockIdx.x;
    const int seg = (cols + 255) / 256;
    const int c0 = tid * seg, c1 = min(cols, c0 + seg);
    const unsigned char *p = rgb + 3 * (r * cols);
    unsigned s = 0;
    for (int c = c0; c < c1; c++) s += ((unsigned)p[3 * c] + (unsigned)p[3 * c + 1] + (unsigned)p[3 * c + 2]) / 3u;
    part[tid] = s;
    __syncthreads();
    // exclusive prefix of the 256 segment totals (Hillis-Steele in LDS)
    unsigned incl = s;
    for (int d = 1; d < 256; d <<= 1) {
        const unsigned t = tid >= d ? part[tid - d] : 0u;
        __syncthreads();
        incl += t;
        part[tid] = incl;
        __syncthreads();
    }
    unsigned run = incl - s;
    for (int c = c0; c < c1; c++) {
        run += ((unsigned)p[3 * c] + (unsigned)p[3 * c + 1] + (unsigned)p[3 * c + 2]) / 3u;
        out[r * cols + c] = run;
    }
}

__global__ void __launch_bounds__(64) surf_colscan(unsigned *__restrict__ I, int rows, int cols)
{
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= cols) return;
    unsigned acc = 0;
    int r = 0;
    for (; r + 8 <= rows; r += 8) {
        unsigned v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = I[(size_t)(r + k) * cols + c];
#pragma unroll
        for (int k = 0; k < 8; k++) { acc += v[k]; I[(size_t)(r + k) * cols + c] = acc; }
    }
    for (; r < rows; r++) { acc += I[(size_t)r * cols + c]; I[(size_t)r * cols + c] = acc; }
}

// ---- K16, fast forms (wrap-around addition is associative: any scan order gives the reference's bits)
// Row scan: one workgroup per row, 4 consecutive pixels per thread and 1024 per trip: 12-byte RGB loads and 16-byte
// stores, coalesced; a trip's 256 thread totals are scanned with wave shuffles + one LDS exchange.
__global__ void __launch_bounds__(256) surf_gray_rowscan4(const unsigned char *__restrict__ rgb, unsigned *__restrict__ out,
                                                          int cols)
{
    __shared__ unsigned wsum[4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const size_t r = blockIdx.x;
    const unsigned *src = reinterpret_cast<const unsigned *>(rgb + 3 * (r * cols));  // cols % 4 == 0: the row is dword aligned
    unsigned carry = 0;
    for (int c0 = 0; c0 < cols; c0 += 1024) {
        const int c = c0 + 4 * tid;
        unsigned g[4] = {0u, 0u, 0u, 0u};
        if (c < cols) {  // 4 pixels = 12 bytes = 3 dwords
            const unsigned w0 = src[3 * (c >> 2)], w1 = src[3 * (c >> 2) + 1], w2 = src[3 * (c >> 2) + 2];
            g[0] = ((w0 & 0xffu) + ((w0 >> 8) & 0xffu) + ((w0 >> 16) & 0xffu)) / 3u;
            g[1] = ((w0 >> 24) + (w1 & 0xffu) + ((w1 >> 8) & 0xffu)) / 3u;
            g[2] = (((w1 >> 16) & 0xffu) + (w1 >> 24) + (w2 & 0xffu)) / 3u;
            g[3] = (((w2 >> 8) & 0xffu) + ((w2 >> 16) & 0xffu) + (w2 >> 24)) / 3u;
        }
        g[1] += g[0]; g[2] += g[1]; g[3] += g[2];
        unsigned incl = g[3];  // inclusive scan of the thread totals over the wave
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned t = __shfl_up(incl, d);
            if (lane >= d) incl += t;
        }
        if (lane == 63) wsum[wv] = incl;
        __syncthreads();
        unsigned base = carry;
        for (int k = 0; k < wv; k++) base += wsum[k];
        const unsigned total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        base += incl - g[3];  // exclusive prefix of this thread inside the trip
        if (c < cols) *reinterpret_cast<uint4 *>(out + r * cols + c) = make_uint4(base + g[0], base + g[1], base + g[2], base + g[3]);
        carry += total;
        __syncthreads();
    }
}

// Column scan in two kernels: the rows are cut into segments; (1) every (segment, column) adds up its segment,
// (2) every (segment, column) re-reads it, starting from the sum of the segments above.  part: nseg x cols.
__global__ void __launch_bounds__(256) surf_colscan_sums(const unsigned *__restrict__ I, unsigned *__restrict__ part, int rows,
                                                         int cols, int seg_rows)
{
    const int c = blockIdx.x * 256 + threadIdx.x, sgm = blockIdx.y;
    if (c >= cols) return;
    const int r0 = sgm * seg_rows, r1 = min(rows, r0 + seg_rows);
    unsigned acc = 0;
    for (int r = r0; r < r1; r++) acc += I[(size_t)r * cols + c];
    part[(size_t)sgm * cols + c] = acc;
}
__global__ void __launch_bounds__(256) surf_colscan_apply(unsigned *__restrict__ I, const unsigned *__restrict__ part, int rows,
                                                          int cols, int seg_rows)
{
    const int c = blockIdx.x * 256 + threadIdx.x, sgm = blockIdx.y;
    if (c >= cols) return;
    unsigned acc = 0;
    for (int k = 0; k < sgm; k++) acc += part[(size_t)k * cols + c];
    const int r0 = sgm * seg_rows, r1 = min(rows, r0 + seg_rows);
    int r = r0;
    for (; r + 8 <= r1; r += 8) {
        unsigned v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = I[(size_t)(r + k) * cols + c];
#pragma unroll
        for (int k = 0; k < 8; k++) { acc += v[k]; I[(size_t)(r + k) * cols + c] = acc; }
    }
    for (; r < r1; r++) { acc += I[(size_t)r * cols + c]; I[(size_t)r * cols + c] = acc; }
}

// ---- K16, band / strip form: the image is read twice and the table written once (3 + 3 + 4 bytes per pixel; the row scan +
// column scan above moves 3 + 4 + 4 + 8).  A WAVE owns a tile of SI_RB rows x 256 columns (4 pixels = 12 bytes per lane):
//   surf_int_sums   per tile: the column sums over its rows (colsum[band][c]) and the row sums over its columns
//                   (rowsum[row][strip])
//                   and the tile total (bandstrip[band][strip])
//   surf_int_carry  colsum becomes the sum of the bands ABOVE (exclusive scan down the bands), rowsum the sum of the strips to
//                   the LEFT; one more workgroup turns bandstrip into its 2-D exclusive prefix: the table's value above the
//                   band, left of the strip
//   surf_int_apply  per tile: I[r][c] = (bands above or rows of the band <= r, strips to the left: bandstrip + one wave scan
//                   of the band's rowsum) + (rows above the band, columns of the strip <= c: one wave scan of colsum) +
//                   (rows of the band <= r, columns of the strip <= c: running column sums and one wave scan per row).
//                   No barrier in the two big kernels; all sums wrap like the reference's int32.
constexpr int SI_RB = 32;       // rows per band
constexpr int SI_MAX_STRIPS = 32;  // strips of 256 columns the carry kernel's LDS scan holds (wider images: row + column scans)

__device__ __forceinline__ void si_gray4(unsigned w0, unsigned w1, unsigned w2, unsigned (&g)[4])
{   // 4 pixels = 12 bytes; (r + g + b) / 3 in unsigned arithmetic (pixel.h:775-783)
    g[0] = ((w0 & 0xffu) + ((w0 >> 8) & 0xffu) + ((w0 >> 16) & 0xffu)) / 3u;
    g[1] = ((w0 >> 24) + (w1 & 0xffu) + ((w1 >> 8) & 0xffu)) / 3u;
    g[2] = (((w1 >> 16) & 0xffu) + (w1 >> 24) + (w2 & 0xffu)) / 3u;
    g[3] = (((w2 >> 8) & 0xffu) + ((w2 >> 16) & 0xffu) + (w2 >> 24)) / 3u;
}
__device__ __forceinline__ unsigned si_wave_sum(unsigned v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    return v;
}
__device__ __forceinline__ unsigned si_wave_incl(unsigned v, int lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned t = __shfl_up(v, d);
        if (lane >= d) v += t;
    }
    return v;
}
// A lane's 12-byte group in EVERY row of its tile: 32 loads that depend on nothing, issued ahead of the arithmetic a dozen and
// more at a time (the compiler's schedule: the rows are consumed in the order they were asked for).  (Round 3
// fetched four rows at a time, one step ahead: 24 KB in flight per CU, and 8 TB/s x ~2 us of latency wants ~64 KB -- the sums
// kernel ran at 3 TB/s, 16.6 us for 50 MB.)  Rows past the band's end re-read its last row and count as zero.
struct SiTile { unsigned w[SI_RB][3]; };
__device__ __forceinline__ void si_fetch_tile(const unsigned *src, size_t rd, int r0, int r1, SiTile &o)
{
#pragma unroll
    for (int k = 0; k < SI_RB; k++) {
        const unsigned *p = src + (size_t)min(r0 + k, r1 - 1) * rd;
        o.w[k][0] = p[0]; o.w[k][1] = p[1]; o.w[k][2] = p[2];
    }
}
__device__ __forceinline__ void si_gray_row(const SiTile &t, int k, bool in, unsigned (&g)[4])
{
    const unsigned keep = in ? 0xffffffffu : 0u;  // (an AND, not a select: the compiler sinks a selected load into a branch of its own, behind every other load)
    si_gray4(t.w[k][0] & keep, t.w[k][1] & keep, t.w[k][2] & keep, g);
}
// 32 values per lane -> lane L holds the sum over the wave of value L >> 1.  A butterfly that halves the values a lane
// carries at every step (16 + 8 + 4 + 2 + 1 + 1 = 32 exchanges; a butterfly per value takes 6 x 32 = 192 trips through the LDS
// crossbar)
template <int D, int N>
__device__ __forceinline__ void si_halve(unsigned (&t)[SI_RB], int lane)
{
    const bool upper = (lane & D) != 0;  // this half of the pair keeps the upper half of the values
#pragma unroll
    for (int j = 0; j < N / 2; j++) {
        const unsigned send = upper ? t[j] : t[j + N / 2], keep = upper ? t[j + N / 2] : t[j];
        t[j] = keep + (unsigned)__shfl_xor(send, D);
    }
}
__device__ __forceinline__ unsigned si_wave_sums32(unsigned (&t)[SI_RB], int lane)
{
    static_assert(SI_RB == 32, "five halving steps");
    si_halve<32, 32>(t, lane); si_halve<16, 16>(t, lane); si_halve<8, 8>(t, lane); si_halve<4, 4>(t, lane); si_halve<2, 2>(t, lane);
    return t[0] + (unsigned)__shfl_xor(t[0], 1);
}

__global__ void __launch_bounds__(256) surf_int_sums(const unsigned char *__restrict__ rgb, unsigned *__restrict__ colsum,
                                                     unsigned *__restrict__ rowsum, unsigned *__restrict__ bandstrip, int rows, int cols,
                                                     int nstrips)
{
    const int lane = threadIdx.x & 63, s = blockIdx.x * 4 + (threadIdx.x >> 6), b = blockIdx.y;
    const int c = 256 * s + 4 * lane;
    const bool active = s < nstrips && c < cols;
    const int r0 = b * SI_RB, r1 = min(rows, r0 + SI_RB);
    const unsigned *src = reinterpret_cast<const unsigned *>(rgb) + 3 * (size_t)(active ? c >> 2 : 0);
    const size_t rd = 3 * (size_t)(cols >> 2);  // dwords per row
    SiTile tile;
    si_fetch_tile(src, rd, r0, r1, tile);
    unsigned acc[4] = {0u, 0u, 0u, 0u}, t[SI_RB];
#pragma unroll
    for (int k = 0; k < SI_RB; k++) {
        unsigned g[4];
        si_gray_row(tile, k, active && r0 + k < r1, g);
#pragma unroll
        for (int e = 0; e < 4; e++) acc[e] += g[e];
        t[k] = g[0] + g[1] + g[2] + g[3];
    }
    const unsigned row_total = si_wave_sums32(t, lane);  // of row r0 + (lane >> 1)
    if (!(lane & 1) && s < nstrips && r0 + (lane >> 1) < r1) rowsum[(size_t)(r0 + (lane >> 1)) * nstrips + s] = row_total;
    if (active) *reinterpret_cast<uint4 *>(colsum + (size_t)b * cols + c) = make_uint4(acc[0], acc[1], acc[2], acc[3]);
    const unsigned tile_total = si_wave_sum(acc[0] + acc[1] + acc[2] + acc[3]);
    if (lane == 0 && s < nstrips) bandstrip[(size_t)b * nstrips + s] = tile_total;
}

// blocks 0 .. ceil(cols / 64) - 1: 64 columns each, the bands dealt to the workgroup's four waves in runs (a wave's run -- up
// to 32 bands -- sits in registers: one round trip to memory, where a thread walking all bands of its column made sixteen);
// the next block: the band x strip totals; the blocks behind it: a thread per image row turns its strip sums into the sums of
// the strips to the LEFT (the apply kernel summed them itself, lane by lane: up to 15 dependent round trips in front of its tile)
constexpr int SC_RUN = 32;  // bands of a run held in registers
__global__ void __launch_bounds__(256) surf_int_carry(unsigned *__restrict__ colsum, unsigned *__restrict__ bandstrip, unsigned *__restrict__ rowsum,
                                                      int nbands, int rows, int cols, int nstrips, unsigned long long *__restrict__ zero32)
{
    __shared__ unsigned tot[SI_MAX_STRIPS][256 + 1];
    const int tid = threadIdx.x;
    // on its way: the tile's 256-byte counter slot (records, survivors, candidates of K18) starts from zero -- a fill of its own
    // was one more launch in a single tile's chain
    if (zero32 && blockIdx.x == 0 && tid < 32) zero32[tid] = 0ull;
    const int col_blocks = (cols + 63) / 64;
    if ((int)blockIdx.x > col_blocks) {
        const int r = 256 * ((int)blockIdx.x - col_blocks - 1) + tid;
        if (r >= rows) return;
        unsigned *p = rowsum + (size_t)r * nstrips, v[SI_MAX_STRIPS];
#pragma unroll
        for (int j = 0; j < SI_MAX_STRIPS; j++) v[j] = p[min(j, nstrips - 1)];
        unsigned acc = 0;
#pragma unroll
        for (int j = 0; j < SI_MAX_STRIPS; j++)
            if (j < nstrips) { p[j] = acc; acc += v[j]; }
        return;
    }
    if ((int)blockIdx.x < col_blocks) {
        __shared__ unsigned run_total[4][64];
        const int c = 64 * blockIdx.x + (tid & 63), w = tid >> 6;
        const int per = (nbands + 3) / 4, b0 = min(nbands, w * per), b1 = min(nbands, b0 + per);
        const bool live = c < cols;
        unsigned *col = colsum + (live ? c : 0);
        unsigned v[SC_RUN], sum = 0;
        if (per <= SC_RUN) {
#pragma unroll
            for (int k = 0; k < SC_RUN; k++) v[k] = colsum[(size_t)min(b0 + k, nbands - 1) * cols + (live ? c : 0)];
#pragma unroll
            for (int k = 0; k < SC_RUN; k++) sum += b0 + k < b1 ? v[k] : 0u;
        } else {
            for (int bb = b0; bb < b1; bb++) sum += col[(size_t)bb * cols];
        }
        run_total[w][tid & 63] = sum;
        __syncthreads();
        unsigned acc = 0;
        for (int j = 0; j < w; j++) acc += run_total[j][tid & 63];
        if (!live) return;
        if (per <= SC_RUN) {
#pragma unroll
            for (int k = 0; k < SC_RUN; k++)
                if (b0 + k < b1) { col[(size_t)(b0 + k) * cols] = acc; acc += v[k]; }
        } else {
            for (int bb = b0; bb < b1; bb++) { const unsigned x = col[(size_t)bb * cols]; col[(size_t)bb * cols] = acc; acc += x; }
        }
        return;
    }
    // bandstrip[b][s] := sum over bands < b and strips < s (the table's value above the band, left of the strip), in place.
    // Bands in chunks of 256 (a thread per band: exclusive prefix along its row), then per strip an exclusive scan down the
    // chunk's bands (a wave per strip), carried from chunk to chunk.
    __shared__ unsigned carry[SI_MAX_STRIPS];
    for (int s = tid; s < nstrips; s += 256) carry[s] = 0;
    for (int b0 = 0; b0 < nbands; b0 += 256) {
        const int b = b0 + tid;
        unsigned e = 0;
        for (int s = 0; s < nstrips; s++) {
            const unsigned x = b < nbands ? bandstrip[(size_t)b * nstrips + s] : 0u;
            tot[s][tid] = e;
            e += x;
        }
        __syncthreads();
        const int lane = tid & 63;
        for (int s = tid >> 6; s < nstrips; s += 4) {
            unsigned v[4], sum = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) { v[k] = tot[s][4 * lane + k]; sum += v[k]; }
            unsigned run = carry[s] + si_wave_incl(sum, lane) - sum;
#pragma unroll
            for (int k = 0; k < 4; k++) { tot[s][4 * lane + k] = run; run += v[k]; }
            if (lane == 63) carry[s] = run;
        }
        __syncthreads();
        if (b < nbands)
            for (int s = 0; s < nstrips; s++) bandstrip[(size_t)b * nstrips + s] = tot[s][tid];
        __syncthreads();
    }
}

// RES: the table is written in the residue layout (SurfTable: J[row][x % 4][x / 4]) INSTEAD of the plain one: a lane's four
// columns are the four residues at word 64 s + lane of each, so a wave stores four runs of 256 bytes per row.  (Rounds 3-4 re-laid
// the plain table with a kernel of its own, 4 B/px read + 4 written; round 5 wrote both layouts here, 8 B/px; since round 6 every
// reader of the table addresses the residue layout and the plain copy is gone: 4 B/px.)
template <bool RES>
__global__ void __launch_bounds__(256) surf_int_apply(const unsigned char *__restrict__ rgb, const unsigned *__restrict__ colcarry,
                                                      const unsigned *__restrict__ above_left, const unsigned *__restrict__ rowsum,
                                                      unsigned *__restrict__ out, int rows, int cols, int nstrips)
{
    const int lane = threadIdx.x & 63, s = blockIdx.x * 4 + (threadIdx.x >> 6), b = blockIdx.y;
    const int c = 256 * s + 4 * lane;
    const bool active = s < nstrips && c < cols;
    const int r0 = b * SI_RB, r1 = min(rows, r0 + SI_RB);
    const unsigned *src = reinterpret_cast<const unsigned *>(rgb) + 3 * (size_t)(active ? c >> 2 : 0);
    const size_t rd = 3 * (size_t)(cols >> 2);
    // the carries first (few, short), then the whole tile: everything is in flight while the carries are scanned
    const uint4 cc = active ? *reinterpret_cast<const uint4 *>(colcarry + (size_t)b * cols + c) : make_uint4(0u, 0u, 0u, 0u);
    unsigned left_of = lane < SI_RB && r0 + lane < r1 && s < nstrips ? rowsum[(size_t)(r0 + lane) * nstrips + s] : 0u;  // strips to the left, this row
    const unsigned corner = s < nstrips ? above_left[(size_t)b * nstrips + s] : 0u;
    SiTile tile;
    si_fetch_tile(src, rd, r0, r1, tile);
    // rows above the band, columns of the strip up to c + k
    unsigned base[4];
    {
        base[0] = cc.x; base[1] = base[0] + cc.y; base[2] = base[1] + cc.z; base[3] = base[2] + cc.w;
        const unsigned carry = si_wave_incl(base[3], lane) - base[3];
#pragma unroll
        for (int e = 0; e < 4; e++) base[e] += carry;
    }
    // rows <= the lane's row, strips to the left: above the band from surf_int_carry, inside it a wave scan over the row sums
    left_of = si_wave_incl(left_of, lane) + corner;
    const size_t per = (size_t)(cols >> 2);
    unsigned v[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int r = 0; r < SI_RB; r += 4) {
        unsigned q[4][4], e[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            unsigned g[4];
            si_gray_row(tile, r + k, active && r0 + r + k < r1, g);
            v[0] += g[0]; v[1] += g[1]; v[2] += g[2]; v[3] += g[3];  // running column sums inside the band
            q[k][0] = v[0]; q[k][1] = q[k][0] + v[1]; q[k][2] = q[k][1] + v[2]; q[k][3] = q[k][2] + v[3];
            e[k] = q[k][3];
        }
#pragma unroll
        for (int k = 0; k < 4; k++) e[k] = si_wave_incl(e[k], lane) - q[k][3];  // four independent scans
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const unsigned add = e[k] + (unsigned)__builtin_amdgcn_readlane((int)left_of, min(r + k, r1 - 1 - r0));
            if (active && r0 + r + k < r1) {
                const uint4 o = make_uint4(base[0] + add + q[k][0], base[1] + add + q[k][1], base[2] + add + q[k][2], base[3] + add + q[k][3]);
                const size_t row = (size_t)(r0 + r + k) * cols;
                if (RES) {
                    unsigned *j = out + row + (c >> 2);
                    j[0] = o.x; j[per] = o.y; j[2 * per] = o.z; j[3 * per] = o.w;
                } else {
                    *reinterpret_cast<uint4 *>(out + row + c) = o;
                }
            }
        }
    }
}

// hessian_pyramid.h:141-145: Dxx = (double)wide_box - (double)narrow_box * 3.0 -- both sums are exact integers far below 2^31 / 3 (a box
// of the largest filter covers 291 x 193 pixels of at most 255), so the int32 difference is exact and its one conversion
// equals the reference's three double operations bit for bit
// (3 * narrow as a 24-bit multiply: the narrow box holds at most 97 x 193 x 255 < 2^23; the plain product compiles to v_mul_lo_u32, a
// quarter-rate instruction -- 48 of them per thread of the first octave's kernel)
__device__ __forceinline__ double surf_dxx(int wide, int narrow) { return (double)(wide - __mul24(narrow, 3)); }

// (get_sum_of_area, integral_image.h:64-96, in uint32 arithmetic = the reference's wrapping int32, appears below in
// its interior form: br - bl - tr + tl.)

// ---- K17, first octave (step 2: three quarters of all level pixels; the template also instantiates for octave 1, where
// it loses to the gather kernel): the integral-image window of a block of level
// pixels, with the widest filter's reach around it, is staged in LDS once and serves all six intervals (6 x 32 look-ups
// per level pixel from LDS instead of scattered global loads).  Same arithmetic as surf_pyramid below.
#ifndef SURF_LDS_NT
#define SURF_LDS_NT 512  // threads of a first-octave workgroup: the 47 KB window allows three workgroups per CU whatever their size (256: 12 waves per CU; four lanes 0.256 -> 0.244 ms per tile)
#endif
#ifndef SURF_LDS_LY
#define SURF_LDS_LY 16  // level rows per workgroup of the first octave's LDS kernel (imgfd_surf_dev per tile: 8 -> 0.453, 16 -> 0.441-0.455, 32 -> 0.472 ms)
#endif
template <int O>
struct SurfPyrLds {
    static constexpr int STEP = 2 << O;
    static constexpr int LX = O == 0 ? 64 : 32, LY = O == 0 ? SURF_LDS_LY : 8;  // level pixels per workgroup
    static constexpr int LOBE_MAX = STEP * SURF_INT + 1;               // hessian_pyramid.h:119-128: lobe = step*(i+1) + 1
    static constexpr int REACH = (3 * LOBE_MAX) / 2;                   // half of the widest box; "+1" for the l-1 / t-1 corner
    static constexpr int HL = (REACH + 1 + 3) / 4 * 4;                 // left / top margin: a multiple of 4 (16-byte loads) and of STEP
    static constexpr int HR = REACH;
    static constexpr int W = STEP * (LX - 1) + 1 + HL + HR, H = STEP * (LY - 1) + 1 + HL + HR;
    static constexpr int P = (W + 3) / 4 * 4;                          // row pitch in words, a multiple of 4 and of STEP
    static constexpr int NT = SURF_LDS_NT;                            // threads per workgroup
    static constexpr int PX_PER_THREAD = LX * LY / NT;
    static constexpr int BIAS = HL * P + HL / STEP;                    // window words from (row - HL, column - HL) to the centre
    static_assert(HL % STEP == 0 && P % STEP == 0 && (LX * LY) % NT == 0 && LX <= 64 && 64 % LX == 0, "tile geometry");
    static_assert(sizeof(unsigned) * (size_t)H * P <= 120 * 1024, "window fits the LDS");
    // window word of image column x0 + dx: columns are stored by residue mod STEP, so that the lanes of a look-up
    // (columns STEP*lane + const) sit on consecutive words
    static __device__ __forceinline__ int col(int dx) { return (dx % STEP) * (P / STEP) + dx / STEP; }
};

// One interval for the level pixel whose centre sits at window word `ctr` (its column is a multiple of STEP inside the
// window).  The lobe is a compile-time constant, so every look-up is a ds_read with an immediate offset from `ctr`.
// For a centre at least border_px inside the image all four corners of every box exist (l - 1 >= 3*lobe/2 - 1 > 0), so
// integral_image.h:64-96's border cases cannot occur here and br - bl - tr + tl is evaluated directly.
// The window is addressed from `top` = the word of (row - HL, column - HL), the top-left corner of what any filter reaches
// from this centre, as an index the optimiser cannot see through: every look-up then has a NON-NEGATIVE constant offset,
// i.e. an immediate of its ds_read.  (Addressed from the centre, half of the look-ups sit at negative offsets, which the
// 16-bit unsigned offset field cannot hold: one v_add_u32 each, 17 of the kernel's 98 vector instructions per value.)
template <class G, int lobe>
__device__ __forceinline__ double surf_lds_filter(const unsigned *__restrict__ win, unsigned top, double area_inv)
{
    constexpr int off = lobe / 2 + 1;
    auto at = [&](int dy, int dx) __attribute__((always_inline)) -> unsigned {
        const int m = ((dx % G::STEP) + G::STEP) % G::STEP;  // residue of a possibly negative offset
        return win[top + (unsigned)(G::BIAS + dy * G::P + m * (G::P / G::STEP) + (dx - m) / G::STEP)];
    };
    auto box = [&](int cx, int cy, int w, int h) __attribute__((always_inline)) -> int {  // centered_rect(cx, cy, w, h), relative to the centre
        const int l = cx - w / 2, t = cy - h / 2, r = l + w - 1, b = t + h - 1;
        return (int)(at(b, r) - at(b, l - 1) - at(t - 1, r) + at(t - 1, l - 1));
    };
    // :141-145 form (double)box - (double)box * 3.0: integers below 2^26 (a box of at most 291 x 193 pixels of at most 255), so the
    // difference is exact in int32 and ONE conversion gives the reference's double (surf_dxx)
    double Dxx = surf_dxx(box(0, 0, lobe * 3, 2 * lobe - 1), box(0, 0, lobe, 2 * lobe - 1));
    double Dyy = surf_dxx(box(0, 0, 2 * lobe - 1, lobe * 3), box(0, 0, 2 * lobe - 1, lobe));
    double Dxy = (int)((unsigned)box(-off, off, lobe, lobe) + (unsigned)box(off, -off, lobe, lobe) - (unsigned)box(-off, -off, lobe, lobe) -
                       (unsigned)box(off, off, lobe, lobe));
    Dxx *= area_inv; Dyy *= area_inv; Dxy *= area_inv;
    double sign = +1;
    if (Dxx + Dyy < 0) sign = -1;
    double det = Dxx * Dyy - 0.81 * Dxy * Dxy;
    if (det < 0) det = 0;
    return sign * det;
}

// interval IT of the window's own octave O (hessian_pyramid.h:119-128: lobe = step * (i + 1) + 1)
template <int O, int IT>
__device__ __forceinline__ double surf_lds_interval(const unsigned *__restrict__ win, unsigned top, double area_inv)
{
    return surf_lds_filter<SurfPyrLds<O>, SurfPyrLds<O>::STEP * (IT + 1) + 1>(win, top, area_inv);
}

#ifndef SURF_LDS_WAVES
#define SURF_LDS_WAVES 6  // waves per SIMD the first octave's kernel is compiled for (three workgroups of 8 waves per CU)
#endif
// One block of 64 x 16 level pixels per workgroup, three workgroups per CU.  Measured and not kept (profiles/r06/surf_first_octave_phases.txt,
// surf_sparse_first_octave_not_kept.txt): a run of blocks per workgroup with the next window prefetched during the filters (compile-time
// hooks in `git show 30adde6:image_amd/csrc/surf.hip`), and storing only the rows of values the maximum test can read
// (scripts/experiments/r06_surf_sparse_first_octave.patch: 8.9 -> 3.7 B/px written, the kernel 66 -> 75 us, the batch unchanged).
template <int O>
__global__ void __launch_bounds__(SURF_LDS_NT) IMGFD_WAVES_PER_EU(SURF_LDS_WAVES, SURF_LDS_WAVES) surf_pyramid_lds(SurfTable T, double *__restrict__ pyr, SurfGeom g,
                                                                unsigned long long *__restrict__ mask, double thr, int blocks_x, int next_too)
{
    using G = SurfPyrLds<O>;
    HIP_DYNAMIC_SHARED(unsigned, win)  // [G::H][G::P]
    const int tid = threadIdx.x;
    // 1-D grid, XCD-aware order of the blocks (imgfd_xcd_tile): the windows of neighbouring blocks overlap by their halo
    const int blk = (int)imgfd_xcd_tile(blockIdx.x, gridDim.x);
    const int blk_y = blk / blocks_x, blk_x = blk - blk_y * blocks_x;
    const int lc0 = blk_x * G::LX, lr0 = blk_y * G::LY;
    const int x0 = G::STEP * lc0 - G::HL, y0 = G::STEP * lr0 - G::HL;
    const int cols = g.cols, rows = g.rows;
    const unsigned *__restrict__ I = T.p;
    if (T.per) {
        // residue layout: window column dx = 4 w + m of a row is word w of plane m, from word s = x0 / 4 of the plane's row on (s is
        // rarely a multiple of 4: a row of a plane takes NQ aligned quads from s rounded down).  A thread keeps ONE (plane, quad) and
        // walks down the window rows, RPT rows per trip: its table offset grows by a multiple of the row pitch, its window address by a
        // constant (the ds_write's immediate), which of its four words fall inside the window row is decided once -- 16-byte loads,
        // ALL of a thread's quads requested before the first is written.  (A task list cut into (row, plane, quad) by divisions, with
        // 64-bit addresses, was 333 of the kernel's 870 vector instructions per thread: round 6.)  Rows and quads are clamped into
        // the table: what lies outside the image is never looked up by a valid centre.
        static_assert(G::P % 4 == 0, "whole quads per window row");
        constexpr int PW = G::P / 4, NQ = (PW + 3 + 3) / 4, RPT = G::NT / (4 * NQ), TRIPS = (G::H + RPT - 1) / RPT;
        static_assert(RPT >= 1, "a window row's quads fit the workgroup");
        const int slot = tid % (4 * NQ), rw = tid / (4 * NQ);  // rw >= RPT: the workgroup's last threads have no quad
        const int m = slot / NQ, k4 = slot - m * NQ;
        const int s = x0 >> 2, lead = s & 3, quads = T.per >> 2;
        const unsigned in_row = (unsigned)m * (unsigned)T.per + 4u * (unsigned)min(max((s >> 2) + k4, 0), quads - 1);
        uint4 v[TRIPS];
#pragma unroll
        for (int j = 0; j < TRIPS; j++) {
            const int gy = min(max(y0 + min(rw + RPT * j, G::H - 1), 0), rows - 1);
            v[j] = *reinterpret_cast<const uint4 *>(I + (__umul24((unsigned)gy, (unsigned)cols) + in_row));  // the table has < 2^29 words (launch_surf_integral)
        }
        unsigned *dst = win + rw * G::P + G::col(m) + (4 * k4 - lead) * (4 / G::STEP);  // col(4 w + m) = col(m) + 4 w / STEP
        bool ok[4];
#pragma unroll
        for (int i = 0; i < 4; i++) ok[i] = rw < RPT && 4 * k4 + i - lead >= 0 && 4 * k4 + i - lead < PW;
#pragma unroll
        for (int j = 0; j < TRIPS; j++) {  // trip by trip: the first quads are written while the last are still on their way
            const unsigned e[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
            if (RPT * j + RPT <= G::H || rw + RPT * j < G::H) {
#pragma unroll
                for (int i = 0; i < 4; i++)
                    if (ok[i]) dst[RPT * j * G::P + i * (4 / G::STEP)] = e[i];
            }
        }
    } else if (x0 >= 0 && y0 >= 0 && x0 + G::P <= cols && y0 + G::H <= rows && (cols & 3) == 0) {  // plain table, interior
        for (int i = tid; i < G::H * (G::P / 4); i += G::NT) {
            const int ry = i / (G::P / 4), q = i - ry * (G::P / 4);
            const uint4 u = *reinterpret_cast<const uint4 *>(I + (size_t)(y0 + ry) * cols + x0 + 4 * q);
            unsigned *row = win + ry * G::P;
            row[G::col(4 * q)] = u.x; row[G::col(4 * q + 1)] = u.y; row[G::col(4 * q + 2)] = u.z; row[G::col(4 * q + 3)] = u.w;
        }
    } else {  // plain table at the image border: clamped coordinates
        for (int i = tid; i < G::H * G::P; i += G::NT) {
            const int ry = i / G::P, rx = i - ry * G::P;
            const int gy = min(max(y0 + ry, 0), rows - 1), gx = min(max(x0 + rx, 0), cols - 1);
            win[ry * G::P + G::col(rx)] = T.at(gy, gx);
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < G::PX_PER_THREAD; k++) {
        const int e = tid + G::NT * k;  // level pixel of the block: row-major, LX per row
        const int lr = lr0 + e / G::LX, lc = lc0 + e % G::LX;
        const int r = lr * G::STEP, c = lc * G::STEP;
        // a wave = 64 consecutive level pixels of one row (LX = 64, lc0 a multiple of 64): its ballot is one word of the
        // level's threshold mask (|det| >= thr: the only pixels surf_nms_list has to look at); every lane votes
        const bool inside = lr < g.nr[O] && lc < g.nc[O];
        unsigned top = (unsigned)((r - y0) * G::P + (c - x0) / G::STEP - G::BIAS);  // (c - x0) is a multiple of STEP: residue plane 0
        IMGFD_OPAQUE(top);
        double *dst = pyr + (size_t)lr * g.nc[O] + lc;
#define SPL_DO(IT)                                                                                        \
        {                                                                                                  \
            const SurfLevel &L = g.lev[O * SURF_INT + IT];                                                 \
            const int bp = L.border_px;                                                                    \
            bool hot = false;                                                                              \
            if (inside && !(r < bp || r >= rows - bp || c < bp || c >= cols - bp)) {                       \
                const double val = surf_lds_interval<O, IT>(win, top, L.area_inv);                         \
                IMGFD_OUT_STORE(val, &dst[L.plane]);                                                       \
                hot = fabs(val) >= thr;                                                                    \
            }                                                                                              \
            const unsigned long long word = __ballot(hot);                                                 \
            if ((tid & 63) == 0 && lr < g.nr[O] && lc < g.nc[O]) mask[L.mask + (size_t)lr * ((g.nc[O] + 63) / 64) + (lc >> 6)] = word; \
        }
        SPL_DO(1) SPL_DO(2) SPL_DO(3) SPL_DO(4)  // intervals 0 and 5 are not built (surf_geometry)
#undef SPL_DO
    }
    // ---- The first two built intervals of the NEXT octave (lobes 2 STEP * 2 + 1 and 2 STEP * 3 + 1: the second reaches exactly as far as this
    // octave's widest filter, for which the window is cut) from the same window: a quarter as many level pixels, one value per
    // thread, all stored.  In the gather kernel these two intervals of octave 1 cost the batch 15 us per 4096^2 tile, here 4.
    // A wave = two rows of LX / 2 level pixels: its ballot is the low or high half of a mask word for each of them.
    if (next_too) {
        static_assert(G::NT == 2 * (G::LX / 2) * (G::LY / 2) && G::LX == 64, "a value per thread; half a mask word per block and row");
        constexpr int N = O + 1, NSTEP = 2 * G::STEP;
        const int e = tid & (G::NT / 2 - 1);
        const int lr = blk_y * (G::LY / 2) + e / (G::LX / 2), lc = blk_x * (G::LX / 2) + e % (G::LX / 2);
        const int r = lr * NSTEP, c = lc * NSTEP;
        const bool inside = lr < g.nr[N] && lc < g.nc[N];
        unsigned top = (unsigned)((r - y0) * G::P + (c - x0) / G::STEP - G::BIAS);
        IMGFD_OPAQUE(top);
        const int words = (g.nc[N] + 63) / 64;
#define SPL_NEXT(IT)                                                                                      \
        {                                                                                                  \
            const SurfLevel &L = g.lev[N * SURF_INT + IT];                                                 \
            const int bp = L.border_px;                                                                    \
            bool hot = false;                                                                              \
            if (inside && !(r < bp || r >= rows - bp || c < bp || c >= cols - bp)) {                       \
                const double nv = surf_lds_filter<G, NSTEP * (IT + 1) + 1>(win, top, L.area_inv);          \
                IMGFD_OUT_STORE(nv, &pyr[L.plane + (size_t)lr * g.nc[N] + lc]);                            \
                hot = fabs(nv) >= thr;                                                                     \
            }                                                                                              \
            const unsigned long long word = __ballot(hot);                                                 \
            if ((tid & 31) == 0 && lr < g.nr[N] && (lc >> 6) < words) {                                    \
                unsigned *half = reinterpret_cast<unsigned *>(mask + L.mask + (size_t)lr * words + (lc >> 6)) + ((lc >> 5) & 1); \
                half[0] = (unsigned)(word >> (tid & 32));                                                  \
                if (blk_x == blocks_x - 1 && ((lc >> 5) & 1) == 0) half[1] = 0u;  /* no block to the right writes it */ \
            }                                                                                              \
        }
        if (tid < G::NT / 2) SPL_NEXT(1) else SPL_NEXT(2)
#undef SPL_NEXT
    }
}

template <int O>
static imgfd_status launch_surf_pyramid_lds(imgfd_ctx *ctx, const SurfTable &T, double *d_pyr, const SurfGeom &g,
                                            unsigned long long *d_mask, double thr, bool next_too)
{
    static_assert(SurfPyrLds<O>::LX == 64, "a wave's ballot is one mask word");
    using G = SurfPyrLds<O>;
    const size_t lds = sizeof(unsigned) * (size_t)G::H * G::P;
    IMGFD_HIP(ctx, hipFuncSetAttribute((const void *)surf_pyramid_lds<O>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int bx = ceil_div(g.nc[O], G::LX), by = ceil_div(g.nr[O], G::LY);
    hipLaunchKernelGGL(surf_pyramid_lds<O>, dim3((unsigned)(bx * by)), dim3(G::NT), lds, ctx->stream, T, d_pyr, g, d_mask, thr, bx, next_too ? 1 : 0);
    return IMGFD_OK;
}

// ---- K17, octaves 1-3.  A level pixel of octave o sits on image columns that are multiples of 4, 8, 16: a wave of 64
// consecutive level pixels reads one word out of every `step` of the plain table (25 / 12 / 6 % of each cache line used, 32
// look-ups per pixel and interval).  In the residue layout (SurfTable) the lanes of an octave-1 look-up read 64 consecutive
// words (octave 2: every second, octave 3: every fourth).
//
// One launch for the three octaves; a workgroup = 4 level rows x 64 level columns of one octave, a thread evaluates the four
// built intervals of one level pixel (the filters look at the same neighbourhood of the table).  Order of the workgroups
// (SurfBands): the image is cut into bands of 4 * step(octave 3) = 64 rows; band j holds the block rows 4j .. 4j+3 of octave 1,
// 2j, 2j+1 of octave 2 and j of octave 3 -- everything that is centred on those image rows -- and workgroup ids are remapped so
// that each XCD (ids are dealt round-robin to the 8 XCDs, each with an L2 of its own) owns a contiguous run of bands.  (Rounds 3-5
// ran octave after octave, each XCD a band of LEVEL rows per octave: the table was streamed from HBM once per octave, 9.1 B per
// tile pixel fetched for a 4 B table.)
struct SurfBands {
    int gx[SURF_OCT], gy[SURF_OCT];  // workgroups across / down per octave (0: the octave is not in this launch)
    int it0[SURF_OCT];               // first interval this launch builds of the octave (1; 3 where the first octave's kernel built 1 and 2)
    int per_band;                    // 4 gx[1] + 2 gx[2] + gx[3]
    int nbands;
    unsigned total;                  // nbands * per_band, rounded up to a multiple of 8
};
static SurfBands surf_bands(const SurfGeom &g)
{
    SurfBands b;
    memset(&b, 0, sizeof b);
    for (int o = 0; o < SURF_OCT; o++) b.it0[o] = 1;
    for (int o = 1; o < SURF_OCT; o++)
        if (g.nr[o] >= 1 && g.nc[o] >= 1) { b.gx[o] = (g.nc[o] + 63) / 64; b.gy[o] = (g.nr[o] + 3) / 4; }
    b.per_band = 4 * b.gx[1] + 2 * b.gx[2] + b.gx[3];
    b.nbands = std::max(std::max((b.gy[1] + 3) / 4, (b.gy[2] + 1) / 2), b.gy[3]);
    b.total = (unsigned)align_up((size_t)b.nbands * b.per_band, (size_t)8);
    return b;
}
// workgroup id -> (octave, block column, block row); false: nothing to do
__device__ __forceinline__ bool surf_band_block(const SurfBands &b, unsigned id, int &o, int &bx, int &by)
{
    const unsigned nid = (id & 7u) * (b.total >> 3) + (id >> 3);  // XCD id % 8 owns a contiguous run of the band order
    const int band = (int)(nid / (unsigned)b.per_band);
    int r = (int)(nid - (unsigned)band * (unsigned)b.per_band);
    if (band >= b.nbands) return false;
    if (r < 4 * b.gx[1]) { o = 1; by = 4 * band + r / b.gx[1]; bx = r % b.gx[1]; }
    else if ((r -= 4 * b.gx[1]) < 2 * b.gx[2]) { o = 2; by = 2 * band + r / b.gx[2]; bx = r % b.gx[2]; }
    else { r -= 2 * b.gx[2]; o = 3; by = band; bx = r; }
    return by < b.gy[o];
}

// plain table (images whose width is no multiple of 16, or too small for the band / strip scans): address arithmetic per look-up
__global__ void __launch_bounds__(256) surf_pyramid_plain(const unsigned *__restrict__ I, double *__restrict__ pyr, SurfGeom g, SurfBands bands,
                                                          unsigned long long *__restrict__ mask, double thr)
{
    int o, bx, by;
    if (!surf_band_block(bands, blockIdx.x, o, bx, by)) return;
    const int lc = bx * 64 + (threadIdx.x & 63);
    const int lr = by * 4 + (threadIdx.x >> 6);
    const int step = g.lev[o * SURF_INT].step, cols = g.cols;
    const int r = lr * step, c = lc * step;
    const bool in_level = lr < g.nr[o] && lc < g.nc[o];
    const unsigned *ctr = I + (size_t)r * cols + c;
#pragma unroll 1
    for (int it = bands.it0[o]; it < SURF_INT - 1; it++) {
        const SurfLevel &L = g.lev[o * SURF_INT + it];
        const bool inside = in_level && !(r < L.border_px || r >= g.rows - L.border_px || c < L.border_px || c >= cols - L.border_px);
        bool hot = false;
        if (inside) {
            const int lobe = L.lobe, off = L.off;
            // A centre at least border_px = ceil(3(2i+3)/2) * step inside the image keeps every box corner inside it in
            // every octave (3*lobe/2 + 1 < border_px), so the border cases of integral_image.h:64-96 cannot occur: the 32
            // look-ups are issued without branches in between (one memory round trip instead of sixteen).
            auto at = [&](int dy, int dx) __attribute__((always_inline)) -> unsigned { return ctr[(long)dy * cols + dx]; };
            auto box = [&](int cx, int cy, int w, int h) __attribute__((always_inline)) -> int {  // centered_rect relative to the centre
                const int l = cx - w / 2, t = cy - h / 2, rr = l + w - 1, b = t + h - 1;
                return (int)(at(b, rr) - at(b, l - 1) - at(t - 1, rr) + at(t - 1, l - 1));
            };
            double Dxx = surf_dxx(box(0, 0, lobe * 3, 2 * lobe - 1), box(0, 0, lobe, 2 * lobe - 1));       // :141-142
            double Dyy = surf_dxx(box(0, 0, 2 * lobe - 1, lobe * 3), box(0, 0, 2 * lobe - 1, lobe));       // :144-145
            double Dxy = (int)((unsigned)box(-off, off, lobe, lobe) + (unsigned)box(off, -off, lobe, lobe) -
                               (unsigned)box(-off, -off, lobe, lobe) - (unsigned)box(off, off, lobe, lobe));  // :147-150
            Dxx *= L.area_inv; Dyy *= L.area_inv; Dxy *= L.area_inv;
            double sign = +1;
            if (Dxx + Dyy < 0) sign = -1;
            double det = Dxx * Dyy - 0.81 * Dxy * Dxy;
            if (det < 0) det = 0;
            IMGFD_OUT_STORE(sign * det, &pyr[L.plane + (size_t)lr * g.nc[o] + lc]);
            hot = det >= thr;
        }
        // one wave = 64 consecutive level pixels of a row = one word of the level's threshold mask
        const unsigned long long word = __ballot(hot);
        if ((threadIdx.x & 63) == 0 && in_level) mask[L.mask + (size_t)lr * ((g.nc[o] + 63) / 64) + (lc >> 6)] = word;
    }
}

// ---- K17, octaves 1-3 from the residue layout with the look-ups as buffer loads.  A level pixel of these octaves sits on a
// column that is a multiple of 4, so the word of look-up (dy, dx) is
//     (r + dy) * cols + ((c + dx) & 3) * per + ((c + dx) >> 2)  =  [r * cols + c / 4]  +  [dy * cols + (dx & 3) * per + (dx >> 2)]:
// a per-lane base plus a constant of the (octave, interval, look-up).  The 12 x 32 constants come from the host
// (SurfTaps, made non-negative by moving the interval's smallest one into the base), land in scalar registers, and each
// look-up is ONE buffer load with the lane's byte offset in a VGPR and the constant in an SGPR (address arithmetic per look-up --
// column residue, quotient, row x pitch in 64 bits -- cost ~7 vector instructions each, 224 per level-pixel value against ~60 for
// the determinant itself: round 3).
struct SurfTaps {
    int w[(SURF_OCT - 1) * (SURF_INT - 2)][32];  // word offsets >= 0 of the 32 look-ups from (lane base + adj), box by box, corner by corner
    int adj[(SURF_OCT - 1) * (SURF_INT - 2)];    // the interval's smallest offset (<= 0)
};
static void surf_make_taps(const SurfGeom &g, SurfTaps *t)
{
    const int cols = g.cols, per = cols >> 2;
    for (int o = 1; o < SURF_OCT; o++)
        for (int it = 1; it < SURF_INT - 1; it++) {
            const SurfLevel &L = g.lev[o * SURF_INT + it];
            const int lobe = L.lobe, off = L.off;
            long w[32];
            int n = 0;
            auto at = [&](int dy, int dx) { w[n++] = (long)dy * cols + (long)(dx & 3) * per + (dx >> 2); };
            auto box = [&](int cx, int cy, int bw, int bh) {  // the corners in the order the kernel's box() reads them
                const int l = cx - bw / 2, tp = cy - bh / 2, rr = l + bw - 1, b = tp + bh - 1;
                at(b, rr); at(b, l - 1); at(tp - 1, rr); at(tp - 1, l - 1);
            };
            box(0, 0, lobe * 3, 2 * lobe - 1); box(0, 0, lobe, 2 * lobe - 1);
            box(0, 0, 2 * lobe - 1, lobe * 3); box(0, 0, 2 * lobe - 1, lobe);
            box(-off, off, lobe, lobe); box(off, -off, lobe, lobe); box(-off, -off, lobe, lobe); box(off, off, lobe, lobe);
            long lo = 0;
            for (int k = 0; k < 32; k++) lo = std::min(lo, w[k]);
            const int e = (o - 1) * (SURF_INT - 2) + it - 1;
            t->adj[e] = (int)lo;
            for (int k = 0; k < 32; k++) t->w[e][k] = (int)(w[k] - lo);
        }
}

__global__ void __launch_bounds__(256) surf_pyramid_taps(const unsigned *__restrict__ J, double *__restrict__ pyr, SurfGeom g, SurfBands bands,
                                                         SurfTaps taps, unsigned long long *__restrict__ mask, double thr)
{
    int o, bx, by;
    if (!surf_band_block(bands, blockIdx.x, o, bx, by)) return;
    const int lc = bx * 64 + (threadIdx.x & 63);
    const int lr = by * 4 + (threadIdx.x >> 6);
    const int step = g.lev[o * SURF_INT].step, cols = g.cols;
    const int r = lr * step, c = lc * step;
    const bool in_level = lr < g.nr[o] && lc < g.nc[o];
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned *>(J), 0, (int)((unsigned)g.rows * (unsigned)cols * 4u), 0x00027000);
    const int base = r * cols + (c >> 2);  // word of (r, c) in J: c is a multiple of 4
#pragma unroll 1
    for (int it = bands.it0[o]; it < SURF_INT - 1; it++) {
        const SurfLevel &L = g.lev[o * SURF_INT + it];
        const int e = (o - 1) * (SURF_INT - 2) + it - 1;
        const bool inside = in_level && !(r < L.border_px || r >= g.rows - L.border_px || c < L.border_px || c >= cols - L.border_px);
        bool hot = false;
        if (inside) {
            const int voff = (base + taps.adj[e]) * 4;
            unsigned v[32];
#pragma unroll
            for (int k = 0; k < 32; k++) v[k] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, taps.w[e][k] * 4, 0);
            auto box = [&](int j) __attribute__((always_inline)) -> int { return (int)(v[4 * j] - v[4 * j + 1] - v[4 * j + 2] + v[4 * j + 3]); };
            double Dxx = surf_dxx(box(0), box(1));                                                        // :141-142
            double Dyy = surf_dxx(box(2), box(3));                                                        // :144-145
            double Dxy = (int)((unsigned)box(4) + (unsigned)box(5) - (unsigned)box(6) - (unsigned)box(7));  // :147-150
            Dxx *= L.area_inv; Dyy *= L.area_inv; Dxy *= L.area_inv;
            double sign = +1;
            if (Dxx + Dyy < 0) sign = -1;
            double det = Dxx * Dyy - 0.81 * Dxy * Dxy;
            if (det < 0) det = 0;
            IMGFD_OUT_STORE(sign * det, &pyr[L.plane + (size_t)lr * g.nc[o] + lc]);
            hot = det >= thr;
        }
        const unsigned long long word = __ballot(hot);
        if ((threadIdx.x & 63) == 0 && in_level) mask[L.mask + (size_t)lr * ((g.nc[o] + 63) / 64) + (lc >> 6)] = word;
    }
}

struct SurfRecord {
    unsigned long long key;  // ((octave*8 + interval) << 40) | row << 20 | column : the reference's emission order
    double x, y, scale, score, laplacian;
};

struct SurfBlocks {
    unsigned first[SURF_OCT + 1];  // first workgroup id of each octave in a launch that covers several (empty ranges allowed)
};
__device__ __forceinline__ int surf_octave_of_block(const SurfBlocks &b, unsigned bid)
{
    int o = 0;
#pragma unroll
    for (int k = 1; k < SURF_OCT; k++) o += bid >= b.first[k] ? 1 : 0;
    return o;
}

struct SurfNmsParams {
    int border_next[SURF_INT];  // get_border_size(i+1) for the interval handled (level coordinates)
    double thr;
    unsigned long long cap;     // records / survivors the buffers hold
    SurfBlocks blocks;          // all octaves in one launch
    // blockIdx.y = tile of a group (imgfd_surf_dev): distances between consecutive tiles' buffers, in elements
    size_t tile_pyr, tile_mask, tile_rec, tile_count, tile_table, tile_surv, tile_cand;
    unsigned long long cand_cap;  // entries of the candidate list (one per built level pixel: a threshold of 0 marks them all)
    // Intervals 0 and 5 of an octave are not in the pyramid buffer (surf_geometry): the maximum test computes the nine values it
    // needs of them from the integral image on the spot -- a few thousand survivors per tile (round 5: a third of all level-pixel
    // values and of the pyramid's HBM writes less).
    SurfTable integral;
};

// ---- K18: get_interest_points (hessian_pyramid.h:453-506) in two kernels.
//
// The pyramid kernels published which level pixels reach the threshold, one bit each -- a fraction of a percent in octave 0,
// a few percent above (measured on the bench tiles): some 4 x 10^5 marked pixels per 4096^2 tile, of which ~10^4 are interest
// points.
//   surf_nms_screen  a workgroup takes up to 256 consecutive mask words of one interval, a thread each, and turns their set bits
//                    into a dense list in LDS (popcount, block scan, one entry per bit: thread << 6 | bit); the threads then share
//                    the listed pixels evenly.  A pixel's own 3x3 first (most marked pixels are not the largest of their own
//                    3x3), then the 3x3 of every neighbouring interval that is in the pyramid buffer, nine loads back to back per
//                    step and nothing kept between the steps: a pixel that no STORED value beats is appended to the survivor list
//                    (its key: octave, interval, row, column).  17 registers of values at a time: eight waves per SIMD.
//   surf_nms_finish  nine threads per survivor, one per position of the 3x3: each fetches its position's value in the three
//                    intervals -- a load where the interval is stored, the 32 table look-ups of build_pyramid (:119-171) where it
//                    is not (interval 0 below i = 1, interval 5 above i = 4: surf_geometry) -- and leaves the absolute values in
//                    LDS; the survivor's first thread finishes is_maximum_in_region (:324-356) against the computed interval and
//                    runs interpolate_point (:411-446) with the closed-form 3x3 inverse (matrix_la.h:922-962), f64, the
//                    reference's operation order (no contraction).
// (Rounds 3-5 did all of this in one kernel that kept a pixel's 27 values and its record in registers across workgroup barriers
// and wave-cooperative rounds: 158 registers + scratch, three waves per SIMD, ~50 us of dependent round trips per workgroup --
// 54 us per tile whatever the number of tiles in flight: profiles/r06/surf_timeline_group8.txt.)
// geometry of dlib's build_pyramid(img, 4, 6, 2) for a per-LANE (octave, interval): arithmetic instead of the SurfGeom tables (a
// kernel argument indexed by a lane-varying value is copied to scratch memory first: surf_nms_finish took 169 us that way)
__device__ __forceinline__ int surf_pick4(const int (&a)[SURF_OCT], int o) { return o == 0 ? a[0] : o == 1 ? a[1] : o == 2 ? a[2] : a[3]; }
struct SurfLaneLevel {
    int nr, nc;
    size_t plane0;  // first built plane (interval 1) of the octave; interval it is (it - 1) * nr * nc further
};
__device__ __forceinline__ SurfLaneLevel surf_lane_level(const SurfGeom &g, int o)
{
    SurfLaneLevel L;
    L.nr = surf_pick4(g.nr, o); L.nc = surf_pick4(g.nc, o);
    size_t p = 0;
#pragma unroll
    for (int k = 0; k < SURF_OCT - 1; k++) p += k < o ? (size_t)(SURF_INT - 2) * g.nr[k] * g.nc[k] : 0;
    L.plane0 = p;
    return L;
}

// All three kernels append to lists in global memory, and a returning atomic on ONE address costs ~9 ns whoever issues it (7 680
// waves of a first cut of surf_nms_list, one atomic each: 69 us for a kernel that moves 6 MB; the round-5 kernel's one atomic per
// workgroup trip was a third of its time): every kernel here gathers what a whole 1024-thread workgroup appends in LDS first and
// takes its places in the list with one atomic.
//
// K18a: mask bits -> keys.  A thread takes one mask word of one interval.
#define NMS_WORDS 1024  /* mask words per workgroup of surf_nms_list */
__host__ __device__ inline unsigned surf_nms_blocks(int nr, int nc)
{
    return (unsigned)(((size_t)nr * ((nc + 63) / 64) + NMS_WORDS - 1) / NMS_WORDS);
}
__global__ void __launch_bounds__(NMS_WORDS) surf_nms_list(SurfGeom g, SurfNmsParams q, unsigned long long *__restrict__ cand,
                                                           unsigned long long *__restrict__ ncand, const unsigned long long *__restrict__ mask)
{
    __shared__ unsigned wave_sum[NMS_WORDS / 64];
    __shared__ unsigned long long base_s;
    {
        const size_t t = blockIdx.y;
        mask += t * q.tile_mask; cand += t * q.tile_cand; ncand += t * q.tile_count;
    }
    const int o = surf_octave_of_block(q.blocks, blockIdx.x), tid = threadIdx.x, lane = tid & 63;  // workgroup-uniform octave and interval
    const int nr = g.nr[o], nc = g.nc[o];
    const int wpr = (nc + 63) / 64;
    const size_t words = (size_t)nr * wpr;
    const unsigned per = surf_nms_blocks(nr, nc), id = blockIdx.x - q.blocks.first[o];
    const int i = (int)(id / per) + 1;
    const size_t w = (size_t)(id % per) * NMS_WORDS + tid;
    unsigned long long word = w < words ? mask[g.lev[o * SURF_INT + i].mask + w] : 0ull;
    const int r = (int)(w / wpr), c0 = (int)(w % wpr) * 64;
    // :474-476 -- a marked pixel closer than border_next + 1 to the level's edge is no candidate
    const int b = q.border_next[i];
    if (r < b + 1 || r >= nr - b - 1) word = 0;
    else {
        const int lo = b + 1 - c0, hi = nc - b - 1 - c0;  // columns [lo, hi) of this word are inside
        if (hi <= 0 || lo >= 64) word = 0;
        else {
            if (lo > 0) word &= ~0ull << lo;
            if (hi < 64) word &= (1ull << hi) - 1ull;
        }
    }
    const unsigned cnt = (unsigned)__popcll(word);
    unsigned incl = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned t = __shfl_up(incl, d);
        if (lane >= d) incl += t;
    }
    if (lane == 63) wave_sum[tid >> 6] = incl;
    __syncthreads();
    unsigned pos = incl - cnt, total = 0;
#pragma unroll
    for (int k = 0; k < NMS_WORDS / 64; k++) {
        if (k < (tid >> 6)) pos += wave_sum[k];
        total += wave_sum[k];
    }
    if (!total) return;  // workgroup-uniform
    if (tid == 0) base_s = atomicAdd(ncand, (unsigned long long)total);
    __syncthreads();
    unsigned long long at = base_s + pos;
    const unsigned long long key0 = ((unsigned long long)(o * 8 + i) << 40) | ((unsigned long long)r << 20) | (unsigned long long)c0;
    while (word) {
        const int bit = __ffsll((long long)word) - 1;
        word &= word - 1;
        if (at < q.cand_cap) cand[at] = key0 + (unsigned long long)bit;
        at++;
    }
}

// K18b: a thread per candidate.  The pixel's own 3x3 first (most marked pixels are not the largest of their own 3x3), then the 3x3
// of every neighbouring interval that is in the pyramid buffer, nine loads back to back per step and nothing kept between the
// steps: a pixel that no STORED value beats is appended to the survivor list.
// the 3x3 around (r, c) of one stored level: true = no value there is strictly larger in magnitude than val
__device__ __forceinline__ bool surf_nms_3x3_ok(const double *__restrict__ P, int nc, double val)
{
    double v[9];
#pragma unroll
    for (int k = 0; k < 9; k++) v[k] = P[(size_t)(k / 3) * nc + (k % 3)];
    bool big = false;
#pragma unroll
    for (int k = 0; k < 9; k++) big |= fabs(v[k]) > val;
    return !big;
}
#define NS_NT 512    /* threads of surf_nms_screen */
#define NS_KEEP 1024 /* survivors a workgroup gathers in LDS before it takes places in the list */
__global__ void __launch_bounds__(NS_NT) surf_nms_screen(const double *__restrict__ pyr, SurfGeom g, SurfNmsParams q,
                                                         const unsigned long long *__restrict__ cand, const unsigned long long *__restrict__ ncand,
                                                         unsigned long long *__restrict__ surv, unsigned long long *__restrict__ nsurv)
{
    __shared__ unsigned long long kept[NS_KEEP];
    __shared__ unsigned nkept;
    __shared__ unsigned long long base_s;
    {
        const size_t t = blockIdx.y;
        pyr += t * q.tile_pyr; cand += t * q.tile_cand; ncand += t * q.tile_count; surv += t * q.tile_surv; nsurv += t * q.tile_count;
    }
    const unsigned long long n_all = *ncand;
    const unsigned long long n = n_all < q.cand_cap ? n_all : q.cand_cap;
    const int tid = threadIdx.x;
    if (tid == 0) nkept = 0;
    __syncthreads();
    auto flush = [&]() {  // every thread of the workgroup
        const unsigned m = nkept;
        if (m) {
            if (tid == 0) base_s = atomicAdd(nsurv, (unsigned long long)m);
            __syncthreads();
            for (unsigned e = tid; e < m; e += NS_NT)
                if (base_s + e < q.cap) surv[base_s + e] = kept[e];
            __syncthreads();
            if (tid == 0) nkept = 0;
        }
        __syncthreads();
    };
    for (unsigned long long k0 = (unsigned long long)blockIdx.x * NS_NT; k0 < n; k0 += (unsigned long long)gridDim.x * NS_NT) {  // workgroup-uniform trips
        const unsigned long long k = k0 + tid;
        if (k < n) {
            const unsigned long long key = cand[k];
            const int o = (int)(key >> 43) & 7, i = (int)(key >> 40) & 7, r = (int)(key >> 20) & 0xfffff, c = (int)(key & 0xfffff);
            const SurfLaneLevel L = surf_lane_level(g, o);
            const size_t lvl = (size_t)L.nr * L.nc;
            const double *own = pyr + L.plane0 + (size_t)(i - 1) * lvl + (size_t)(r - 1) * L.nc + (c - 1);
            const double val = fabs(own[L.nc + 1]);
            // is_maximum_in_region :324-356: rejected by any strictly larger value in the 3x3x3 block; the stored part of it here
            bool keep = val >= q.thr && surf_nms_3x3_ok(own, L.nc, val);
            if (keep && i < SURF_INT - 2) keep = surf_nms_3x3_ok(own + lvl, L.nc, val);  // interval 5 is not built
            if (keep && i > 1) keep = surf_nms_3x3_ok(own - lvl, L.nc, val);             // nor is interval 0
            if (keep) kept[atomicAdd(&nkept, 1u)] = key;  // (at most NS_NT per trip, and a trip starts with at most NS_KEEP - NS_NT)
        }
        __syncthreads();
        if (nkept > NS_KEEP - NS_NT) flush();  // workgroup-uniform
    }
    flush();
}

// one level-pixel value of the Hessian pyramid straight from the integral image (build_pyramid, hessian_pyramid.h:119-171): the
// determinant with the sign of the trace at the level pixel whose centre is image (r, c)
__device__ __forceinline__ double surf_level_value(const SurfTable &T, int lobe, double area_inv, int r, int c)
{
    const int off = lobe / 2 + 1;
    auto box = [&](int cx, int cy, int w, int h) __attribute__((always_inline)) -> unsigned {  // centered_rect relative to the centre
        const int l = c + cx - w / 2, t = r + cy - h / 2, rr = l + w - 1, b = t + h - 1;
        return T.at(b, rr) - T.at(b, l - 1) - T.at(t - 1, rr) + T.at(t - 1, l - 1);
    };
    // wide - 3 narrow in wrap-around arithmetic = the exact integer (surf_dxx)
    double Dxx = (double)(int)(box(0, 0, lobe * 3, 2 * lobe - 1) - 3u * box(0, 0, lobe, 2 * lobe - 1));
    double Dyy = (double)(int)(box(0, 0, 2 * lobe - 1, lobe * 3) - 3u * box(0, 0, 2 * lobe - 1, lobe));
    double Dxy = (double)(int)(box(-off, off, lobe, lobe) + box(off, -off, lobe, lobe) - box(-off, -off, lobe, lobe) - box(off, off, lobe, lobe));
    Dxx *= area_inv; Dyy *= area_inv; Dxy *= area_inv;
    double sign = +1;
    if (Dxx + Dyy < 0) sign = -1;
    double det = Dxx * Dyy - 0.81 * Dxy * Dxy;
    if (det < 0) det = 0;
    return sign * det;
}

#define NF_NT 256
#define NF_SURV (NF_NT / 9)  /* survivors per workgroup trip of surf_nms_finish: 9 threads each */
__global__ void __launch_bounds__(NF_NT) surf_nms_finish(const double *__restrict__ pyr, SurfGeom g, SurfNmsParams q,
                                                         const unsigned long long *__restrict__ surv, const unsigned long long *__restrict__ nsurv,
                                                         SurfRecord *__restrict__ out, unsigned long long *__restrict__ count)
{
    __shared__ double v[NF_SURV][3][9];  // |value| [interval below, own, above][position]
    __shared__ double raw_s[NF_SURV];
    __shared__ unsigned nhit;
    __shared__ unsigned long long base_s;
    {
        const size_t t = blockIdx.y;
        pyr += t * q.tile_pyr; surv += t * q.tile_surv; nsurv += t * q.tile_count; out += t * q.tile_rec; count += t * q.tile_count;
        q.integral.p += t * q.tile_table;
    }
    const int tid = threadIdx.x, s = tid / 9, j = tid - 9 * s;
    const unsigned long long ns_all = *nsurv;
    if (ns_all > q.cap) {  // more survivors than their list holds: report it the way too many records are reported (the caller redoes the tile)
        if (blockIdx.x == 0 && tid == 0) *count = ns_all;
        return;
    }
    const unsigned ns = (unsigned)ns_all;
    for (unsigned base = blockIdx.x * NF_SURV; base < ns; base += gridDim.x * NF_SURV) {  // workgroup-uniform trips
        const bool live = s < NF_SURV && base + s < ns;
        int o = 0, i = 0, r = 0, c = 0;
        if (tid == 0) nhit = 0;
        if (live) {
            const unsigned long long key = surv[base + s];
            o = (int)(key >> 43) & 7; i = (int)(key >> 40) & 7; r = (int)(key >> 20) & 0xfffff; c = (int)(key & 0xfffff);
            const SurfLaneLevel L = surf_lane_level(g, o);
            const int pr = r - 1 + j / 3, pc = c - 1 + j % 3;  // this thread's position of the 3x3
            const size_t at = (size_t)pr * L.nc + pc, lvl = (size_t)L.nr * L.nc;
#pragma unroll
            for (int d = 0; d < 3; d++) {
                const int it = i - 1 + d;
                double w;
                if (it >= 1 && it <= SURF_INT - 2) w = pyr[L.plane0 + (size_t)(it - 1) * lvl + at];
                else {  // hessian_pyramid.h:119-128 for this lane's (octave, interval); positions inside border_next are valid in every interval
                    const int step = 2 << o, lobe = step * (it + 1) + 1;
                    const double tl = 3.0 * lobe;
                    w = surf_level_value(q.integral, lobe, 1.0 / (tl * tl), pr * step, pc * step);
                }
                if (d == 1 && j == 4) raw_s[s] = w;
                v[s][d][j] = fabs(w);
            }
        }
        __syncthreads();
        bool hit = false;
        unsigned slot = 0;
        SurfRecord rec;
        if (live && j == 0) {
            const double val = v[s][1][4], raw = raw_s[s];
            bool ok = true;
            if (i == 1 || i == SURF_INT - 2) {  // the interval that was computed here has not been compared yet
                const int d = i == 1 ? 0 : 2;
#pragma unroll
                for (int k = 0; k < 9; k++) ok &= !(v[s][d][k] > val);
            }
            if (ok) {
                // interpolate_point :411-446 (on the absolute values, as is_maximum_in_region left them)
#define V(d, dy, dx) v[s][d][3 * (1 + (dy)) + 1 + (dx)]
                const double g0 = (V(1, 0, 1) - V(1, 0, -1)) / 2.0;
                const double g1 = (V(1, 1, 0) - V(1, -1, 0)) / 2.0;
                const double g2 = (V(2, 0, 0) - V(0, 0, 0)) / 2.0;
                const double Dxx = (V(1, 0, 1) + V(1, 0, -1)) - 2 * val;
                const double Dyy = (V(1, 1, 0) + V(1, -1, 0)) - 2 * val;
                const double Dss = (V(2, 0, 0) + V(0, 0, 0)) - 2 * val;
                const double Dxy = (V(1, 1, 1) + V(1, -1, -1) - V(1, -1, 1) - V(1, 1, -1)) / 4.0;
                const double Dxs = (V(2, 0, 1) + V(0, 0, -1) - V(0, 0, 1) - V(2, 0, -1)) / 4.0;
                const double Dys = (V(2, 1, 0) + V(0, -1, 0) - V(0, 1, 0) - V(2, -1, 0)) / 4.0;
#undef V
                // inv() of the symmetric 3x3 [a b c; d e f; g h i], matrix_la.h:922-962 with det :1576-1590
                const double ma = Dxx, mb = Dxy, mc = Dxs, md = Dxy, me = Dyy, mf = Dys, mg = Dxs, mh = Dys, mi = Dss;
                double de = ma * (me * mi - mf * mh) - mb * (md * mi - mf * mg) + mc * (md * mh - me * mg);
                double v00 = 1, v01 = 0, v02 = 0, v10 = 0, v11 = 1, v12 = 0, v20 = 0, v21 = 0, v22 = 1;
                if (de != 0) {
                    de = 1.0 / de;
                    v00 = (me * mi - mf * mh) * de; v10 = (mf * mg - md * mi) * de; v20 = (md * mh - me * mg) * de;
                    v01 = (mc * mh - mb * mi) * de; v11 = (ma * mi - mc * mg) * de; v21 = (mb * mg - ma * mh) * de;
                    v02 = (mb * mf - mc * me) * de; v12 = (mc * md - ma * mf) * de; v22 = (ma * me - mb * md) * de;
                }
                const double ix = -(v00 * g0 + v01 * g1 + v02 * g2);
                const double iy = -(v10 * g0 + v11 * g1 + v12 * g2);
                const double iz = -(v20 * g0 + v21 * g1 + v22 * g2);
                if (fmax(fabs(ix), fmax(fabs(iy), fabs(iz))) < 0.5) {
                    rec.key = ((unsigned long long)(o * 8 + i) << 40) | ((unsigned long long)r << 20) | (unsigned long long)c;
                    const double step = (double)(2 << o);  // get_step_size(o) = 2 * 2^o = std::pow(2.0, o + 1.0)
                    rec.x = (c + ix) * step;
                    rec.y = (r + iy) * step;
                    const double lobe = step * (i + iz + 1) + 1;
                    rec.scale = 1.2 / 9.0 * (3 * lobe);
                    rec.score = val;
                    rec.laplacian = raw > 0 ? +1.0 : -1.0;  // get_laplacian :294-297
                    hit = true;
                    slot = atomicAdd(&nhit, 1u);
                }
            }
        }
        __syncthreads();
        if (tid == 0 && nhit) base_s = atomicAdd(count, (unsigned long long)nhit);  // the records of a trip take their places with one atomic
        __syncthreads();
        if (hit && base_s + slot < q.cap) out[base_s + slot] = rec;
        __syncthreads();
    }
}

// ---- ranking on the device (imgfd_surf_dev): get_surf_points, surf.h:268-285, without the host
// One workgroup per tile.  Order of two records: higher score first; equal scores (exact ties of two determinants): the
// point get_interest_points emitted first comes first (what a stable sort would do; the reference's std::sort over
// reverse iterators leaves the order of exact ties to the library's introsort).  As a 128-bit value: (score bits, ~key),
// larger = better (scores are non-negative doubles: their bit patterns order like the values; keys are unique).
//   1. radix select, passes of 8 bits from the top over shrinking candidate lists, until the winners and the candidates still
//      open fit the LDS sort (lim > 2048: until the lim best records are known, at most 16 passes)
//   2. they are ranked among themselves (bitonic sort of up to 2048 composites in LDS; all pairs beyond); the lim best stay
//   3. in rank order: drop the points whose 32*scale box leaves the image (:271-285), compact with a block scan, write
//      x, y, scale for K19 and the head of the feature record (x, y, -, pyramid_scale, score, laplacian)
// counts_out[0] = points kept (or -candidates when the record buffer overflowed: nothing is written then).
#define SR_NT 1024
#define SR_SORT 2048  /* selected records the LDS sort of surf_rank_select holds; more: all-pairs ranking */
struct SurfRankParams {
    const SurfRecord *rec;
    const unsigned long long *count;
    unsigned long long cap;
    unsigned lim;            // min(max_points, rows of the feature buffer)
    int rows, cols;
    unsigned *sel, *order;   // scratch, lim entries each
    unsigned *cand;          // scratch, 2 x cap entries (the candidate lists of the radix select)
    unsigned sort_cap;       // selected records the LDS sort takes (SR_SORT; tests lower it to reach the all-pairs ranking)
    double *pts;             // lim x 3 for K19
    double *feat;            // lim x 70 feature records of this tile
    long long *count_out;    // d_counts[f]
    unsigned *m_out;         // the same number for the K19 kernels
    // blockIdx.x = tile of a group: distances between consecutive tiles' buffers, in elements (count_out, m_out: 1)
    size_t tile_rec, tile_count, tile_sel, tile_cand, tile_pts, tile_feat;
};

__device__ __forceinline__ unsigned long long surf_score_bits(double v)
{
    unsigned long long b;
    memcpy(&b, &v, sizeof b);
    return b;
}
__device__ __forceinline__ unsigned surf_comp_byte(const SurfRecord &r, int pass)  // pass 0 = most significant byte
{
    const unsigned long long hi = surf_score_bits(r.score), lo = ~r.key;
    return pass < 8 ? (unsigned)(hi >> (8 * (7 - pass))) & 0xffu : (unsigned)(lo >> (8 * (15 - pass))) & 0xffu;
}
// a > b in the ranking order
__device__ __forceinline__ bool surf_better(unsigned long long sa, unsigned long long ka, unsigned long long sb, unsigned long long kb)
{
    return sa > sb || (sa == sb && ka < kb);
}

// LDS written by some lanes of a wave, read by others: program order inside the wave is the synchronisation; the fences keep
// the compiler from moving a read over another lane's write
__device__ __forceinline__ void sr_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__global__ void __launch_bounds__(SR_NT) surf_rank_select(SurfRankParams q)
{
    __shared__ unsigned hist[256];
    __shared__ unsigned want, nsel, ncand, cut, scan[SR_NT / 64], wtot[4], carry;
    __shared__ unsigned long long st_s[SR_SORT], st_k[SR_SORT];
    __shared__ unsigned sidx[SR_SORT];
    const int tid = threadIdx.x;
    {
        const size_t t = blockIdx.x;
        q.rec += t * q.tile_rec; q.count += t * q.tile_count; q.sel += t * q.tile_sel; q.order += t * q.tile_sel; q.cand += t * q.tile_cand;
        q.pts += t * q.tile_pts; q.feat += t * q.tile_feat; q.count_out += t; q.m_out += t;
    }
    const unsigned long long cnt = *q.count;
    if (cnt > q.cap) {  // more candidates than the record buffer holds: report, leave the feature rows alone
        if (tid == 0) { *q.count_out = -(long long)cnt; *q.m_out = 0; }
        return;
    }
    const unsigned n = (unsigned)cnt;
    const unsigned lim = min(q.lim, n);
    if (lim == 0) {
        if (tid == 0) { *q.count_out = 0; *q.m_out = 0; }
        return;
    }
    // ---- 1 + 2. radix select with shrinking candidate lists: a pass builds the histogram of one composite byte over the
    // current candidates, finds the bin in which the `want`-th best of them lies, and parts them: candidates in higher bins
    // are selected, those in that bin are the next pass's candidates (the first passes -- sign and exponent bytes, equal for
    // nearly all records -- keep everything; from the first mantissa byte on a pass keeps ~1/256).  Every record is read
    // twice in pass 0 and rarely again (round 2 scanned all n records in each of 8-16 passes: 205 us per tile).
    if (tid == 0) { want = lim; nsel = 0; ncand = 0; }
    __syncthreads();
    unsigned mc = n;                      // candidates of the current pass (list `cur`; pass 0: all records)
    unsigned *cur = q.cand, *nxt = q.cand + q.cap;
    bool ident = true;                    // cur is the identity
    if (n <= lim) {
        for (unsigned i = tid; i < n; i += SR_NT) q.sel[i] = i;
        if (tid == 0) nsel = n;
        mc = 0;
    }
    const unsigned sort_limit = min((unsigned)SR_SORT, q.sort_cap);
    for (int pass = 0; pass < 16 && mc > 0; pass++) {
        // Winners so far + the candidates still open fit the LDS sort: it settles the rest (the later passes -- few candidates,
        // two dependent loads and seven barriers each -- were 8 x ~4 us for one workgroup)
        if (nsel + mc <= sort_limit) break;
        for (int i = tid; i < 256; i += SR_NT) hist[i] = 0;
        if (tid == 0) cut = 0;  // (want <= mc always: some bin takes it)
        __syncthreads();
        for (unsigned i0 = 0; i0 < mc; i0 += SR_NT) {  // every lane makes every trip (wave intrinsics inside)
            const unsigned i = i0 + tid;
            int b = -1;
            if (i < mc) b = (int)surf_comp_byte(q.rec[ident ? i : cur[i]], pass);
            // histogram increment.  The sign / exponent bytes are the same for nearly every record, and thousands of atomics on
            // one LDS word serialise: a wave whose live lanes all hold one value adds their number once.  Diverse bytes (the
            // mantissa passes) go to their bins lane by lane (aggregating value by value took up to 64 rounds per trip there:
            // most of the 200 us this kernel needed in round 2).
            const unsigned long long livem = __ballot(b >= 0);
            if (livem) {
                const int lead = __shfl(b, __ffsll((long long)livem) - 1);
                if (__all(b < 0 || b == lead)) {
                    if (((int)__lane_id()) == __ffsll((long long)livem) - 1) atomicAdd(&hist[lead], (unsigned)__popcll(livem));
                } else if (b >= 0) {
                    atomicAdd(&hist[b], 1u);
                }
            }
        }
        __syncthreads();
        // the bin in which the `want`-th best candidate lies: suffix sums of the histogram from the top, four waves (one thread
        // walking the bins -- up to 256 dependent LDS reads in each of the 8-16 passes -- was half of this kernel's 64 us)
        const unsigned want_in = want;  // read by every thread before the barrier, rewritten behind it
        unsigned hb = 0, incl = 0;
        if (tid < 256) {
            hb = hist[255 - tid];
            incl = si_wave_incl(hb, tid & 63);
            if ((tid & 63) == 63) wtot[tid >> 6] = incl;
        }
        __syncthreads();
        if (tid < 256) {
            for (int w = 0; w < (tid >> 6); w++) incl += wtot[w];
            if (incl >= want_in && incl - hb < want_in) { cut = 255u - (unsigned)tid; want = want_in - (incl - hb); }
        }
        if (tid == 0) ncand = 0;
        __syncthreads();
        const unsigned cb = cut, keep_all = hist[cb] == want ? 1u : 0u;  // the whole bin is needed: no further pass
        if (!keep_all && hist[cb] == mc) {  // every candidate holds this byte (sign / exponent): nothing to part
            __syncthreads();
            continue;
        }
        for (unsigned i0 = 0; i0 < mc; i0 += SR_NT) {
            const unsigned i = i0 + tid;
            unsigned idx = 0, b = 0;
            const bool live = i < mc;
            if (live) { idx = ident ? i : cur[i]; b = surf_comp_byte(q.rec[idx], pass); }
            const bool win = live && (b > cb || (keep_all && b == cb)), stay = live && !keep_all && b == cb;
            const unsigned long long mw = __ballot(win), ms = __ballot(stay);
            const int lane = (int)__lane_id();
            unsigned bw = 0, bs = 0;
            if (lane == 0) { if (mw) bw = atomicAdd(&nsel, (unsigned)__popcll(mw)); if (ms) bs = atomicAdd(&ncand, (unsigned)__popcll(ms)); }
            bw = __shfl(bw, 0); bs = __shfl(bs, 0);
            const unsigned long long below = (1ull << lane) - 1ull;
            if (win) q.sel[bw + (unsigned)__popcll(mw & below)] = idx;
            if (stay) nxt[bs + (unsigned)__popcll(ms & below)] = idx;
        }
        __syncthreads();
        mc = keep_all ? 0u : ncand;
        unsigned *t = cur; cur = nxt; nxt = t;
        ident = false;
        __syncthreads();
    }
    __syncthreads();
    // the records to rank: the winners (every one of them better than every open candidate) and, if the select stopped early,
    // the open candidates; the best `m` of them in rank order are the result
    const unsigned ns = nsel, m_all = ns + mc, m = min(m_all, lim);
    if (m_all <= sort_limit) {
        // up to SR_SORT records: bitonic sort of (score bits, key, record index) in LDS, better first (all pairs -- below -- cost
        // 10^6 comparisons in this one workgroup for the R default of 1000 points: ~80 us of the kernel's 140)
        unsigned N = 2;
        while (N < m_all) N <<= 1;
        for (unsigned i = tid; i < N; i += SR_NT) {
            if (i < m_all) {
                const unsigned idx = i < ns ? q.sel[i] : (ident ? i - ns : cur[i - ns]);
                const SurfRecord &r = q.rec[idx];
                sidx[i] = idx; st_s[i] = surf_score_bits(r.score); st_k[i] = r.key;
            } else { sidx[i] = 0u; st_s[i] = 0ull; st_k[i] = ~0ull; }  // padding ranks after every record
        }
        __syncthreads();
        // A thread's pair (lo, hi) lies in the block of 2j elements around it: for j <= 64 the 64 threads of a wave work on 128
        // elements no other wave touches in that step -- and in every later step of the same merge -- so those steps need
        // no workgroup barrier (program order inside the wave + a wave-scope fence); 10 barriers instead of 55 for 1024 records
        for (unsigned k = 2; k <= N; k <<= 1)
            for (unsigned j = k >> 1; j > 0; j >>= 1) {
                for (unsigned t = tid; t < N / 2; t += SR_NT) {
                    const unsigned lo = ((t & ~(j - 1)) << 1) | (t & (j - 1)), hi = lo | j;
                    const unsigned long long sa = st_s[lo], sb = st_s[hi], ka = st_k[lo], kb = st_k[hi];
                    if (surf_better(sa, ka, sb, kb) != ((lo & k) == 0)) {  // the composites themselves move (sorting positions
                        st_s[lo] = sb; st_s[hi] = sa;                      // instead put two dependent, bank-conflicting reads
                        st_k[lo] = kb; st_k[hi] = ka;                      // into every step)
                        const unsigned ia = sidx[lo];
                        sidx[lo] = sidx[hi]; sidx[hi] = ia;
                    }
                }
                if (j > 64 || (j == 1 && k >= 128) || N / 2 > SR_NT) __syncthreads();  // (the merge after k = 128 starts with other waves' elements)
                else sr_wave_sync();
            }
        __syncthreads();
        for (unsigned i = tid; i < m; i += SR_NT) q.order[i] = sidx[i];
    } else
    for (unsigned base = 0; base < m; base += SR_NT) {
        const unsigned i = base + tid;
        unsigned long long si = 0, ki = 0;
        if (i < m) { const SurfRecord &r = q.rec[q.sel[i]]; si = surf_score_bits(r.score); ki = r.key; }
        unsigned rank = 0;
        for (unsigned jb = 0; jb < m; jb += SR_NT) {
            __syncthreads();
            if (jb + tid < m) { const SurfRecord &r = q.rec[q.sel[jb + tid]]; st_s[tid] = surf_score_bits(r.score); st_k[tid] = r.key; }
            __syncthreads();
            const unsigned jn = min((unsigned)SR_NT, m - jb);
            if (i < m)
                for (unsigned j = 0; j < jn; j++) rank += surf_better(st_s[j], st_k[j], si, ki) ? 1u : 0u;
        }
        if (i < m) q.order[rank] = q.sel[i];
    }
    __syncthreads();
    // ---- 3. box test, compaction in rank order, outputs
    if (tid == 0) carry = 0;
    __syncthreads();
    for (unsigned base = 0; base < m; base += SR_NT) {
        const unsigned k = base + tid;
        bool keep = false;
        SurfRecord r;
        if (k < m) {
            r = q.rec[q.order[k]];
            const unsigned long bs = (unsigned long)(32.0 * r.scale);
            const long px = (long)floor(r.x + 0.5), py = (long)floor(r.y + 0.5);
            const long l = px - (long)bs / 2, t = py - (long)bs / 2, rr = l + (long)bs - 1, b = t + (long)bs - 1;
            keep = l >= 0 && t >= 0 && rr <= q.cols - 1 && b <= q.rows - 1;
        }
        // positions: ballot inside the wave, the 16 wave totals through LDS (a Hillis-Steele scan over 1024 words took 20 barriers)
        const unsigned long long km = __ballot(keep);
        const int lane = (int)__lane_id(), wave = tid >> 6;
        if (lane == 0) scan[wave] = (unsigned)__popcll(km);
        __syncthreads();
        unsigned before = carry, total = 0;
        for (int w = 0; w < SR_NT / 64; w++) {
            const unsigned c = scan[w];
            if (w < wave) before += c;
            total += c;
        }
        if (keep) {
            const unsigned pos = before + (unsigned)__popcll(km & ((1ull << lane) - 1ull));
            q.pts[3 * pos] = r.x; q.pts[3 * pos + 1] = r.y; q.pts[3 * pos + 2] = r.scale;
            double *h = q.feat + (size_t)pos * 70;
            h[0] = r.x; h[1] = r.y; h[2] = 0.0; h[3] = r.scale; h[4] = r.score; h[5] = r.laplacian;
        }
        __syncthreads();
        if (tid == 0) carry += total;
        __syncthreads();
    }
    if (tid == 0) { *q.count_out = (long long)carry; *q.m_out = carry; }
}

// surf_describe.hip
// m_dev != nullptr: m is an upper bound (the grid); the number of points is read from *m_dev on the device
// grp != nullptr: one launch for a group of tiles (blockIdx.y), their buffers SurfGroup's strides apart
imgfd_status launch_surf_orient(imgfd_ctx *ctx, const SurfTable &T, const double *d_pts, int m,
                                double *d_samples, double *d_trig, const unsigned *m_dev = nullptr, const SurfGroup *grp = nullptr);
imgfd_status launch_surf_desc(imgfd_ctx *ctx, const SurfTable &T, const double *d_pts,
                              const double *d_trig, int m, double *d_des, int des_stride, double *d_angle,
                              const unsigned *m_dev = nullptr, const SurfGroup *grp = nullptr);

namespace {

struct SurfDevice {
    unsigned *integral = nullptr;         // the table, in the layout launch_surf_integral chooses (SurfTable)
    unsigned long long *mask = nullptr;   // threshold masks of the built pyramid levels (SurfGeom::mask_words words)
    double *pyr = nullptr;
    size_t pyr_bytes = 0;
    SurfRecord *rec = nullptr;
    unsigned long long *count = nullptr;  // records found
    unsigned long long *cands = nullptr;  // keys of the marked level pixels (surf_nms_list -> surf_nms_screen), cand_cap entries
    unsigned long long cand_cap = 0;      //   = the built level pixels of the pyramid: a threshold of 0 marks them all
    unsigned long long *surv = nullptr;   // keys of the level pixels no stored value beats (surf_nms_screen -> surf_nms_finish), cap entries
    // counters: count[0] records found, count[1] survivors, count[2] candidates (one 256-byte slot)
    unsigned long long cap = 0;
};

// scratch the band / strip form of the integral image needs (bytes)
size_t surf_integral_scratch(int rows, int cols)
{
    const size_t nb = ceil_div(rows, SI_RB), ns = ceil_div(cols, 256);
    return sizeof(unsigned) * (nb * (size_t)cols + (size_t)rows * ns + nb * ns) + 512;
}
// the pyramid buffer is idle until the integral image is complete: it lends the scans their scratch
size_t surf_pyr_bytes(const SurfGeom &g, size_t pyr_total)
{
    return std::max(std::max<size_t>(8 * pyr_total, 8), std::max(surf_integral_scratch(g.rows, g.cols), sizeof(unsigned) * 32 * (size_t)g.cols));
}

size_t surf_ws_bytes(const SurfGeom &g, size_t pyr_total, unsigned long long cap)
{
    const size_t n = (size_t)g.rows * g.cols;
    return align_up(3 * n, 256) + align_up(4 * n, 256) + align_up(surf_pyr_bytes(g, pyr_total), 256) + align_up(8 * std::max<size_t>(g.mask_words, 1), 256) +
           align_up(sizeof(SurfRecord) * cap, 256) + align_up(sizeof(unsigned long long) * cap, 256) + align_up(8 * std::max<size_t>(pyr_total, 1), 256) + 4096;
}

// K16 on the context's stream.  allow_residue: the band / strip form may write the table in the residue layout (SurfTable; the
// return value's `per` says whether it did): images whose width is a multiple of 16 and whose table stays below 2 GiB (the gather
// kernel of octaves 1-3 addresses it with 32-bit byte offsets).
// zero32 (optional): a 256-byte counter slot to clear on the way
SurfTable launch_surf_integral(imgfd_ctx *ctx, const uint8_t *d_rgb, unsigned *d_I, int rows, int cols, void *scratch, size_t scratch_bytes,
                               bool allow_residue, unsigned long long *zero32 = nullptr)
{
    SurfTable T{d_I, rows, cols, 0};
    const bool vec = cols % 4 == 0 && (size_t)d_rgb % 4 == 0 && (size_t)d_I % 16 == 0;
    if (vec && scratch && (size_t)scratch % 16 == 0 && scratch_bytes >= surf_integral_scratch(rows, cols) && (size_t)rows * cols >= 65536 &&
        ceil_div(cols, 256) <= SI_MAX_STRIPS) {
        const int nb = ceil_div(rows, SI_RB), ns = ceil_div(cols, 256);
        unsigned *colsum = (unsigned *)scratch;          // nb x cols (16-byte aligned rows: cols % 4 == 0)
        unsigned *rowsum = colsum + (size_t)nb * cols;   // rows x ns
        unsigned *bandstrip = rowsum + (size_t)rows * ns;  // nb x ns
        const dim3 grid(ceil_div(ns, 4), nb);
        hipLaunchKernelGGL(surf_int_sums, grid, dim3(256), 0, ctx->stream, d_rgb, colsum, rowsum, bandstrip, rows, cols, ns);
        hipLaunchKernelGGL(surf_int_carry, dim3(ceil_div(cols, 64) + 1 + ceil_div(rows, 256)), dim3(256), 0, ctx->stream, colsum, bandstrip, rowsum, nb, rows,
                           cols, ns, zero32);
        const bool res = allow_residue && cols % 16 == 0 && (size_t)rows * cols * 4 < ((size_t)1 << 31);
        if (res)
            hipLaunchKernelGGL(surf_int_apply<true>, grid, dim3(256), 0, ctx->stream, d_rgb, (const unsigned *)colsum, (const unsigned *)bandstrip,
                               (const unsigned *)rowsum, d_I, rows, cols, ns);
        else
            hipLaunchKernelGGL(surf_int_apply<false>, grid, dim3(256), 0, ctx->stream, d_rgb, (const unsigned *)colsum, (const unsigned *)bandstrip,
                               (const unsigned *)rowsum, d_I, rows, cols, ns);
        T.per = res ? cols >> 2 : 0;
        return T;
    }
    if (zero32 && hipMemsetAsync(zero32, 0, 256, ctx->stream) != hipSuccess) T.p = nullptr;  // (the caller checks hipGetLastError)
    if (vec)
        hipLaunchKernelGGL(surf_gray_rowscan4, dim3(rows), dim3(256), 0, ctx->stream, d_rgb, d_I, cols);
    else
        hipLaunchKernelGGL(surf_gray_rowscan, dim3(rows), dim3(256), 0, ctx->stream, d_rgb, d_I, cols);
    const int nseg = rows >= 1024 ? 32 : (rows >= 256 ? 8 : 1);
    if (nseg > 1 && scratch && scratch_bytes >= sizeof(unsigned) * (size_t)nseg * cols) {
        const int seg_rows = ceil_div(rows, nseg);
        dim3 grid(ceil_div(cols, 256), ceil_div(rows, seg_rows));
        hipLaunchKernelGGL(surf_colscan_sums, grid, dim3(256), 0, ctx->stream, d_I, (unsigned *)scratch, rows, cols, seg_rows);
        hipLaunchKernelGGL(surf_colscan_apply, grid, dim3(256), 0, ctx->stream, d_I, (const unsigned *)scratch, rows, cols, seg_rows);
    } else {
        hipLaunchKernelGGL(surf_colscan, dim3(ceil_div(cols, 64)), dim3(64), 0, ctx->stream, d_I, rows, cols);
    }
    return T;
}

// Launches of a scope go to stream `s` instead of the context's own (the launch helpers of this library take the stream from the
// context; a context is used by one host thread at a time)
struct SurfStreamScope {
    imgfd_ctx *c;
    hipStream_t old;
    SurfStreamScope(imgfd_ctx *ctx, hipStream_t s) : c(ctx), old(ctx->stream) { ctx->stream = s; }
    ~SurfStreamScope() { c->stream = old; }
};

// FRONT of a tile: K16 + K17, the kernels that fill the chip (integral image, Hessian pyramid + threshold masks), on the context's
// stream; says in *table where and how the integral image lies.
// fork (optional): the context's companion.  Octave 0 and octaves 1-3 are two kernels that both read the finished table and
// write different planes: with a companion at hand (a call with ONE tile has no other tile to fill the chip with) the second
// runs on its stream beside the first -- a VALU / LDS bound kernel next to one that waits for its gathers -- and whatever is
// queued on the context's stream next waits for both.
imgfd_status surf_front(imgfd_ctx *ctx, const uint8_t *d_rgb, const SurfGeom &g, double thr, const SurfDevice &d, SurfTable *table,
                        imgfd_ctx *fork = nullptr)
{
    const SurfTable T = launch_surf_integral(ctx, d_rgb, d.integral, g.rows, g.cols, d.pyr, d.pyr_bytes, true, d.count);
    if (!T.p) return imgfd_fail(ctx, IMGFD_ERR_HIP, "hipMemsetAsync failed (SURF counters)");
    if (table) *table = T;
    static_assert(SURF_INT == 6 && SURF_OCT == 4, "dlib's build_pyramid(img, 4, 6, 2): the kernels unroll its geometry");
    SurfBands bands = surf_bands(g);
    // intervals 1 and 2 of octave 1 come out of the first octave's windows where its blocks cover the octave (always, for dlib's geometry)
    using G0 = SurfPyrLds<0>;
    const bool next_too = g.nr[0] >= 1 && g.nc[0] >= 1 && bands.gx[1] > 0 && ceil_div(g.nr[1], G0::LY / 2) <= ceil_div(g.nr[0], G0::LY) &&
                          ceil_div(g.nc[1], G0::LX / 2) <= ceil_div(g.nc[0], G0::LX);
    if (next_too) bands.it0[1] = 3;
    hipStream_t upper = ctx->stream;  // the stream of the gather kernel (octaves 1-3)
    if (fork && bands.total) {
        IMGFD_HIP(ctx, hipEventRecord(ctx->ev_gate, ctx->stream));
        IMGFD_HIP(ctx, hipStreamWaitEvent(fork->stream, ctx->ev_gate, 0));
        upper = fork->stream;
    }
    // octave 0 (three quarters of all level pixels): the table window of a block of level pixels in LDS
    if (g.nr[0] >= 1 && g.nc[0] >= 1) IMGFD_TRY(launch_surf_pyramid_lds<0>(ctx, T, d.pyr, g, d.mask, thr, next_too));
    // (ALL of octave 1 from LDS needs a window that reaches 32 pixels instead of 20: +25 KB per block, two workgroups per CU -- 0.204 ->
    // 0.217 ms per tile before the extra arithmetic, `profiles/r06/surf_first_octave_phases.txt`; a window kernel of its own for
    // octave 1 measured 219 us against 164 us for the gather kernel in round 3)
    if (bands.total) {
        if (T.per) {
            SurfTaps taps;
            surf_make_taps(g, &taps);
            hipLaunchKernelGGL(surf_pyramid_taps, dim3(bands.total), dim3(256), 0, upper, (const unsigned *)d.integral, d.pyr, g, bands, taps, d.mask, thr);
        } else {
            hipLaunchKernelGGL(surf_pyramid_plain, dim3(bands.total), dim3(256), 0, upper, (const unsigned *)d.integral, d.pyr, g, bands, d.mask, thr);
        }
    }
    if (upper != ctx->stream) {
        IMGFD_HIP(ctx, hipEventRecord(ctx->ev_gate2, upper));
        IMGFD_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_gate2, 0));
    }
    IMGFD_HIP(ctx, hipGetLastError());
    return IMGFD_OK;
}

// BACK, first part: K18 (maximum test + interpolation from the masks) on the context's stream, ONE launch for the `tiles` tiles of a
// group (d = the first tile's buffers, d_next = the second's: the distances between consecutive sets); leaves the records
// (unordered) + counts on the device.  Latency-bound (a chain of dependent round trips per workgroup, issue 0.09), like the
// ranking and K19 behind it.
imgfd_status surf_back_nms(imgfd_ctx *ctx, const SurfGeom &g, double thr, const SurfDevice &d, const SurfTable &T, int tiles = 1,
                           const SurfDevice *d_next = nullptr)
{
    SurfNmsParams q;
    q.thr = thr; q.cap = d.cap; q.cand_cap = d.cand_cap;
    q.integral = T;
    q.tile_pyr = q.tile_mask = q.tile_rec = q.tile_count = q.tile_table = q.tile_surv = q.tile_cand = 0;
    if (tiles > 1 && d_next) {
        q.tile_pyr = (size_t)(d_next->pyr - d.pyr); q.tile_mask = (size_t)(d_next->mask - d.mask); q.tile_rec = (size_t)(d_next->rec - d.rec);
        q.tile_count = (size_t)(d_next->count - d.count); q.tile_table = (size_t)(d_next->integral - d.integral);
        q.tile_surv = (size_t)(d_next->surv - d.surv); q.tile_cand = (size_t)(d_next->cands - d.cands);
    }
    // (the counters -- [0] records, [1] survivors, [2] candidates, a 256-byte slot per tile -- were cleared by the tile's front)
    for (int i = 0; i < SURF_INT; i++) q.border_next[i] = (int)surf_border_of(std::min(i + 1, SURF_INT - 1));
    unsigned nb = 0;
    for (int o = 0; o < SURF_OCT; o++) {
        q.blocks.first[o] = nb;
        if (g.nr[o] >= 1 && g.nc[o] >= 1) nb += surf_nms_blocks(g.nr[o], g.nc[o]) * (SURF_INT - 2);
    }
    q.blocks.first[SURF_OCT] = nb;
    if (nb) {
        unsigned long long *ncount = d.count;  // [0] records, [1] survivors, [2] candidates
        hipLaunchKernelGGL(surf_nms_list, dim3(nb, (unsigned)tiles), dim3(NMS_WORDS), 0, ctx->stream, g, q, d.cands, ncount + 2, (const unsigned long long *)d.mask);
        // grids that cover full lists in a few trips per workgroup; the numbers of candidates and survivors are read on the device
        const unsigned scr = (unsigned)std::max<unsigned long long>(1, std::min<unsigned long long>(512, (d.cand_cap + NS_NT - 1) / NS_NT));
        hipLaunchKernelGGL(surf_nms_screen, dim3(scr, (unsigned)tiles), dim3(NS_NT), 0, ctx->stream, (const double *)d.pyr, g, q, (const unsigned long long *)d.cands,
                           (const unsigned long long *)(ncount + 2), d.surv, ncount + 1);
        const unsigned fin = (unsigned)std::max<unsigned long long>(1, std::min<unsigned long long>(1024, (d.cap + NF_SURV - 1) / NF_SURV));
        hipLaunchKernelGGL(surf_nms_finish, dim3(fin, (unsigned)tiles), dim3(NF_NT), 0, ctx->stream, (const double *)d.pyr, g, q, (const unsigned long long *)d.surv,
                           (const unsigned long long *)(ncount + 1), d.rec, ncount);
    }
    IMGFD_HIP(ctx, hipGetLastError());
    return IMGFD_OK;
}

// K16-K18 for one image already in device memory, one after the other on the context's stream
imgfd_status surf_device_stages(imgfd_ctx *ctx, const uint8_t *d_rgb, const SurfGeom &g, double thr, const SurfDevice &d, SurfTable *table,
                                imgfd_ctx *fork = nullptr)
{
    SurfTable T;
    IMGFD_TRY(surf_front(ctx, d_rgb, g, thr, d, &T, fork));
    if (table) *table = T;
    return surf_back_nms(ctx, g, thr, d, T);
}

// uploads the image, runs the device stages, returns the interest points as the device left them (pts) and the order in which
// the reference emits them (emission: indices into pts, ascending key -- 16-byte (key, index) pairs are sorted, not the 48-byte
// records: 0.28 -> 0.1 ms for the 10 716 records of a bench tile)
// (and, if asked, where the integral image sits in the workspace: valid until the next call carves the arena)
void surf_emission_order(const std::vector<SurfRecord> &pts, std::vector<unsigned> &emission)
{
    std::vector<std::pair<unsigned long long, unsigned>> order(pts.size());
    for (size_t k = 0; k < pts.size(); k++) order[k] = {pts[k].key, (unsigned)k};
    std::sort(order.begin(), order.end());  // keys are unique
    emission.resize(pts.size());
    for (size_t k = 0; k < pts.size(); k++) emission[k] = order[k].second;
}
// emission == nullptr: the caller will ask for the order only if it needs it (surf_emission_order)
imgfd_status surf_points_host(imgfd_ctx *ctx, const void *rgb, int kind, int rows, int cols, double thr,
                              std::vector<SurfRecord> &pts, std::vector<unsigned> *emission, SurfTable *d_integral)
{
    pts.clear();
    if (emission) emission->clear();
    if (rows < 1 || cols < 1) return IMGFD_OK;
    IMGFD_HIP(ctx, hipSetDevice(ctx->device));
    SurfGeom g;
    const size_t total = surf_geometry(rows, cols, &g);
    const size_t n = (size_t)rows * cols;
    SurfDevice d;
    d.cap = 1ull << 16;
    for (;;) {
        IMGFD_TRY(ws_reserve(ctx, surf_ws_bytes(g, total, d.cap) + upload_stage_bytes(kind, 3 * n)));
        ctx->ws_used = 0;
        uint8_t *d_rgb = (uint8_t *)ws_alloc(ctx, 3 * n);
        d.integral = (unsigned *)ws_alloc(ctx, 4 * n);
        d.mask = (unsigned long long *)ws_alloc(ctx, 8 * std::max<size_t>(g.mask_words, 1));
        d.pyr_bytes = surf_pyr_bytes(g, total);
        d.pyr = (double *)ws_alloc(ctx, d.pyr_bytes);
        d.rec = (SurfRecord *)ws_alloc(ctx, sizeof(SurfRecord) * d.cap);
        d.surv = (unsigned long long *)ws_alloc(ctx, sizeof(unsigned long long) * d.cap);
        d.cand_cap = std::max<size_t>(total, 1);
        d.cands = (unsigned long long *)ws_alloc(ctx, 8 * (size_t)d.cand_cap);
        d.count = (unsigned long long *)ws_alloc(ctx, 256);
        if (!d_rgb || !d.integral || !d.mask || !d.pyr || !d.rec || !d.surv || !d.cands || !d.count) return imgfd_fail(ctx, IMGFD_ERR_OOM, "workspace reservation too small");
        IMGFD_TRY(upload_image(ctx, rgb, kind, 3 * n, d_rgb));
        imgfd_ctx *fork = nullptr;
        IMGFD_TRY(ctx_side(ctx, &fork));
        SurfTable T;
        IMGFD_TRY(surf_device_stages(ctx, d_rgb, g, thr, d, &T, fork));
        // the count and, speculatively, the first 16 384 records come back together into pinned memory: one wait instead of two
        // (a 4096^2 tile at the R default threshold has ~10^4 records)
        const size_t spec = (size_t)std::min<unsigned long long>(d.cap, 16384);
        IMGFD_TRY(pin_reserve(ctx, 256 + sizeof(SurfRecord) * spec));
        unsigned long long *h_cnt = reinterpret_cast<unsigned long long *>(ctx->pin);
        SurfRecord *h_rec = reinterpret_cast<SurfRecord *>(ctx->pin + 256);
        IMGFD_HIP(ctx, hipMemcpyAsync(h_cnt, d.count, sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
        IMGFD_HIP(ctx, hipMemcpyAsync(h_rec, d.rec, sizeof(SurfRecord) * spec, hipMemcpyDeviceToHost, ctx->stream));
        IMGFD_HIP(ctx, hipStreamSynchronize(ctx->stream));
        const unsigned long long cnt = *h_cnt;
        if (cnt > d.cap) { d.cap = cnt + 1024; continue; }  // rare: more candidates than the record buffer holds
        pts.resize((size_t)cnt);
        if (cnt) memcpy(pts.data(), h_rec, sizeof(SurfRecord) * std::min<size_t>((size_t)cnt, spec));
        if (cnt > spec) IMGFD_HIP(ctx, hipMemcpyAsync(pts.data() + spec, d.rec + spec, sizeof(SurfRecord) * (cnt - spec), hipMemcpyDeviceToHost, ctx->stream));
        if (d_integral) *d_integral = T;
        IMGFD_HIP(ctx, hipStreamSynchronize(ctx->stream));
        break;
    }
    if (emission) surf_emission_order(pts, *emission);
    return IMGFD_OK;
}

// interest_point::operator<, hessian_pyramid.h:31, on (score, index into pts)
struct SurfScored {
    double score;
    unsigned idx;
};
struct ScoreLess {
    bool operator()(const SurfScored &a, const SurfScored &b) const { return a.score < b.score; }
};

// get_surf_points, surf.h:268-285: strongest first (std::sort over reverse iterators, as the reference calls it), at
// most max_points, and only points whose 32*scale box lies inside the image.  keep: indices into pts, in the reference's order.
// The reference sorts its 40-byte interest_points; where std::sort leaves exact score ties depends on the comparisons and on the
// sequence it starts from -- the emission order -- not on what else an element carries: 16-byte (score, index) pairs in emission
// order go through the same std::sort (0.5 -> 0.15 ms for 10 716 records).
// Fast path first: where the `lim` strongest scores are pairwise different and none of them equals a score behind the cut, their
// order is the same whatever sorts them -- a partition (nth_element) and a sort of `lim` records say so in ~0.1 ms; only
// otherwise (exact ties: synthetic images) the whole sequence goes through the reference's std::sort in emission order.
void surf_select(const std::vector<SurfRecord> &pts, long max_points, int rows, int cols, std::vector<size_t> &keep)
{
    const size_t n = pts.size(), lim = std::min((size_t)max_points, n);
    std::vector<SurfScored> sc(n);
    for (size_t k = 0; k < n; k++) sc[k] = {pts[k].score, (unsigned)k};
    auto stronger = [](const SurfScored &a, const SurfScored &b) { return a.score > b.score; };
    bool unique = lim > 0;
    if (lim > 0) {
        if (lim < n) {
            std::nth_element(sc.begin(), sc.begin() + (lim - 1), sc.end(), stronger);
            double rest = sc[lim].score;
            for (size_t k = lim + 1; k < n; k++) rest = std::max(rest, sc[k].score);
            unique = sc[lim - 1].score > rest;  // (position lim - 1 holds the weakest of the strongest)
        }
        if (unique) {
            std::sort(sc.begin(), sc.begin() + lim, stronger);
            for (size_t k = 1; k < lim && unique; k++) unique = sc[k - 1].score > sc[k].score;
        }
    }
    if (!unique && lim > 0) {
        std::vector<unsigned> emission;
        surf_emission_order(pts, emission);
        for (size_t k = 0; k < n; k++) sc[k] = {pts[emission[k]].score, emission[k]};
        std::sort(sc.rbegin(), sc.rend(), ScoreLess());
    }
    keep.clear();
    for (size_t k = 0; k < lim; k++) {
        const SurfRecord &p = pts[sc[k].idx];
        const unsigned long bs = (unsigned long)(32.0 * p.scale);
        const long px = (long)floor(p.x + 0.5), py = (long)floor(p.y + 0.5);
        const long l = px - (long)bs / 2, t = py - (long)bs / 2, r = l + (long)bs - 1, b = t + (long)bs - 1;
        if (l >= 0 && t >= 0 && r <= cols - 1 && b <= rows - 1) keep.push_back(sc[k].idx);
    }
}

// Helper threads of a context for the host half of imgfd_surf's K19 (atan2 of 109 samples per point: ~3 ms on one core for the
// 945 points of a bench tile).  Created on first use, parked on a condition variable between calls, joined when the context goes.
class SurfPool {
public:
    explicit SurfPool(unsigned n)
    {
        threads_.reserve(n);  // no reallocation (and no exception) while threads are running
        for (unsigned t = 0; t < n; t++) {
            try {
                threads_.emplace_back([this, t] { loop(t); });
            } catch (...) {  // no more threads to be had: the ones that exist do the work
                break;
            }
        }
    }
    ~SurfPool()
    {
        {
            std::lock_guard<std::mutex> g(m_);
            quit_ = true;
            generation_++;
        }
        start_.notify_all();
        for (auto &th : threads_) th.join();
    }
    // work(j0, j1) over [0, items), one share per thread (the caller takes a share too)
    void run(size_t items, const std::function<void(size_t, size_t)> &work)
    {
        const size_t parts = threads_.size() + 1, per = (items + parts - 1) / parts;
        {
            std::lock_guard<std::mutex> g(m_);
            work_ = &work; items_ = items; per_ = per;
            pending_ = threads_.size();
            generation_++;
        }
        start_.notify_all();
        const size_t j0 = std::min(items, threads_.size() * per);
        if (j0 < items) work(j0, items);  // the last share, here
        std::unique_lock<std::mutex> g(m_);
        done_.wait(g, [this] { return pending_ == 0; });
        work_ = nullptr;
    }

private:
    void loop(unsigned t)
    {
        unsigned long seen = 0;
        for (;;) {
            const std::function<void(size_t, size_t)> *w;
            size_t j0, j1;
            {
                std::unique_lock<std::mutex> g(m_);
                start_.wait(g, [&] { return generation_ != seen; });
                seen = generation_;
                if (quit_) return;
                w = work_;
                j0 = std::min(items_, (size_t)t * per_);
                j1 = std::min(items_, j0 + per_);
            }
            if (w && j0 < j1) (*w)(j0, j1);
            {
                std::lock_guard<std::mutex> g(m_);
                pending_--;
            }
            done_.notify_one();
        }
    }
    std::vector<std::thread> threads_;
    std::mutex m_;
    std::condition_variable start_, done_;
    const std::function<void(size_t, size_t)> *work_ = nullptr;
    size_t items_ = 0, per_ = 0, pending_ = 0;
    unsigned long generation_ = 0;
    bool quit_ = false;
};

// K19 buffers for m points: device side in the context's aux buffer, host side in its pinned buffer
struct SurfK19 {
    double *d_pts, *d_samples, *d_trig, *d_des;  // m x 3, m x 218, m x 5, m x 64
    double *h_pts, *h_samples, *h_trig, *h_des;
};

imgfd_status surf_k19_carve(imgfd_ctx *ctx, size_t m, SurfK19 *k)
{
    const size_t per = 3 + 2 * SURF_NSAMP + 5 + 64;
    IMGFD_TRY(aux_reserve(ctx, sizeof(double) * per * m));
    IMGFD_TRY(pin_reserve(ctx, sizeof(double) * per * m));
    double *d = reinterpret_cast<double *>(ctx->aux), *h = reinterpret_cast<double *>(ctx->pin);
    k->d_pts = d; k->d_samples = d + 3 * m; k->d_trig = k->d_samples + 2 * SURF_NSAMP * m; k->d_des = k->d_trig + 5 * m;
    k->h_pts = h; k->h_samples = h + 3 * m; k->h_trig = k->h_samples + 2 * SURF_NSAMP * m; k->h_des = k->h_trig + 5 * m;
    return IMGFD_OK;
}

// imgfd_surf's K19: Haar sampling and the descriptor on the device, atan2 / sin / cos on the host's glibc
imgfd_status surf_describe_assisted(imgfd_ctx *ctx, const SurfTable &T, const std::vector<SurfRecord> &pts,
                                    const std::vector<size_t> &keep, imgfd_surf_out *out)
{
    const size_t m = keep.size();
    SurfK19 k;
    IMGFD_TRY(surf_k19_carve(ctx, m, &k));
    for (size_t j = 0; j < m; j++) {
        const SurfRecord &p = pts[keep[j]];
        k.h_pts[3 * j] = p.x; k.h_pts[3 * j + 1] = p.y; k.h_pts[3 * j + 2] = p.scale;
    }
    IMGFD_HIP(ctx, hipMemcpyAsync(k.d_pts, k.h_pts, sizeof(double) * 3 * m, hipMemcpyHostToDevice, ctx->stream));
    IMGFD_TRY(launch_surf_orient(ctx, T, k.d_pts, (int)m, k.d_samples, nullptr));
    IMGFD_HIP(ctx, hipMemcpyAsync(k.h_samples, k.d_samples, sizeof(double) * 2 * SURF_NSAMP * m, hipMemcpyDeviceToHost, ctx->stream));
    IMGFD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    // points are independent: share them out over the context's helper threads (parked between calls: starting and joining fifteen
    // threads per call was 0.4 of the 1.0 ms this step took in round 5); every point runs the code one thread would run
    auto work = [&](size_t j0, size_t j1) {
        for (size_t j = j0; j < j1; j++)
            surf_orient_host(k.h_samples + 2 * SURF_NSAMP * j, k.h_samples + 2 * SURF_NSAMP * j + SURF_NSAMP, k.h_trig + 5 * j);
    };
    if (m < 128) {
        work(0, m);
    } else {
        if (!ctx->surf_pool) {
            ctx->surf_pool = new SurfPool(std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency())));
            ctx->surf_pool_free = [](void *p) { delete static_cast<SurfPool *>(p); };
        }
        static_cast<SurfPool *>(ctx->surf_pool)->run(m, work);
    }
    IMGFD_HIP(ctx, hipMemcpyAsync(k.d_trig, k.h_trig, sizeof(double) * 5 * m, hipMemcpyHostToDevice, ctx->stream));
    IMGFD_TRY(launch_surf_desc(ctx, T, k.d_pts, k.d_trig, (int)m, k.d_des, 64, nullptr));
    IMGFD_HIP(ctx, hipMemcpyAsync(k.h_des, k.d_des, sizeof(double) * 64 * m, hipMemcpyDeviceToHost, ctx->stream));
    IMGFD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (size_t j = 0; j < m; j++) out->angle[j] = k.h_trig[5 * j];
    memcpy(out->surf, k.h_des, sizeof(double) * 64 * m);
    return IMGFD_OK;
}

}  // namespace

extern "C" {

imgfd_status imgfd_k_surf_integral(imgfd_ctx *ctx, const uint8_t *rgb, int rows, int cols, int32_t *out)
try {
    if (!ctx) return IMGFD_ERR_INVALID;
    if (!rgb || !out || rows < 1 || cols < 1) return imgfd_fail(ctx, IMGFD_ERR_INVALID, "imgfd_k_surf_integral: bad argument");
    IMGFD_HIP(ctx, hipSetDevice(ctx->device));
    const size_t n = (size_t)rows * cols;
    const size_t part_bytes = std::max(sizeof(unsigned) * 32 * (size_t)cols, surf_integral_scratch(rows, cols));  // scratch of the scans
    IMGFD_TRY(ws_reserve(ctx, align_up(3 * n, 256) + align_up(4 * n, 256) + align_up(part_bytes, 256) + 512));
    uint8_t *d_rgb = (uint8_t *)ws_alloc(ctx, 3 * n);
    unsigned *d_I = (unsigned *)ws_alloc(ctx, 4 * n);
    void *d_part = ws_alloc(ctx, part_bytes);
    if (!d_rgb || !d_I || !d_part) return imgfd_fail(ctx, IMGFD_ERR_OOM, "workspace reservation too small");
    IMGFD_HIP(ctx, hipMemcpyAsync(d_rgb, rgb, 3 * n, hipMemcpyHostToDevice, ctx->stream));
    // the kernels and the layout the product path runs for this shape; a table in the residue layout (SurfTable) is put back
    // into row-major order on the host: the doorway returns integral_image.h:33-62's table either way
    const SurfTable T = launch_surf_integral(ctx, d_rgb, d_I, rows, cols, d_part, part_bytes, true);
    IMGFD_HIP(ctx, hipGetLastError());
    if (!T.per) {
        IMGFD_HIP(ctx, hipMemcpyAsync(out, d_I, 4 * n, hipMemcpyDeviceToHost, ctx->stream));
        IMGFD_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return IMGFD_OK;
    }
    std::vector<int32_t> j(n);
    IMGFD_HIP(ctx, hipMemcpyAsync(j.data(), d_I, 4 * n, hipMemcpyDeviceToHost, ctx->stream));
    IMGFD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int r = 0; r < rows; r++)
        for (int x = 0; x < cols; x++) out[(size_t)r * cols + x] = j[(size_t)r * cols + (size_t)(x & 3) * T.per + (x >> 2)];
    return IMGFD_OK;
} catch (const std::bad_alloc &) {
    return imgfd_fail(ctx, IMGFD_ERR_OOM, "imgfd_k_surf_integral: out of host memory");
} catch (...) {
    return imgfd_fail(ctx, IMGFD_ERR_HIP, "imgfd_k_surf_integral: unexpected C++ exception");
}

imgfd_status imgfd_surf_interest_points(imgfd_ctx *ctx, const uint8_t *rgb, int rows, int cols, double detection_threshold,
                                        double *points, int64_t cap, int64_t *n)
try {
    if (!ctx) return IMGFD_ERR_INVALID;
    if (!rgb || !n || rows < 0 || cols < 0 || cap < 0 || (cap && !points) || !(detection_threshold >= 0) || !frame_fits(rows, cols, 3))
        return imgfd_fail(ctx, IMGFD_ERR_INVALID, "imgfd_surf_interest_points: bad argument");
    std::vector<SurfRecord> pts;
    std::vector<unsigned> emission;
    IMGFD_TRY(surf_points_host(ctx, rgb, IMGFD_SRC_U8, rows, cols, detection_threshold, pts, &emission, nullptr));
    *n = (int64_t)pts.size();
    for (size_t k = 0; k < pts.size() && (int64_t)k < cap; k++) {
        const SurfRecord &p = pts[emission[k]];
        double *o = points + 5 * k;
        o[0] = p.x; o[1] = p.y; o[2] = p.scale; o[3] = p.score; o[4] = p.laplacian;
    }
    return IMGFD_OK;
} catch (const std::bad_alloc &) {
    return imgfd_fail(ctx, IMGFD_ERR_OOM, "imgfd_surf_interest_points: out of host memory");
} catch (...) {
    return imgfd_fail(ctx, IMGFD_ERR_HIP, "imgfd_surf_interest_points: unexpected C++ exception");
}

imgfd_status imgfd_surf_points_dev(imgfd_ctx *ctx, const uint8_t *d_rgb, int n_frames, int rows, int cols,
                                   size_t frame_stride_bytes, double detection_threshold, imgfd_surf_point *d_points,
                                   int64_t cap, int64_t *d_counts)
{
    if (!ctx) return IMGFD_ERR_INVALID;
    if (!d_rgb || !d_points || !d_counts || n_frames < 0 || rows < 1 || cols < 1 || cap < 1 || !(detection_threshold >= 0) ||
        !frame_fits(rows, cols, 3))
        return imgfd_fail(ctx, IMGFD_ERR_INVALID, "imgfd_surf_points_dev: bad argument");
    static_assert(sizeof(imgfd_surf_point) == sizeof(SurfRecord), "record layout");
    if (!n_frames) return IMGFD_OK;
    IMGFD_HIP(ctx, hipSetDevice(ctx->device));
    SurfGeom g;
    const size_t total = surf_geometry(rows, cols, &g);
    const size_t n = (size_t)rows * cols;
    IMGFD_TRY(ws_reserve(ctx, align_up(4 * n, 256) + align_up(8 * std::max<size_t>(g.mask_words, 1), 256) + align_up(surf_pyr_bytes(g, total), 256) +
                              align_up(sizeof(unsigned long long) * (size_t)cap, 256) + align_up(8 * std::max<size_t>(total, 1), 256) + 4096));
    SurfDevice d;
    d.integral = (unsigned *)ws_alloc(ctx, 4 * n);
    d.mask = (unsigned long long *)ws_alloc(ctx, 8 * std::max<size_t>(g.mask_words, 1));
    d.pyr_bytes = surf_pyr_bytes(g, total);
    d.pyr = (double *)ws_alloc(ctx, d.pyr_bytes);
    d.surv = (unsigned long long *)ws_alloc(ctx, sizeof(unsigned long long) * (size_t)cap);
    d.cand_cap = std::max<size_t>(total, 1);
    d.cands = (unsigned long long *)ws_alloc(ctx, 8 * (size_t)d.cand_cap);
    d.count = (unsigned long long *)ws_alloc(ctx, 256);
    if (!d.integral || !d.mask || !d.pyr || !d.surv || !d.cands || !d.count) return imgfd_fail(ctx, IMGFD_ERR_OOM, "workspace reservation too small");
    d.cap = (unsigned long long)cap;
    imgfd_ctx *fork = nullptr;
    IMGFD_TRY(ctx_side(ctx, &fork));
    for (int f = 0; f < n_frames; f++) {  // tiles are processed back to back on the context's stream, no host sync
        d.rec = reinterpret_cast<SurfRecord *>(d_points) + (size_t)f * cap;
        IMGFD_TRY(surf_device_stages(ctx, d_rgb + (size_t)f * frame_stride_bytes, g, detection_threshold, d, nullptr, fork));
        IMGFD_HIP(ctx, hipMemcpyAsync(d_counts + f, d.count, sizeof(int64_t), hipMemcpyDeviceToDevice, ctx->stream));  // (the counter slot also holds the lists' lengths)
    }
    return IMGFD_OK;
}

// imgfd_surf_dev / imgfd_surf_dev_redo.  `only` (redo): the tiles to run, with the record capacity each asked for; else all tiles.
//
// Schedule.  A tile is a chain of ten kernels: a FRONT that fills the chip (integral image 3 kernels, first octave, octaves 1-3:
// ~210 us one after the other on a 4096^2 tile, streaming the tile and its table) and a BACK of latency-bound kernels that do not
// (maximum test 54 us at issue 0.09, ranking 50 us in ONE workgroup, orientation 22, descriptor 33: chains of dependent scattered
// reads).  Rounds 3-5 ran whole chains on up to four streams ("lanes"): the small kernels of one tile were to fill the gaps of the
// other tiles' big ones -- but lanes that start together stay in step, beside another tile's gathers every streaming kernel ran
// at a quarter of its own rate (surf_int_sums 48-94 us instead of 13), and 0.222 ms per tile came out where the fronts alone take
// 0.144 (profiles/r06/surf_leave_one_out.txt, surf_timeline_two_lanes.txt).  Round 6: tiles go in GROUPS ("surf_group").  The
// fronts of a group run one after the other on a few streams ("surf_lanes": the first octave's kernel beside the next tile's
// scans), every tile into a buffer set of its own; then the back of the WHOLE group is four launches -- maximum test, ranking
// (a workgroup per tile: eight in flight instead of one), orientation, descriptor, blockIdx.y = tile -- so that the
// latency-bound kernels fill the chip with a group's worth of independent chains.  Groups of 4 on three streams are the default: over
// seconds of back-to-back calls (the power limit) 0.196 ms per tile against 0.2145 for groups of 8 on two streams, which win a 100 ms
// probe (profiles/r06/surf_groups_sustained.txt).
static imgfd_status surf_dev_run(imgfd_ctx *ctx, const uint8_t *d_rgb, int n_frames, int rows, int cols, size_t frame_stride_bytes,
                                 long max_points, double detection_threshold, double *d_features, int64_t cap, int64_t *d_counts,
                                 const std::vector<std::pair<int, unsigned long long>> *only)
{
    IMGFD_HIP(ctx, hipSetDevice(ctx->device));
    SurfGeom g;
    const size_t total = surf_geometry(rows, cols, &g);
    const size_t n = (size_t)rows * cols;
    const int n_run = only ? (int)only->size() : n_frames;
    if (!n_run) return IMGFD_OK;
    constexpr int MAX_LANES = 4, MAX_GROUP = 16;
    const int G = only ? 1 : std::max(1, std::min(std::min(ctx->tune.surf_group, MAX_GROUP), n_run));  // tiles per group
    const int nlanes = std::max(1, std::min(std::min(ctx->tune.surf_lanes, MAX_LANES), G));             // front streams
    imgfd_ctx *lane[MAX_LANES];
    lane[0] = ctx;
    for (int l = 1; l < nlanes; l++) {  // lane l's stream: the companion of lane l-1's context
        const imgfd_status st = ctx_side(lane[l - 1], &lane[l]);
        if (st != IMGFD_OK) { ctx->err = lane[l - 1]->err; return st; }
    }
    imgfd_ctx *fork = nullptr;  // one tile: its two pyramid kernels side by side (surf_front)
    if (G == 1) IMGFD_TRY(ctx_side(ctx, &fork));
    // More than one group: two BANKS of buffer sets, and the back of group k on a stream of its own beside the fronts of group k + 1
    const int banks = !only && n_run > G ? 2 : 1;
    imgfd_ctx *back = ctx;
    if (banks == 2) {
        IMGFD_TRY(ctx_side(lane[nlanes - 1], &back));
        while (ctx->surf_ev.size() < MAX_LANES + 2) {  // front done per lane, back done per bank
            hipEvent_t e;
            IMGFD_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
            ctx->surf_ev.push_back(e);
        }
    }
    SurfDevice sets[2][MAX_GROUP];
    unsigned *sel = nullptr, *cand = nullptr, *m_dev = nullptr;
    double *k19 = nullptr;
    unsigned lim = 0;
    auto carve = [&](unsigned long long rcap) -> imgfd_status {
        // no more records than the buffer holds can be asked for (and the all-pairs ranking is quadratic in that number)
        lim = (unsigned)std::min<unsigned long long>((unsigned long long)std::min<int64_t>((int64_t)max_points, cap), rcap);
        const imgfd_status st = ws_reserve(ctx, align_up(3 * n, 256) + (size_t)banks * G * (surf_ws_bytes(g, total, rcap) - align_up(3 * n, 256)) +
                                                    (size_t)G * (align_up(sizeof(unsigned) * 2 * (size_t)lim, 256) + align_up(sizeof(unsigned) * 2 * (size_t)rcap, 256) +
                                                                 align_up(sizeof(double) * 8 * (size_t)lim, 256)) + 4096);
        if (st != IMGFD_OK) return imgfd_fail(ctx, st, "imgfd_surf_dev: workspace allocation failed");
        (void)ws_alloc(ctx, 3 * n);  // the slot imgfd_surf uses for the uploaded image (same carving, same size function)
        for (int t = 0; t < banks * G; t++) {  // buffer sets, carved alike: consecutive sets are the same distance apart
            SurfDevice &d = sets[t / G][t % G];
            d.cap = rcap;
            d.integral = (unsigned *)ws_alloc(ctx, 4 * n);
            d.mask = (unsigned long long *)ws_alloc(ctx, 8 * std::max<size_t>(g.mask_words, 1));
            d.pyr_bytes = surf_pyr_bytes(g, total);
            d.pyr = (double *)ws_alloc(ctx, d.pyr_bytes);
            d.rec = (SurfRecord *)ws_alloc(ctx, sizeof(SurfRecord) * rcap);
            d.surv = (unsigned long long *)ws_alloc(ctx, sizeof(unsigned long long) * rcap);
            d.cand_cap = std::max<size_t>(total, 1);
            d.cands = (unsigned long long *)ws_alloc(ctx, 8 * (size_t)d.cand_cap);
            if (!d.integral || !d.mask || !d.pyr || !d.rec || !d.surv || !d.cands) return imgfd_fail(ctx, IMGFD_ERR_OOM, "workspace reservation too small");
        }
        unsigned long long *counters = (unsigned long long *)ws_alloc(ctx, 256 * (size_t)banks * G);  // one 256-byte slot per tile, side by side: one memset clears a group's
        if (!counters) return imgfd_fail(ctx, IMGFD_ERR_OOM, "workspace reservation too small");
        for (int t = 0; t < banks * G; t++) sets[t / G][t % G].count = counters + 32 * (size_t)t;
        sel = (unsigned *)ws_alloc(ctx, align_up(sizeof(unsigned) * 2 * (size_t)lim, 256) * G);
        cand = (unsigned *)ws_alloc(ctx, align_up(sizeof(unsigned) * 2 * (size_t)rcap, 256) * G);
        k19 = (double *)ws_alloc(ctx, align_up(sizeof(double) * 8 * (size_t)lim, 256) * G);  // per tile: x, y, scale | angle, sin, cos, sin(-), cos(-)
        m_dev = (unsigned *)ws_alloc(ctx, sizeof(unsigned) * MAX_GROUP);
        if (!sel || !cand || !k19 || !m_dev) return imgfd_fail(ctx, IMGFD_ERR_OOM, "workspace reservation too small");
        return IMGFD_OK;
    };
    // tiles f0 .. f0 + ng - 1 (ng <= G), group number k: fronts on the lanes, then the group's back (one bank: on the context's
    // stream; two: on the back stream, behind events)
    auto run_group = [&](int f0, int ng, const int *frames, int k) -> imgfd_status {
        SurfDevice *set = sets[k % banks];
        hipEvent_t *ev_lane = banks == 2 ? ctx->surf_ev.data() : nullptr, *ev_bank = banks == 2 ? ctx->surf_ev.data() + MAX_LANES : nullptr;
        if (banks == 2 && k >= 2)  // this bank's sets were last read by the back of the group two before
            for (int l = 0; l < std::min(nlanes, ng); l++) IMGFD_HIP(ctx, hipStreamWaitEvent(lane[l]->stream, ev_bank[k % 2], 0));
        SurfTable T{nullptr, rows, cols, 0};
        for (int t = 0; t < ng; t++) {
            const int f = frames ? frames[t] : f0 + t;
            imgfd_ctx *c = lane[t % nlanes];
            SurfStreamScope scope(ctx, c->stream);  // every buffer lives in the context's arena; the lane lends its stream
            SurfTable Tt;
            IMGFD_TRY(surf_front(ctx, d_rgb + (size_t)f * frame_stride_bytes, g, detection_threshold, set[t], &Tt, fork));
            if (t == 0) T = Tt;  // same geometry, same carving: every tile's table lies alike, set[t].integral apart
        }
        if (banks == 2) {
            for (int l = 0; l < std::min(nlanes, ng); l++) {  // the back waits for every lane's fronts
                IMGFD_HIP(ctx, hipEventRecord(ev_lane[l], lane[l]->stream));
                IMGFD_HIP(ctx, hipStreamWaitEvent(back->stream, ev_lane[l], 0));
            }
        } else {
            for (int l = 1; l < std::min(nlanes, ng); l++) {
                hipEvent_t ev = lane[l - 1]->ev_join;  // the event pair of (lane l-1, its companion)
                IMGFD_HIP(ctx, hipEventRecord(ev, lane[l]->stream));
                IMGFD_HIP(ctx, hipStreamWaitEvent(ctx->stream, ev, 0));
            }
        }
        SurfStreamScope back_scope(ctx, back->stream);
        const SurfDevice *next = ng > 1 ? &set[1] : nullptr;
        IMGFD_TRY(surf_back_nms(ctx, g, detection_threshold, set[0], T, ng, next));
        const size_t sel_stride = align_up(sizeof(unsigned) * 2 * (size_t)lim, 256) / sizeof(unsigned);
        const size_t cand_stride = align_up(sizeof(unsigned) * 2 * (size_t)set[0].cap, 256) / sizeof(unsigned);
        const size_t k19_stride = align_up(sizeof(double) * 8 * (size_t)lim, 256) / sizeof(double);
        const int fa = frames ? frames[0] : f0;
        double *feat = d_features + (size_t)fa * (size_t)cap * 70;
        SurfRankParams q;
        q.rec = set[0].rec; q.count = set[0].count; q.cap = set[0].cap; q.lim = lim; q.rows = rows; q.cols = cols; q.sel = sel; q.order = sel + lim;
        q.cand = cand; q.sort_cap = (unsigned)std::max(0, ctx->tune.surf_sort_cap);
        q.pts = k19; q.feat = feat; q.count_out = reinterpret_cast<long long *>(d_counts) + fa; q.m_out = m_dev;
        q.tile_rec = next ? (size_t)(next->rec - set[0].rec) : 0; q.tile_count = next ? (size_t)(next->count - set[0].count) : 0;
        q.tile_sel = sel_stride; q.tile_cand = cand_stride; q.tile_pts = k19_stride; q.tile_feat = (size_t)cap * 70;
        hipLaunchKernelGGL(surf_rank_select, dim3((unsigned)ng), dim3(SR_NT), 0, ctx->stream, q);
        const SurfGroup grp{ng, next ? (size_t)(next->integral - set[0].integral) : 0, k19_stride, k19_stride, (size_t)cap * 70};
        IMGFD_TRY(launch_surf_orient(ctx, T, k19, (int)lim, nullptr, k19 + 3 * (size_t)lim, m_dev, &grp));
        IMGFD_TRY(launch_surf_desc(ctx, T, k19, k19 + 3 * (size_t)lim, (int)lim, feat + 6, 70, feat + 2, m_dev, &grp));
        if (banks == 2) IMGFD_HIP(ctx, hipEventRecord(ev_bank[k % 2], back->stream));
        return IMGFD_OK;
    };
    if (!only) {
        const unsigned long long rec_cap = (unsigned long long)std::max(16, ctx->tune.surf_rec_cap);  // candidate records per tile (262144: 14 MB); a tile with more reports -candidates
        IMGFD_TRY(carve(rec_cap));
        int k = 0;
        for (int f0 = 0; f0 < n_frames; f0 += G, k++) {
            // the lanes' streams start behind what the context's stream holds: the tiles' producer (first group) and, with one bank,
            // the back of the group before (whose buffer sets the fronts are about to overwrite)
            if (nlanes > 1 && (k == 0 || banks == 1)) {
                IMGFD_HIP(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
                for (int l = 1; l < nlanes; l++) IMGFD_HIP(ctx, hipStreamWaitEvent(lane[l]->stream, ctx->ev_fork, 0));
            }
            IMGFD_TRY(run_group(f0, std::min(G, n_frames - f0), nullptr, k));
        }
        if (banks == 2) {  // whoever waits for the context's stream waits for the lanes and for the last backs
            for (int l = 1; l < nlanes; l++) {
                IMGFD_HIP(ctx, hipEventRecord(ctx->surf_ev[l], lane[l]->stream));
                IMGFD_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->surf_ev[l], 0));
            }
            for (int b = 0; b < std::min(2, k); b++) IMGFD_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->surf_ev[MAX_LANES + b], 0));
        }
    } else {
        // redo: one tile at a time, each with the buffer it asked for (the arena is carved anew per tile: ws_reserve waits for
        // everything queued before when it has to grow, and the context's stream orders the rest)
        for (const auto &job : *only) {
            IMGFD_HIP(ctx, hipStreamSynchronize(ctx->stream));
            IMGFD_TRY(carve(job.second));
            const int f = job.first;
            IMGFD_TRY(run_group(f, 1, &f, 0));
        }
    }
    IMGFD_HIP(ctx, hipGetLastError());
    return IMGFD_OK;
}

imgfd_status imgfd_surf_dev(imgfd_ctx *ctx, const uint8_t *d_rgb, int n_frames, int rows, int cols, size_t frame_stride_bytes,
                            long max_points, double detection_threshold, double *d_features, int64_t cap, int64_t *d_counts)
try {
    if (!ctx) return IMGFD_ERR_INVALID;
    if (!d_rgb || !d_features || !d_counts || n_frames < 0 || rows < 1 || cols < 1 || cap < 1 || !(max_points > 0) ||
        !(detection_threshold >= 0) || !frame_fits(rows, cols, 3))
        return imgfd_fail(ctx, IMGFD_ERR_INVALID, "imgfd_surf_dev: bad argument");
    if (!n_frames) return IMGFD_OK;
    return surf_dev_run(ctx, d_rgb, n_frames, rows, cols, frame_stride_bytes, max_points, detection_threshold, d_features, cap, d_counts, nullptr);
} catch (const std::bad_alloc &) {
    return imgfd_fail(ctx, IMGFD_ERR_OOM, "imgfd_surf_dev: out of host memory");
} catch (...) {
    return imgfd_fail(ctx, IMGFD_ERR_HIP, "imgfd_surf_dev: unexpected C++ exception");
}

// A tile with more candidates than its record buffer holds has left -candidates in its count and no features (its best
// records may be among those that found no room).  This call waits for the context's stream, reads the counts of the batch back
// and redoes such tiles with a buffer of the size they asked for.
imgfd_status imgfd_surf_dev_redo(imgfd_ctx *ctx, const uint8_t *d_rgb, int n_frames, int rows, int cols, size_t frame_stride_bytes,
                                 long max_points, double detection_threshold, double *d_features, int64_t cap, int64_t *d_counts,
                                 int *n_redone)
try {
    if (!ctx) return IMGFD_ERR_INVALID;
    if (n_redone) *n_redone = 0;
    if (!d_rgb || !d_features || !d_counts || n_frames < 0 || rows < 1 || cols < 1 || cap < 1 || !(max_points > 0) ||
        !(detection_threshold >= 0) || !frame_fits(rows, cols, 3))
        return imgfd_fail(ctx, IMGFD_ERR_INVALID, "imgfd_surf_dev_redo: bad argument");
    if (!n_frames) return IMGFD_OK;
    IMGFD_HIP(ctx, hipSetDevice(ctx->device));
    std::vector<int64_t> h((size_t)n_frames);
    for (int tries = 0; tries < 4; tries++) {
        IMGFD_HIP(ctx, hipMemcpyAsync(h.data(), d_counts, sizeof(int64_t) * (size_t)n_frames, hipMemcpyDeviceToHost, ctx->stream));
        IMGFD_HIP(ctx, hipStreamSynchronize(ctx->stream));
        std::vector<std::pair<int, unsigned long long>> jobs;
        for (int f = 0; f < n_frames; f++)
            if (h[(size_t)f] < 0) jobs.emplace_back(f, (unsigned long long)(-h[(size_t)f]) + 1024);
        if (jobs.empty()) return IMGFD_OK;
        if (n_redone && !tries) *n_redone = (int)jobs.size();
        IMGFD_TRY(surf_dev_run(ctx, d_rgb, n_frames, rows, cols, frame_stride_bytes, max_points, detection_threshold, d_features, cap, d_counts, &jobs));
    }
    IMGFD_HIP(ctx, hipMemcpyAsync(h.data(), d_counts, sizeof(int64_t) * (size_t)n_frames, hipMemcpyDeviceToHost, ctx->stream));
    IMGFD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int f = 0; f < n_frames; f++)
        if (h[(size_t)f] < 0) return imgfd_fail(ctx, IMGFD_ERR_OOM, "imgfd_surf_dev_redo: a tile's candidates do not fit its record buffer");
    return IMGFD_OK;
} catch (const std::bad_alloc &) {
    return imgfd_fail(ctx, IMGFD_ERR_OOM, "imgfd_surf_dev_redo: out of host memory");
} catch (...) {
    return imgfd_fail(ctx, IMGFD_ERR_HIP, "imgfd_surf_dev_redo: unexpected C++ exception");
}

static imgfd_status surf_host(imgfd_ctx *ctx, const void *rgb, int kind, int rows, int cols, long max_points,
                              double detection_threshold, imgfd_surf_out *out)
{
    if (!ctx || !out) return IMGFD_ERR_INVALID;
    memset(out, 0, sizeof *out);
    if (!rgb || rows < 0 || cols < 0 || !(max_points > 0) || !(detection_threshold >= 0) || !frame_fits(rows, cols, 3))
        return imgfd_fail(ctx, IMGFD_ERR_INVALID, "imgfd_surf: bad argument (DLIB_ASSERT of surf.h:243-248)");
    std::vector<SurfRecord> pts;
    SurfTable T{nullptr, rows, cols, 0};
    IMGFD_TRY(surf_points_host(ctx, rgb, kind, rows, cols, detection_threshold, pts, nullptr, &T));
    if (pts.empty()) return IMGFD_OK;
    std::vector<size_t> keep;
    surf_select(pts, max_points, rows, cols, keep);
    const size_t m = keep.size();
    if (!m) return IMGFD_OK;
    double *data = (double *)malloc(sizeof(double) * m * 70);
    if (!data) return imgfd_fail(ctx, IMGFD_ERR_OOM, "malloc of the SURF output failed");
    out->n = (int64_t)m; out->data = data;
    out->x = data; out->y = data + m; out->angle = data + 2 * m; out->pyramid_scale = data + 3 * m;
    out->score = data + 4 * m; out->laplacian = data + 5 * m; out->surf = data + 6 * m;
    for (size_t j = 0; j < m; j++) {
        const SurfRecord &p = pts[keep[j]];
        out->x[j] = p.x; out->y[j] = p.y; out->pyramid_scale[j] = p.scale; out->score[j] = p.score; out->laplacian[j] = p.laplacian;
    }
    const imgfd_status st = surf_describe_assisted(ctx, T, pts, keep, out);
    if (st != IMGFD_OK) {
        free(data);
        memset(out, 0, sizeof *out);
    }
    return st;
}

imgfd_status imgfd_surf(imgfd_ctx *ctx, const uint8_t *rgb, int rows, int cols, long max_points, double detection_threshold,
                        imgfd_surf_out *out)
{
    return imgfd_guard(ctx, [&] { return surf_host(ctx, rgb, IMGFD_SRC_U8, rows, cols, max_points, detection_threshold, out); });
}

imgfd_status imgfd_surf_i32(imgfd_ctx *ctx, const int32_t *x, int rows, int cols, long max_points, double detection_threshold,
                            imgfd_surf_out *out)
{
    return imgfd_guard(ctx, [&] { return surf_host(ctx, x, IMGFD_SRC_I32, rows, cols, max_points, detection_threshold, out); });
}

}  // extern "C"
