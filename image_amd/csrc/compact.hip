// image_amd/csrc/compact.hip -- raster-ordered stream compaction shared by Harris NMS and FAST-9.
//
// Both reference detectors return their features in raster order (Harris: corners_row[] concatenated
// row by row, harris.cpp:251-252; FAST-9: the y/x double loop of f9.cpp:2959-2960), and that order is
// part of what the R functions return.  The detector kernels therefore do not append with atomics;
// they publish one bit per pixel (a 64-bit __ballot word per wave) plus integer per-row counts.  This
// file turns that into the ordered list:
//   scatter   : one wave per row, lane l owns mask word l (+64k): wave-level prefix of popcounts, then
//               each lane walks its set bits and writes records at (offset of the row) + prefix + k.  The offset of a
//               workgroup's first row is the sum of the counts of all rows above it, which the workgroup adds up itself
//               (ny <= SCATTER_SELF_SCAN_MAX_ROWS: a few KB from L2 -- one launch less per detector, round 5);
//   rows_scan : one workgroup per frame, exclusive scan of the ny row counts -> row offsets + total (taller frames, and
//               calls that want the counts only)
// Integer-only bookkeeping: the output is deterministic and identical to a sequential scan.
#include "common.h"
#include "harris_device.h"

#define SCAN_NT 256  // four waves: a 16-wave workgroup never found a CU with 16 free slots while another stream kept refilling them with 4-wave workgroups (rows_scan: 686 us in the two-stream step)

size_t compact_bytes(int nx, int ny, int n_frames)
{
    const size_t wpr = (size_t)ceil_div(nx, 64);
    return align_up(sizeof(unsigned long long) * wpr * ny * n_frames, 256) +
           2 * align_up(sizeof(unsigned) * (size_t)ny * n_frames, 256);
}

imgfd_status compact_carve(imgfd_ctx *ctx, int nx, int ny, int n_frames, CompactBuffers *cb)
{
    cb->words_per_row = ceil_div(nx, 64);
    cb->mask = (unsigned long long *)ws_alloc(ctx, sizeof(unsigned long long) * (size_t)cb->words_per_row * ny * n_frames);
    cb->rowcount = (unsigned *)ws_alloc(ctx, sizeof(unsigned) * (size_t)ny * n_frames);
    cb->rowoff = (unsigned *)ws_alloc(ctx, sizeof(unsigned) * (size_t)ny * n_frames);
    if (!cb->mask || !cb->rowcount || !cb->rowoff) return imgfd_fail(ctx, IMGFD_ERR_OOM, "workspace reservation too small");
    return IMGFD_OK;
}

imgfd_status compact_clear(imgfd_ctx *ctx, const CompactBuffers &cb, int ny, int n_frames)
{
    IMGFD_HIP(ctx, hipMemsetAsync(cb.rowcount, 0, sizeof(unsigned) * (size_t)ny * n_frames, ctx->stream));
    return IMGFD_OK;
}

__global__ void __launch_bounds__(SCAN_NT) rows_scan(const unsigned *__restrict__ rowcount,
                                                     unsigned *__restrict__ rowoff, int ny,
                                                     long long *__restrict__ counts)
{
    __shared__ unsigned wsum[SCAN_NT / 64];
    __shared__ unsigned carry_s;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const unsigned *rc = rowcount + (size_t)blockIdx.x * ny;
    unsigned *ro = rowoff + (size_t)blockIdx.x * ny;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < ny; base += SCAN_NT) {
        const int i = base + tid;
        const unsigned v = i < ny ? rc[i] : 0u;
        // inclusive scan inside the wave
        unsigned incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned t = __shfl_up(incl, d);
            if (lane >= d) incl += t;
        }
        if (lane == 63) wsum[wv] = incl;
        __syncthreads();
        unsigned wbase = 0;
        for (int w = 0; w < wv; w++) wbase += wsum[w];
        const unsigned carry = carry_s;
        if (i < ny) ro[i] = carry + wbase + incl - v;
        __syncthreads();
        if (tid == SCAN_NT - 1) carry_s = carry + wbase + incl;
        __syncthreads();
    }
    if (tid == 0) counts[blockIdx.x] = (long long)carry_s;
}

struct AbcSource {  // KIND 2..4: strength recomputed from the structure tensor (measure = KIND - 2)
    const float *A, *B, *C;
    float k;
};

// KIND 0: imgfd_corner {x, y, R[y*nx+x]};  KIND 1: imgfd_point {x, y};  KIND 2+m: imgfd_corner with R = response_m(A,B,C)
// SELF_SCAN: `rowoff` holds the row COUNTS and the workgroup derives its rows' offsets (and, the last one, the frame's total)
#define SCATTER_SELF_SCAN_MAX_ROWS 16384
template <int KIND, bool SELF_SCAN>
__global__ void __launch_bounds__(256) scatter_rows(const unsigned long long *__restrict__ mask,
                                                    const unsigned *__restrict__ rowoff, int words_per_row,
                                                    int nx, int ny, const float *__restrict__ R, AbcSource abc,
                                                    void *__restrict__ out, long long cap, long long *__restrict__ counts)
{
    const int lane = threadIdx.x & 63;
    const int y = blockIdx.x * 4 + (threadIdx.x >> 6);  // one wave per row
    const int frame = blockIdx.y;
    const bool row_ok = y < ny;
    const int yy = row_ok ? y : 0;
    const unsigned long long *mrow = mask + ((size_t)frame * ny + yy) * words_per_row;
    unsigned running;
    if (SELF_SCAN) {
        __shared__ unsigned part[4];
        const unsigned *rc = rowoff + (size_t)frame * ny;
        const int y0 = blockIdx.x * 4;
        unsigned sum = 0;
        for (int i = threadIdx.x; i < y0; i += 256) sum += rc[i];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) sum += __shfl_xor(sum, d);
        if (lane == 0) part[threadIdx.x >> 6] = sum;
        __syncthreads();
        running = part[0] + part[1] + part[2] + part[3];
        for (int r = y0; r < min(y, ny); r++) running += rc[r];  // the rows of the waves before this one (at most three, wave-uniform; never past the frame's last row)
        if (y == ny - 1 && lane == 0) counts[frame] = (long long)(running + rc[y]);
    } else {
        running = row_ok ? rowoff[(size_t)frame * ny + yy] : 0u;
    }
    for (int w0 = 0; w0 < words_per_row; w0 += 64) {
        const int w = w0 + lane;
        unsigned long long m = (row_ok && w < words_per_row) ? mrow[w] : 0ull;
        const unsigned c = (unsigned)__popcll(m);
        unsigned incl = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned t = __shfl_up(incl, d);
            if (lane >= d) incl += t;
        }
        const unsigned total = __shfl(incl, 63);
        unsigned pos = running + incl - c;
        while (m) {
            const int b = __ffsll((long long)m) - 1;
            m &= m - 1;
            const int x = w * 64 + b;
            if ((long long)pos < cap) {
                if (KIND != 1) {
                    imgfd_corner *o = reinterpret_cast<imgfd_corner *>(out) + (size_t)frame * cap + pos;
                    const size_t p = ((size_t)frame * ny + y) * nx + x;
                    o->x = (float)x;
                    o->y = (float)y;
                    if (KIND == 0) o->R = R[p];
                    else o->R = harris_response_value<(KIND >= 2 ? KIND - 2 : 0)>(abc.A[p], abc.B[p], abc.C[p], abc.k);
                } else {
                    imgfd_point *o = reinterpret_cast<imgfd_point *>(out) + (size_t)frame * cap + pos;
                    o->x = x;
                    o->y = y;
                }
            }
            pos++;
        }
        running += total;
    }
}

static imgfd_status compact_emit_impl(imgfd_ctx *ctx, const CompactBuffers &cb, int nx, int ny, int n_frames, int kind,
                                      const float *d_R, const AbcSource &abc, void *d_out, int64_t cap, int64_t *d_counts)
{
    // small batches only: a single frame saves the rows_scan launch of its critical chain; at 32 frames the 540 workgroups per frame that each
    // add up to 2160 counts take 29 us where rows_scan + scatter_rows took 24.5 (profiles/r05/a_pmc_all_kernels.txt)
    const bool self_scan = cap > 0 && ny <= SCATTER_SELF_SCAN_MAX_ROWS && n_frames < 8;
    if (!self_scan)
        hipLaunchKernelGGL(rows_scan, dim3(n_frames), dim3(SCAN_NT), 0, ctx->stream, cb.rowcount, cb.rowoff, ny,
                           (long long *)d_counts);
    if (cap <= 0) {  // counts only: nothing to emit
        IMGFD_HIP(ctx, hipGetLastError());
        return IMGFD_OK;
    }
    dim3 grid(ceil_div(ny, 4), n_frames);
#define SC_LAUNCH(K)                                                                                                                      \
    do {                                                                                                                                  \
        if (self_scan)                                                                                                                    \
            hipLaunchKernelGGL((scatter_rows<K, true>), grid, dim3(256), 0, ctx->stream, cb.mask, (const unsigned *)cb.rowcount, cb.words_per_row, nx, ny, \
                               d_R, abc, d_out, (long long)cap, (long long *)d_counts);                                                   \
        else                                                                                                                              \
            hipLaunchKernelGGL((scatter_rows<K, false>), grid, dim3(256), 0, ctx->stream, cb.mask, (const unsigned *)cb.rowoff, cb.words_per_row, nx, ny,  \
                               d_R, abc, d_out, (long long)cap, (long long *)d_counts);                                                   \
    } while (0)
    switch (kind) {
        case 0: SC_LAUNCH(0); break;
        case 1: SC_LAUNCH(1); break;
        case 2: SC_LAUNCH(2); break;
        case 3: SC_LAUNCH(3); break;
        default: SC_LAUNCH(4); break;
    }
#undef SC_LAUNCH
    IMGFD_HIP(ctx, hipGetLastError());
    return IMGFD_OK;
}

imgfd_status compact_emit(imgfd_ctx *ctx, const CompactBuffers &cb, int nx, int ny, int n_frames, int kind,
                          const float *d_R, void *d_out, int64_t cap, int64_t *d_counts)
{
    AbcSource none{nullptr, nullptr, nullptr, 0.f};
    return compact_emit_impl(ctx, cb, nx, ny, n_frames, kind, d_R, none, d_out, cap, d_counts);
}

// corner records whose strength is recomputed from A, B, C (fused response + NMS path: no R plane exists)
imgfd_status compact_emit_abc(imgfd_ctx *ctx, const CompactBuffers &cb, int nx, int ny, int n_frames, const float *d_A,
                              const float *d_B, const float *d_C, int measure, float k, void *d_out, int64_t cap,
                              int64_t *d_counts)
{
    AbcSource abc{d_A, d_B, d_C, k};
    const int m = measure == IMGFD_SHI_TOMASI_MEASURE ? 1 : (measure == IMGFD_HARMONIC_MEAN_MEASURE ? 2 : 0);
    return compact_emit_impl(ctx, cb, nx, ny, n_frames, 2 + m, nullptr, abc, d_out, cap, d_counts);
}
