// image_amd/csrc/fir_tensor.hip -- the Harris structure-tensor pass (K3) for gfx950.
//
// Replaces compute_autocorrelation_matrix(), image.CornerDetectionHarris/src/harris.cpp:44-70: the float products
// Ix*Ix, Ix*Iy, Iy*Iy (:57-62) followed by three in-place discrete Gaussians (:67-69, gaussian.cpp:289-395), i.e. per
// plane a horizontal and a vertical 1-D pass, each accumulated in double in the reference's order
//     B[0]*x[i] + sum_j B[j]*(x[i-j] + x[i+j]),   j ascending, pair added first, ONE rounding to float per pass
// (gaussian.cpp:351-359, 382-390; borders: left/top -k -> k, right/bottom n-1+k -> n-k, :345-349).  FMA = false
// issues exactly that sequence (library built -ffp-contract=off); FMA = true fuses only sum = fma(B[j], pair, sum).
// With OUT = 2 the corner response of harris.cpp:78-133 (Harris measure) is evaluated on the smoothed A, B, C before
// they leave the CU and only R is written (4 B/px instead of 12).
//
// The reference's double accumulation makes this pass f64-issue bound on the vector pipe (90 f64 operations per
// pixel before conversions), so the kernel is organised around the f64 instruction count, not around bytes:
//
//   * one 384-thread workgroup (6 waves) owns a 128-column strip of one frame and marches down a segment of rows in
//     chunks of 16 rows; wave pair {0,1} / {2,3} / {4,5} works on plane A / B / C in both passes (wave-uniform).
//   * row pass: a thread owns 16 consecutive pixels of one row of its plane: 16+2R window values are read from the raw
//     Ix/Iy tile in LDS with ds_read_b128 (row pitch = odd number of 16-byte slots: every 16-lane group of the b128
//     read pattern hits 16 distinct slot banks), multiplied in float, widened ONCE to double ((16+2R)/16 conversions
//     per pixel) and slid through the 16 outputs in registers; the rounded floats go to an LDS ring [plane][16][128].
//   * column pass: a thread owns ONE column of one plane for the whole segment and keeps the last 16 row-filtered
//     values of that column in registers as doubles (a circular buffer indexed by row mod 16 -- static indices, since
//     a chunk is 16 rows): per output row one ds_read_b32 (lane <-> column: conflict-free), one conversion, the 2R+1
//     tap chain, one rounding, one coalesced 256-byte store per wave.  No window re-reads, no halo rows in LDS.
//   * the Ix/Iy tile of chunk c+1 is fetched (aligned float4 loads, straight-line) while chunk c computes and is
//     written to LDS between the passes; border strips rebuild the reflected halo columns inside LDS.
//
// LDS: 2 x 16 x 37 float4 (raw) + 3 x 16 x 128 floats (ring) = 43.5 KB -> three workgroups (18 waves) per CU.
// HBM traffic is the algorithmic 8 B read + 12 B (or 4 B) written per pixel plus strip/segment halos (served by L2).
#include "common.h"
#include "fir_device.h"
#include "harris_device.h"

typedef float ft_v4f __attribute__((vector_size(16)));  // native vector: always promoted to registers
#ifdef HIPEMU
typedef volatile ft_v4f ft_lds_v4f;
#else
// a volatile 16-byte read that is known to address LDS (a volatile access through a generic pointer becomes a flat load)
typedef volatile __attribute__((address_space(3))) ft_v4f ft_lds_v4f;
#endif

struct TensorParams {
    const float *ix;
    const float *iy;
    float *out0, *out1, *out2;  // A, B, C -- or R in out0 (OUT = 2)
    int nx, ny;
    long frame_stride;  // elements between frames (planes are packed: pitch nx)
    int seg_rows;       // output rows per workgroup segment
    int xcd_remap;
    float k;            // Harris constant (OUT = 2)
    double B[8];        // taps B[0..R], R <= 7
};

template <int R, int TW_>
struct TensorGeom {
    static constexpr int TW = TW_, CH = 16, NT = 3 * TW_, PX = 16;
    static constexpr int NS = TW / PX;            // 16-pixel strips per tile row
    static constexpr int RPITCH = TW == 128 ? 128 : TW + 4;  // ring row pitch in floats (TW = 256: +1 slot, see ft_row_task)
    static constexpr int HALO = (R + 3) / 4 * 4;  // tile halo in whole float4 slots: x0-HALO is 16-byte aligned
    static constexpr int W = TW + 2 * HALO, W4 = W / 4;
    static constexpr int P4 = W4 | 1;             // row pitch in float4 slots, odd (see header)
    static constexpr int OFF = HALO - R;          // window start inside a strip's first slot
    static constexpr int NW = PX + 2 * R;         // window length
    static constexpr int NW4 = (OFF + NW + 3) / 4;
    static constexpr int TILE4 = 2 * CH * W4;     // float4 slots of one chunk's tile (Ix and Iy)
    static constexpr int NL = (TILE4 + NT - 1) / NT;
    static constexpr size_t LDS_BYTES = sizeof(float4) * 2 * CH * P4 + sizeof(float) * 3 * CH * RPITCH;
    static_assert(TW == 128 || TW == 256, "strip widths with a conflict-free lane mapping");
    static_assert(2 * R <= CH, "the column pass reaches 2R rows back into the previous chunk");
    static_assert(16 * (NS - 1) + 4 * NW4 <= W, "row-pass window reads stay inside the tile row");
};

// ring position of column c of a row: the four float4 slots of each 16-column strip are rotated by (strip / 2), so the
// row pass's ds_write_b128 (8 lanes = 8 strips of one row) covers 8 distinct slots mod 8 and the column pass's
// ds_read_b32 still reads 64 consecutive dwords per wave in some order
__device__ __forceinline__ int ft_ring_slot4(int s, int h) { return 4 * s + ((h + (s >> 1)) & 3); }
__device__ __forceinline__ int ft_ring_col(int c) { return 4 * ft_ring_slot4(c >> 4, (c >> 2) & 3) + (c & 3); }

// row pass of one (row, strip) of plane PL: raw tile -> 16 row-filtered floats in the ring
template <int R, int TW, bool FMA, int PL>
__device__ __forceinline__ void ft_row_task(const float4 *raw4, float *ring, int r, int s, const double *B)
{
    using G = TensorGeom<R, TW>;
    constexpr int NW = G::NW, NW4 = G::NW4, OFF = G::OFF;
    float wx[NW4 * 4], wy[NW4 * 4];
    // volatile: keeps each window slot ONE 16-byte LDS read (ds_read_b128).  Left to itself the optimiser splits the
    // float4 loads into the elements the window uses and re-pairs them as ds_read2_b32, whose 32-lane groups then hit
    // the strips' 64-byte stride 4-way (PMC: 59 % of the LDS cycles were bank conflicts)
    const ft_lds_v4f *rx = (const ft_lds_v4f *)(raw4 + (0 * G::CH + r) * G::P4 + 4 * s);
    const ft_lds_v4f *ry = (const ft_lds_v4f *)(raw4 + (1 * G::CH + r) * G::P4 + 4 * s);
#pragma unroll
    for (int q = 0; q < NW4; q++) {
        if (PL != 2) {
            const ft_v4f v = rx[q];
            wx[4 * q] = v[0]; wx[4 * q + 1] = v[1]; wx[4 * q + 2] = v[2]; wx[4 * q + 3] = v[3];
        }
        if (PL != 0) {
            const ft_v4f u = ry[q];
            wy[4 * q] = u[0]; wy[4 * q + 1] = u[1]; wy[4 * q + 2] = u[2]; wy[4 * q + 3] = u[3];
        }
    }
    double d[NW];
#pragma unroll
    for (int k = 0; k < NW; k++) {
        float v;
        if (PL == 0) v = wx[OFF + k] * wx[OFF + k];       // harris.cpp:59
        else if (PL == 1) v = wx[OFF + k] * wy[OFF + k];  // harris.cpp:60
        else v = wy[OFF + k] * wy[OFF + k];               // harris.cpp:61
        d[k] = (double)v;
    }
    float o[G::PX];
    fir_window8<R, FMA, G::PX>(d, B, o);
    float4 *dst = reinterpret_cast<float4 *>(ring + (PL * G::CH + r) * G::RPITCH);
#pragma unroll
    for (int h = 0; h < 4; h++) dst[ft_ring_slot4(s, h)] = make_float4(o[4 * h], o[4 * h + 1], o[4 * h + 2], o[4 * h + 3]);
}

#ifdef HIPEMU
#define FT_WAVES_PER_EU(tw)
#else
// Register budget = what the wave placement needs, not what the occupancy API answers.  A workgroup's waves are dealt to
// the CU's four SIMDs in turn, so only multiples of 4 waves load the SIMDs evenly:
//   TW 256: 12 waves = 3 per SIMD, one workgroup per CU (85 KB of LDS): 168 VGPRs.
//   TW 128: 6 waves land 2,2,1,1; two workgroups can need FOUR slots on a SIMD: 128 VGPRs (with the allocator's own
//           160 the hardware admitted ONE workgroup per CU: 1.4 resident waves per SIMD measured), and the SIMDs that
//           hold two waves of a workgroup set the pace of its barriers -- 75 % of the f64 rate at best (measured).
#define FT_WAVES_PER_EU(tw) __attribute__((amdgpu_waves_per_eu((tw) == 256 ? 3 : 4, (tw) == 256 ? 3 : 4)))
#endif
// OUT 0: A, B, C stored straight from the column pass (one dword per lane and row, 256 B per wave)
// OUT 1: A, B, C staged through the ring and stored as float4 rows
// OUT 2: corner response (Harris measure) computed from the staged A, B, C; only R is stored (float4 rows)
template <int R, int TW, bool FMA, bool VEC, int OUT>
__global__ void __launch_bounds__(3 * TW) FT_WAVES_PER_EU(TW) fir_tensor(TensorParams p)
{
    using G = TensorGeom<R, TW>;
    constexpr int CH = G::CH, NT = G::NT, W4 = G::W4, P4 = G::P4, HALO = G::HALO, NL = G::NL, RP = G::RPITCH;

    HIP_DYNAMIC_SHARED(float4, smem4)
    float4 *raw4 = smem4;                                              // [2][CH][P4]
    float *ring = reinterpret_cast<float *>(smem4 + 2 * CH * P4);      // [3][CH][RP], columns permuted per strip

    const int tid = threadIdx.x;
    int bx = blockIdx.x, by = blockIdx.y;
    if (p.xcd_remap) {
        // workgroup ids are dealt round-robin to the 8 XCDs: let XCD x own a contiguous run of (strip, segment) tiles, so
        // the halo columns/rows two neighbouring workgroups share are re-read from that XCD's own L2
        const int total = gridDim.x * gridDim.y;
        const int id = bx + gridDim.x * by;
        const int q = total >> 3, rem = total & 7;
        const int xcd = id & 7, local = id >> 3;
        const int nid = xcd * q + min(xcd, rem) + local;
        bx = nid % (int)gridDim.x;
        by = nid / (int)gridDim.x;
    }
    const int frame = blockIdx.z;
    const int x0 = bx * TW;
    const int y0 = by * p.seg_rows;
    const int nrows = min(p.ny, y0 + p.seg_rows) - y0;
    const int nchunks = (nrows + 2 * R + CH - 1) / CH;
    const int ybase = y0 - R;
    const float *ixf = p.ix + (size_t)frame * p.frame_stride;
    const float *iyf = p.iy + (size_t)frame * p.frame_stride;

    // ---- tile staging: slot i = tid + l*NT of the chunk's 2 x CH x W4 float4 slots (constant over chunks)
    int trow[NL], tlds[NL], txo[NL];
    bool tpl[NL];
#pragma unroll
    for (int l = 0; l < NL; l++) {
        const int i = min(tid + l * NT, G::TILE4 - 1);
        const int pl = i / (CH * W4), rem = i - pl * (CH * W4);
        trow[l] = rem / W4;
        const int q = rem - trow[l] * W4;
        tlds[l] = (pl * CH + trow[l]) * P4 + q;
        tpl[l] = pl != 0;
        // VEC: every slot is one aligned float4 from an in-range address; slots hanging over the left/right image border
        // fetch a neighbouring quad and are rewritten in LDS below.  !VEC: element loads with the reflection applied.
        txo[l] = VEC ? min(max(x0 - HALO + 4 * q, 0), max(p.nx - 4, 0)) : x0 - HALO + 4 * q;
    }
    const bool x_inside = x0 - HALO >= 0 && x0 - HALO + G::W <= p.nx;
    ft_v4f pre[NL];

    auto prefetch = [&](int chunk) __attribute__((always_inline)) {
        const int yc = ybase + chunk * CH;
#pragma unroll
        for (int l = 0; l < NL; l++) {
            if ((l + 1) * NT <= G::TILE4 || tid + l * NT < G::TILE4) {
                const int gy = fir_reflect(yc + trow[l], p.ny);
                const float *rowp = (tpl[l] ? iyf : ixf) + (size_t)gy * p.nx;
                if (VEC) {
                    pre[l] = *reinterpret_cast<const ft_v4f *>(rowp + txo[l]);
                } else {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; e++) v[e] = rowp[fir_reflect(txo[l] + e, p.nx)];
                    pre[l] = ft_v4f{v[0], v[1], v[2], v[3]};
                }
            }
        }
    };
    auto commit = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int l = 0; l < NL; l++)
            if ((l + 1) * NT <= G::TILE4 || tid + l * NT < G::TILE4) reinterpret_cast<ft_v4f *>(raw4)[tlds[l]] = pre[l];
        if (VEC && !x_inside) {
            // border strips only (workgroup-uniform): rebuild the reflected halo columns from the columns of the same
            // LDS row.  left: x = -k -> k;  right: x = nx-1+k -> nx-k  (gaussian.cpp:345-349)
            __syncthreads();
            float *rawf = reinterpret_cast<float *>(raw4);
            for (int i = tid; i < 2 * CH * 2 * HALO; i += NT) {
                const int h = i % (2 * HALO), rr = i / (2 * HALO);  // rr = plane*CH + row
                int c, x;
                if (h < HALO) { c = h; x = x0 - HALO + c; if (x >= 0) continue; }
                else { x = p.nx + (h - HALO); c = x - x0 + HALO; if (c >= G::W) continue; }
                const int sc = fir_reflect(x, p.nx) - x0 + HALO;
                if (sc < 0 || sc >= G::W) continue;
                rawf[rr * P4 * 4 + c] = rawf[rr * P4 * 4 + sc];
            }
        }
    };

    // ---- per-thread roles (the plane is wave-uniform: TW threads = TW/64 waves per plane, one per SIMD for TW = 256)
    const int plane = __builtin_amdgcn_readfirstlane(tid / TW);
    const int idx = tid - plane * TW;
    // row pass: (row of the chunk, 16-pixel strip) per lane.  ds_read_b128 is served in 16-lane groups {0-3,12-15,20-27},
    // {4-11,16-19,28-31} (+32); with an odd slot pitch these mappings give every group 16 distinct slot banks:
    //   TW 128: 8 strips per row, a wave covers 8 rows:  row = lane / 8, strip = lane % 8
    //   TW 256: 16 strips per row, a wave covers 4 rows: row = lane % 4, strip = lane / 4
    const int lane = idx & 63, wvp = idx >> 6;
    const int rr = TW == 128 ? wvp * 8 + (lane >> 3) : wvp * 4 + (lane & 3);
    const int rs = TW == 128 ? (lane & 7) : (lane >> 2);
    const int col = idx;                            // column pass: column of the strip
    const int scol = ft_ring_col(col);
    const int gx = x0 + col;
    float *outp = (plane == 0 ? p.out0 : plane == 1 ? p.out1 : p.out2) + (size_t)frame * p.frame_stride;
    float wo[2 * R];  // the column's last 2R row-filtered values of the previous chunk (kept as floats: 2R registers)
#pragma unroll
    for (int i = 0; i < 2 * R; i++) wo[i] = 0.f;

    prefetch(0);
    commit();
    for (int chunk = 0; chunk < nchunks; chunk++) {
        __syncthreads();  // raw tile of this chunk complete; ring free (column pass / output phase of the previous chunk done)

        // ---- row pass
        if (plane == 0) ft_row_task<R, TW, FMA, 0>(raw4, ring, rr, rs, p.B);
        else if (plane == 1) ft_row_task<R, TW, FMA, 1>(raw4, ring, rr, rs, p.B);
        else ft_row_task<R, TW, FMA, 2>(raw4, ring, rr, rs, p.B);
        __syncthreads();
        // the raw tile is free: fetch the next one into registers now (in flight during the column pass, written to LDS
        // after it -- the loads precede the column pass's stores in the memory queue, so waiting for them does not wait
        // for the stores).  Unconditional: past the last chunk it fetches clamped rows that are never used.
        prefetch(chunk + 1);

        // ---- column pass.  wn[2R + r] = this chunk's row r of the column, wn[0..2R) = the previous chunk's last 2R rows;
        // output row oi_base + r is centred on wn[R + r]; all indices are compile-time constants.  The 16 reads, the
        // conversions and the 16 tap chains carry no control flow, so the chains interleave (a chain alone is
        // latency-bound: 2R+1 dependent f64 operations).
        const float *rcol = ring + plane * CH * RP + scol;
        const int oi_base = chunk * CH - 2 * R;
        float nv[CH];
#pragma unroll
        for (int r = 0; r < CH; r++) nv[r] = rcol[r * RP];
        double wn[CH + 2 * R];
#pragma unroll
        for (int i = 0; i < 2 * R; i++) wn[i] = (double)wo[i];
#pragma unroll
        for (int r = 0; r < CH; r++) wn[2 * R + r] = (double)nv[r];
#pragma unroll
        for (int i = 0; i < 2 * R; i++) wo[i] = nv[CH - 2 * R + i];
        auto tap_chain = [&](int r) __attribute__((always_inline)) -> float {
            const int c = R + r;
            double sum = p.B[0] * wn[c];
#pragma unroll
            for (int j = 1; j <= R; j++) {
                const double pair = wn[c - j] + wn[c + j];
                if (FMA) sum = __builtin_fma(p.B[j], pair, sum);
                else sum += p.B[j] * pair;
            }
            return (float)sum;
        };
        if (OUT == 0) {
            if (gx < p.nx) {  // one exec region around the whole pass: no per-store branches between the chains
                float *dst = outp + (unsigned)(y0 + oi_base) * (unsigned)p.nx + (unsigned)gx;  // one frame < 2^32 px; row oi_base may lie above the segment
                if (oi_base >= 0 && oi_base + CH <= nrows) {  // workgroup-uniform: every row of the chunk is an output row
#pragma unroll
                    for (int r = 0; r < CH; r++) dst[(unsigned)r * (unsigned)p.nx] = tap_chain(r);
                } else {
                    float o[CH];
#pragma unroll
                    for (int r = 0; r < CH; r++) o[r] = tap_chain(r);
#pragma unroll
                    for (int r = 0; r < CH; r++)
                        if (oi_base + r >= 0 && oi_base + r < nrows) outp[(unsigned)(y0 + oi_base + r) * (unsigned)p.nx + (unsigned)gx] = o[r];
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < CH; r++) ring[(plane * CH + r) * RP + scol] = tap_chain(r);  // slot r was read by this very thread
        }
        if (OUT != 0) {
            // ---- output phase: the chunk's smoothed A, B, C rows sit in the ring (slot r <-> output row oi_base + r)
            __syncthreads();
            const float4 *ring4 = reinterpret_cast<const float4 *>(ring);
            constexpr int ROW4 = TW / 4, RP4 = RP / 4;
            if (OUT == 2) {
                for (int i = tid; i < CH * ROW4; i += NT) {
                    const int r = i / ROW4, q = i - r * ROW4;
                    const int oi = oi_base + r, x = x0 + 4 * q;
                    if (oi < 0 || oi >= nrows || x >= p.nx) continue;
                    const int sl = ft_ring_slot4(q >> 2, q & 3);
                    const float4 a = ring4[(0 * CH + r) * RP4 + sl], b = ring4[(1 * CH + r) * RP4 + sl], c = ring4[(2 * CH + r) * RP4 + sl];
                    const ft_v4f v = {harris_response_value<0>(a.x, b.x, c.x, p.k), harris_response_value<0>(a.y, b.y, c.y, p.k),
                                      harris_response_value<0>(a.z, b.z, c.z, p.k), harris_response_value<0>(a.w, b.w, c.w, p.k)};
                    float *dst = p.out0 + (size_t)frame * p.frame_stride + (unsigned)(y0 + oi) * (unsigned)p.nx + (unsigned)x;
                    *reinterpret_cast<ft_v4f *>(dst) = v;
                }
            } else {
                for (int i = tid; i < 3 * CH * ROW4; i += NT) {
                    const int pl = i / (CH * ROW4), rem = i - pl * (CH * ROW4);
                    const int r = rem / ROW4, q = rem - r * ROW4;
                    const int oi = oi_base + r, x = x0 + 4 * q;
                    if (oi < 0 || oi >= nrows || x >= p.nx) continue;
                    const float4 a = ring4[(pl * CH + r) * RP4 + ft_ring_slot4(q >> 2, q & 3)];
                    float *dst = (pl == 0 ? p.out0 : pl == 1 ? p.out1 : p.out2) + (size_t)frame * p.frame_stride +
                                 (unsigned)(y0 + oi) * (unsigned)p.nx + (unsigned)x;
                    *reinterpret_cast<ft_v4f *>(dst) = ft_v4f{a.x, a.y, a.z, a.w};
                }
            }
        }
        commit();  // stage the next chunk's tile; the next row pass reads it (and writes the ring) after the barrier at the top
    }
}

// ------------------------------------------------------------------ host side
bool tensor_fast_path(int R) { return R == 7 || R == 3 || R == 1; }

template <int R, int TW, int OUT>
static imgfd_status launch_tensor_r(imgfd_ctx *ctx, TensorParams &p, int n_frames, bool vec)
{
    using G = TensorGeom<R, TW>;
    const int strips = ceil_div(p.nx, G::TW);
    // Segment length: a workgroup walks (rows + 2R) rows in chunks of CH.  With `slots` workgroups resident on the chip,
    // the pass takes ceil(workgroups / slots) rounds of (chunks per segment) steps: pick the segment count that
    // minimises that product (ties: fewer, longer segments = less halo work).
    static int per_cu = 0;
    if (!per_cu) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void *)fir_tensor<R, TW, true, true, OUT>, G::NT, G::LDS_BYTES) != hipSuccess || n < 1)
            n = 2;
        per_cu = n;
        if (const char *e = getenv("IMGFD_TENSOR_PER_CU")) per_cu = atoi(e) > 0 ? atoi(e) : per_cu;
    }
    const long slots = (long)per_cu * ctx->num_cu;
    long best_cost = -1;
    int seg = p.ny;
    for (int nseg = 1; nseg <= ceil_div(p.ny, G::CH); nseg++) {
        int m = ceil_div(ceil_div(p.ny, nseg) + 2 * R, G::CH);
        if (m < 2) m = 2;
        const int sr = m * G::CH - 2 * R;  // (rows + 2R) fills whole chunks
        const long wgs = (long)strips * ceil_div(p.ny, sr) * n_frames;
        const long cost = ((wgs + slots - 1) / slots) * m;
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; seg = sr; }
    }
    p.seg_rows = seg;
    if (const char *e = getenv("IMGFD_TENSOR_SEG")) if (atoi(e) > 0) p.seg_rows = atoi(e);
    dim3 grid(strips, ceil_div(p.ny, p.seg_rows), n_frames);
    static const char *env = getenv("IMGFD_XCD_REMAP");
    p.xcd_remap = env ? atoi(env) : 1;
    const size_t lds = G::LDS_BYTES;
    auto go = [&](auto kern) -> imgfd_status {
        IMGFD_HIP(ctx, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, grid, dim3(G::NT), lds, ctx->stream, p);
        IMGFD_HIP(ctx, hipGetLastError());
        return IMGFD_OK;
    };
    if (!vec) {  // unaligned planes / rows that are no whole quads: element loads, direct stores (OUT 0 only, checked by the caller)
        if (ctx->fir_mode) return go(fir_tensor<R, TW, true, false, 0>);
        return go(fir_tensor<R, TW, false, false, 0>);
    }
    if (ctx->fir_mode) return go(fir_tensor<R, TW, true, true, OUT>);
    return go(fir_tensor<R, TW, false, true, OUT>);
}

// out_mode 0: A, B, C (direct stores), 1: A, B, C (float4 rows through LDS), 2: Harris response only (d_A receives R).
// Returns IMGFD_ERR_UNSUPPORTED when no specialised kernel serves the radius / alignment (the caller falls back).
imgfd_status launch_tensor_march(imgfd_ctx *ctx, const float *d_Ix, const float *d_Iy, float *d_A, float *d_B, float *d_C,
                                 int nx, int ny, int n_frames, int R, const double *B, float k, int out_mode)
{
    if (!tensor_fast_path(R)) return IMGFD_ERR_UNSUPPORTED;
    TensorParams p;
    memset(&p, 0, sizeof p);
    p.ix = d_Ix; p.iy = d_Iy; p.out0 = d_A; p.out1 = d_B; p.out2 = d_C; p.nx = nx; p.ny = ny;
    p.frame_stride = (long)nx * ny;
    p.k = k;
    memcpy(p.B, B, sizeof(double) * (R + 1));
    // float4 tile loads / row stores need 16-byte aligned planes and whole quads per row (frames are nx*ny floats apart)
    const bool vec = nx % 4 == 0 && nx >= 4 && (size_t)d_Ix % 16 == 0 && (size_t)d_Iy % 16 == 0 && (size_t)d_A % 16 == 0 &&
                     (out_mode == 2 || ((size_t)d_B % 16 == 0 && (size_t)d_C % 16 == 0));
    if (out_mode != 0 && !vec) return IMGFD_ERR_UNSUPPORTED;
    // 256-column strips (12 waves: every SIMD carries three) unless the image is narrow
    static const char *twe = getenv("IMGFD_TENSOR_TW");
    const int tw = twe && atoi(twe) == 128 ? 128 : twe && atoi(twe) == 256 ? 256 : (nx > 384 ? 256 : 128);
#define FT_GO(RR)                                                                                          \
    case RR:                                                                                               \
        if (tw == 256) {                                                                                   \
            if (out_mode == 2) return launch_tensor_r<RR, 256, 2>(ctx, p, n_frames, vec);                  \
            if (out_mode == 1) return launch_tensor_r<RR, 256, 1>(ctx, p, n_frames, vec);                  \
            return launch_tensor_r<RR, 256, 0>(ctx, p, n_frames, vec);                                     \
        }                                                                                                  \
        if (out_mode == 2) return launch_tensor_r<RR, 128, 2>(ctx, p, n_frames, vec);                      \
        if (out_mode == 1) return launch_tensor_r<RR, 128, 1>(ctx, p, n_frames, vec);                      \
        return launch_tensor_r<RR, 128, 0>(ctx, p, n_frames, vec);
    switch (R) {
        FT_GO(7)
        FT_GO(3)
        FT_GO(1)
    }
#undef FT_GO
    return IMGFD_ERR_UNSUPPORTED;
}
