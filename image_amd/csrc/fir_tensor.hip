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
//   * one 768-thread workgroup (12 waves: three per SIMD; multiples of four load the SIMDs evenly) owns a 256-column strip
//     (384 threads / 128 columns for narrow images); four waves work on plane A, four on B, four on C in both passes
//     (wave-uniform).  Workgroups are persistent: one per CU, each walking its share of the (frame, strip, segment) tiles as
//     ONE pipelined sequence of 16-row chunks.
//   * row pass: a thread owns 16 consecutive pixels of one row of its plane: 16+2R window values are read from the raw
//     Ix/Iy tile in LDS with ds_read_b128 (row pitch = odd number of 16-byte slots: every 16-lane group of the b128
//     read pattern hits 16 distinct slot banks), multiplied in float, widened ONCE to double ((16+2R)/16 conversions
//     per pixel) and slid through the 16 outputs in registers; the rounded floats go to an LDS ring [plane][16][TW + 4].
//   * column pass, register-marching: a thread owns ONE column of one plane for the whole tile sequence and keeps the
//     column's last 2R row-filtered floats in registers; per chunk it reads 16 new values (ds_read_b32, lane <-> column:
//     conflict-free), converts the window once and emits 16 output rows.  No window re-reads, no halo rows in LDS.
//   * both passes advance FOUR tap chains together (a dependent f64 instruction issues every ~15 cycles).
//   * per step: barrier 1 -- small phase (ring -> registers, the prefetched Ix/Iy tile of the next chunk -> LDS, finished
//     rows of chunk s-2 -> global, with the corner response and the threshold quads in the response variant) -- barrier 2
//     -- big phase (global fetch of chunk s+1 first, column pass of chunk s-1, row pass of chunk s); border strips rebuild
//     the reflected halo columns inside LDS.
//
// LDS: 85 KB (response variant, 256 columns) to 135 KB (A/B/C doorway with its output buffer): one workgroup per CU.
// 168 VGPRs (amdgpu_waves_per_eu(3, 3)).  HBM traffic is the algorithmic 8 B read + 12 B (or 4.25 B) written per pixel
// plus strip / segment halos (served by L2): measured 1.027 x algorithmic (profiles/r02/c_k3_pmc_summary.txt).
#include "common.h"
#include "fir_device.h"
#include "harris_device.h"
#include "fir_tensor_device.h"

#include <algorithm>
#include <type_traits>

#ifndef FT_PRIO
#define FT_PRIO 2
#endif
#ifndef FT_DHIST
#define FT_DHIST 0  // 1: the column history stays in double (2R conversions fewer per column pass, 2R registers more)
#endif
#ifdef FT_PROFILE
// experiment build only (make EXTRA=-DFT_PROFILE): per-phase shader-clock sums over all waves of fir_tensor
__device__ unsigned long long g_ft_prof[8];
#define FT_T(i)                                                     \
    do {                                                            \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
        if ((threadIdx.x & 63) == 0) prof[i] += t_ - tlast;         \
        tlast = t_;                                                 \
    } while (0)
extern "C" __attribute__((visibility("default"))) int imgfd_debug_ft_profile(unsigned long long *out, int reset)
{
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ft_prof), sizeof(unsigned long long) * 8) != hipSuccess) return 1;
    if (reset) {
        unsigned long long z[8] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_ft_prof), z, sizeof z) != hipSuccess) return 1;
    }
    return 0;
}
#else
#define FT_T(i)
#endif


// s_setprio takes an immediate: the unrolled callers pass compile-time values, this folds to one instruction
__device__ __forceinline__ void ft_setprio(int level)
{
    if (level >= 3) __builtin_amdgcn_s_setprio(3);
    else if (level == 2) __builtin_amdgcn_s_setprio(2);
    else if (level == 1) __builtin_amdgcn_s_setprio(1);
    else __builtin_amdgcn_s_setprio(0);
}

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
    static constexpr size_t lds_bytes(int out) { return sizeof(float4) * 2 * CH * P4 + sizeof(float) * 3 * CH * RPITCH * (out ? 2 : 1); }
    static_assert(TW == 128 || TW == 256, "strip widths with a conflict-free lane mapping");
    static_assert(2 * R <= CH, "the column pass reaches 2R rows back into the previous chunk");
    static_assert(16 * (NS - 1) + 4 * NW4 <= W, "row-pass window reads stay inside the tile row");
};


// ring position of column c of a row: the four float4 slots of each 16-column strip are rotated by (strip / 2), so the
// row pass's ds_write_b128 (8 lanes = 8 strips of one row) covers 8 distinct slots mod 8 and the column pass's
// ds_read_b32 still reads 64 consecutive dwords per wave in some order
__device__ __forceinline__ int ft_ring_slot4(int s, int h) { return 4 * s + ((h + (s >> 1)) & 3); }
__device__ __forceinline__ int ft_ring_col(int c) { return 4 * ft_ring_slot4(c >> 4, (c >> 2) & 3) + (c & 3); }


// row pass, first half: the 16+2R float products of one (row, strip) of plane PL from the raw Ix/Iy tile
template <int R, int TW, int PL>
__device__ __forceinline__ void ft_row_products(const float4 *raw4, int r, int s, float (&pr)[TensorGeom<R, TW>::NW])
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
#pragma unroll
    for (int k = 0; k < NW; k++) {
        if (PL == 0) pr[k] = wx[OFF + k] * wx[OFF + k];       // harris.cpp:59
        else if (PL == 1) pr[k] = wx[OFF + k] * wy[OFF + k];  // harris.cpp:60
        else pr[k] = wy[OFF + k] * wy[OFF + k];               // harris.cpp:61
    }
}


// Register budget = what the wave placement needs, not what the occupancy API answers.  A workgroup's waves are dealt to
// the CU's four SIMDs in turn, so only multiples of 4 waves load the SIMDs evenly:
//   TW 256: 12 waves = 3 per SIMD, one workgroup per CU (85 / 135 KB of LDS): up to 168 VGPRs.
//   TW 128 (narrow images only): 6 waves land 2,2,1,1 -- the SIMDs that hold two waves of a workgroup set the pace of its
//           barriers, 75 % of the f64 rate at best (measured: 48 us per 4K frame against 44); with 128 VGPRs two such
//           workgroups share a CU but the allocator spills, with 160 the hardware admits ONE per CU whatever the
//           occupancy API answers (1.4 resident waves per SIMD measured).  Same 168-register budget, no spills.
#define FT_WAVES_PER_EU(tw) IMGFD_WAVES_PER_EU(3, 3)
// OUT 0: A, B, C stored straight from the column pass (one dword per lane and row, 256 B per wave)
// OUT 1: A, B, C staged through LDS and stored as float4 rows
// OUT 2: corner response (Harris measure) computed from the staged A, B, C; only R is stored (float4 rows)
//
// Schedule.  The workgroup is persistent: the batch's (frame, strip) columns form ONE line of 16-row chunk units, worker w owns
// the units [w * units_per_worker, (w + 1) * units_per_worker) and marches them as ONE sequence of steps (a step = one chunk;
// one warm-up chunk where its share begins inside a column), so consecutive columns overlap like consecutive chunks do and
// no CU idles between workgroups.  Step s, pipelined over three consecutive chunks:
//     barrier 1
//       small phase: every thread pulls its column's 16 row-filtered values of chunk s-1 out of the ring into registers,
//                    writes the Ix/Iy tile of chunk s (fetched into registers during step s-1) to LDS, and -- OUT 1/2 --
//                    stores the finished output rows of chunk s-2 from the output buffer
//     barrier 2
//       big phase:   row pass of chunk s (raw tile -> ring), global fetch of chunk s+1's tile into registers, column pass
//                    of chunk s-1 from registers (-> global, or -> output buffer)
// Everything that costs f64 issue slots sits in the big phase, where the 12 waves (three per SIMD) run ~600 independent
// VALU instructions each without meeting a barrier.
struct TensorPos {  // a step of a worker's sequence: chunk `chunk` of the segment rows [y0, y0 + nrows) of (strip, frame); frame >= n_frames: past the end
    int strip, frame, y0, nrows, chunk;
};

template <int R, int TW, bool FMA, bool VEC, int OUT>
__global__ void __launch_bounds__(3 * TW) FT_WAVES_PER_EU(TW) fir_tensor(TensorParams p)
{
    using G = TensorGeom<R, TW>;
    constexpr int CH = G::CH, NT = G::NT, W4 = G::W4, P4 = G::P4, HALO = G::HALO, NL = G::NL, RP = G::RPITCH;

    HIP_DYNAMIC_SHARED(float4, smem4)
    float4 *raw4 = smem4;                                              // [2][CH][P4]
    float *ring = reinterpret_cast<float *>(smem4 + 2 * CH * P4);      // [3][CH][RP], columns permuted per strip
    float *obuf = ring + 3 * CH * RP;                                  // [3][CH][RP] (OUT 1/2), same column permutation

    const int tid = threadIdx.x;
    // worker index: workgroup ids are dealt round-robin to the 8 XCDs; let XCD x own a contiguous run of workers, so that
    // at any time the tiles in flight on one XCD are neighbours (shared halo columns / rows are re-read from its own L2)
    int worker = blockIdx.x;
    const int workers = gridDim.x;
    if (p.xcd_remap) {
        const int q = workers >> 3, rem = workers & 7;
        const int xcd = worker & 7, local = worker >> 3;
        worker = xcd * q + min(xcd, rem) + local;
    }
    // geometry of a position, recomputed where needed (a handful of scalar operations; keeping it per pipeline stage in
    // registers spilled the scalar file)
#define FT_X0(t) ((t).strip * TW)
#define FT_Y0(t) ((t).y0)
#define FT_NROWS(t) ((t).nrows)
#define FT_NCHUNKS(t) ((FT_NROWS(t) + 2 * R + CH - 1) / CH)
    // Work split.  A (frame, strip) column of ny rows is C = ceil((ny + 2R) / CH) chunks when one worker marches down all of
    // it; the columns of the batch, strip fastest, form ONE line of n_frames * nstrips * C chunk units, and worker w owns units
    // [w * S, (w + 1) * S) of it.  A share that begins k units into a column begins at output row CH * k - 2R -- the row the
    // column's chunk k would begin with -- and pays one chunk of warm-up for it, so no worker runs more than S + 1 steps.
    // (Fixed-size tiles dealt round-robin meant 15 tiles = 15 warm-ups per worker for the 480 columns of a 32-frame 4K batch
    // on 256 workers: 270 steps where this split takes 256.)
    long lin = (long)worker * p.units_per_worker;  // first chunk unit of the worker's next segment
    const long lin_end = min(lin + p.units_per_worker, p.total_units);
    auto next_segment = [&]() __attribute__((always_inline)) -> TensorPos {
        TensorPos n;
        n.chunk = 0;
        if (lin >= lin_end) { n.strip = 0; n.frame = p.n_frames; n.y0 = 0; n.nrows = 0; return n; }
        const long column = lin / p.units_per_column;
        const int k = (int)(lin - column * p.units_per_column);
        const int k_end = (int)min((long)p.units_per_column, lin_end - column * p.units_per_column);
        n.frame = (int)(column / p.nstrips);
        n.strip = (int)(column - (long)n.frame * p.nstrips);
        n.y0 = k == 0 ? 0 : CH * k - 2 * R;
        n.nrows = (k_end == p.units_per_column ? p.ny : CH * k_end - 2 * R) - n.y0;
        lin += k_end - k;
        return n;
    };
    // the step after `t`: next chunk, or chunk 0 of the worker's next segment
    auto next_pos = [&](const TensorPos &t) __attribute__((always_inline)) -> TensorPos {
        if (t.chunk + 1 < FT_NCHUNKS(t)) { TensorPos n = t; n.chunk++; return n; }
        return next_segment();
    };

    // ---- tile staging: slot i = tid + l*NT of a chunk's 2 x CH x W4 float4 slots (plane, row, quad); recomputed where
    // needed rather than kept in registers across the big phase
    auto stage_slot = [&](int l, int &pl, int &row, int &q) __attribute__((always_inline)) {
        const int i = min(tid + l * NT, G::TILE4 - 1);
        pl = i / (CH * W4);
        const int rem = i - pl * (CH * W4);
        row = rem / W4;
        q = rem - row * W4;
    };
    ft_v4f pre[NL];
    // VEC: every slot is fetched as ONE aligned float4 from an in-range address (slots hanging over the left/right image
    // border fetch a neighbouring quad and are rebuilt in LDS by patch_borders()): a straight-line sequence of loads, so
    // they stay asynchronous and the waitcnt bookkeeping stays exact.  !VEC (unaligned planes, rows that are no whole
    // quads): element loads with the reflection of gaussian.cpp:345-349 applied.
    auto prefetch = [&](const TensorPos &t) __attribute__((always_inline)) {
        const int yc = FT_Y0(t) - R + t.chunk * CH, tx0 = FT_X0(t);
        const float *ixf = p.ix + (size_t)t.frame * p.frame_stride;
        const float *iyf = p.iy + (size_t)t.frame * p.frame_stride;
#pragma unroll
        for (int l = 0; l < NL; l++) {
            if ((l + 1) * NT <= G::TILE4 || tid + l * NT < G::TILE4) {
                int pl, row, q;
                stage_slot(l, pl, row, q);
                const int gy = fir_reflect(yc + row, p.ny);
                const float *rowp = (pl ? iyf : ixf) + (size_t)gy * p.nx;
                const int xo = tx0 - HALO + 4 * q;
                if (VEC) {
                    pre[l] = *reinterpret_cast<const ft_v4f *>(rowp + min(max(xo, 0), max(p.nx - 4, 0)));
                } else {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; e++) v[e] = rowp[fir_reflect(xo + e, p.nx)];
                    pre[l] = ft_v4f{v[0], v[1], v[2], v[3]};
                }
            }
        }
    };
    auto commit = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int l = 0; l < NL; l++)
            if ((l + 1) * NT <= G::TILE4 || tid + l * NT < G::TILE4) {
                int pl, row, q;
                stage_slot(l, pl, row, q);
                reinterpret_cast<ft_v4f *>(raw4)[(pl * CH + row) * P4 + q] = pre[l];
            }
    };
    // border strips only (workgroup-uniform): rebuild the reflected halo columns from the columns of the same LDS row.
    // left: x = -k -> k;  right: x = nx-1+k -> nx-k  (gaussian.cpp:345-349)
    auto patch_borders = [&](const TensorPos &t) __attribute__((always_inline)) {
        float *rawf = reinterpret_cast<float *>(raw4);
        const int tx0 = FT_X0(t);
        for (int i = tid; i < 2 * CH * 2 * HALO; i += NT) {
            const int h = i % (2 * HALO), rr_ = i / (2 * HALO);  // rr_ = plane*CH + row
            int c, x;
            if (h < HALO) { c = h; x = tx0 - HALO + c; if (x >= 0) continue; }
            else { x = p.nx + (h - HALO); c = x - tx0 + HALO; if (c >= G::W) continue; }
            const int sc = fir_reflect(x, p.nx) - tx0 + HALO;
            if (sc < 0 || sc >= G::W) continue;
            rawf[rr_ * P4 * 4 + c] = rawf[rr_ * P4 * 4 + sc];
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
    float *const outp = plane == 0 ? p.out0 : plane == 1 ? p.out1 : p.out2;
    using hist_t = std::conditional_t<FT_DHIST != 0, double, float>;
    hist_t wo[2 * R];  // the column's last 2R row-filtered values of the previous chunk
#pragma unroll
    for (int i = 0; i < 2 * R; i++) wo[i] = 0;
    float nv[CH];
#pragma unroll
    for (int i = 0; i < CH; i++) nv[i] = 0.f;

    // output phase (OUT 1/2): rows of the chunk at `t` from the output buffer to global memory as float4s
    auto output_rows = [&](const TensorPos &t) __attribute__((always_inline)) {
        const float4 *ob4 = reinterpret_cast<const float4 *>(obuf);
        constexpr int ROW4 = TW / 4, RP4 = RP / 4;
        const int oi_base = t.chunk * CH - 2 * R, tx0 = FT_X0(t), ty0 = FT_Y0(t), tnrows = FT_NROWS(t);
        if (OUT == 2) {
            for (int i = tid; i < CH * ROW4; i += NT) {
                const int r = i / ROW4, q = i - r * ROW4;
                const int oi = oi_base + r, x = tx0 + 4 * q;
                if (oi < 0 || oi >= tnrows || x >= p.nx) continue;
                const int sl = ft_ring_slot4(q >> 2, q & 3);
                const float4 a = ob4[(0 * CH + r) * RP4 + sl], b = ob4[(1 * CH + r) * RP4 + sl], c = ob4[(2 * CH + r) * RP4 + sl];
                const ft_v4f v = {harris_response_value<0>(a.x, b.x, c.x, p.k), harris_response_value<0>(a.y, b.y, c.y, p.k),
                                  harris_response_value<0>(a.z, b.z, c.z, p.k), harris_response_value<0>(a.w, b.w, c.w, p.k)};
                const size_t px = (size_t)t.frame * p.frame_stride + (unsigned)(ty0 + oi) * (unsigned)p.nx + (unsigned)x;
                *reinterpret_cast<ft_v4f *>(p.out0 + px) = v;
                if (p.tq) p.tq[px >> 2] = (unsigned char)harris_quad_bits(v[0], v[1], v[2], v[3], p.Th);  // kernel-uniform
            }
        } else {
            for (int i = tid; i < 3 * CH * ROW4; i += NT) {
                const int pl = i / (CH * ROW4), rem = i - pl * (CH * ROW4);
                const int r = rem / ROW4, q = rem - r * ROW4;
                const int oi = oi_base + r, x = tx0 + 4 * q;
                if (oi < 0 || oi >= tnrows || x >= p.nx) continue;
                const float4 a = ob4[(pl * CH + r) * RP4 + ft_ring_slot4(q >> 2, q & 3)];
                float *dst = (pl == 0 ? p.out0 : pl == 1 ? p.out1 : p.out2) + (size_t)t.frame * p.frame_stride +
                             (unsigned)(ty0 + oi) * (unsigned)p.nx + (unsigned)x;
                *reinterpret_cast<ft_v4f *>(dst) = ft_v4f{a.x, a.y, a.z, a.w};
            }
        }
    };

    // The two passes of the big phase, in groups of ILP outputs whose tap chains advance together.
    //   DO_COL: column pass of the chunk at `tp` (nv[] = its 16 rows of this thread's column, wo[] = the 2R rows before
    //           them -> 16 output rows).  FULL (workgroup-uniform): every row of the chunk is an output row and the strip
    //           lies inside the image, so the stores need no predicate; otherwise each store is guarded.
    //   DO_ROW: row pass of the current chunk (this thread's 16+2R products pr[] -> 16 row-filtered floats in the ring).
    // (Interleaving the groups of the two passes, so that the stores trickle out over the whole phase, was tried: the two
    // windows together do not fit the 168-register budget -- spills, 116 us per frame.  So was pinning the order of the
    // groups with scheduling fences: +10 %.)
    auto big_phase = [&](auto do_row_tag, auto do_col_tag, auto full_tag, const TensorPos &tp, float (&pr)[G::NW])
                         __attribute__((always_inline)) {
        constexpr bool DO_ROW = decltype(do_row_tag)::value, DO_COL = decltype(do_col_tag)::value, FULL = decltype(full_tag)::value;
        constexpr int ILP = FT_ILP, NG = CH / ILP;
        float cw[CH + 2 * R];  // column window: the previous chunk's last 2R rows, then this chunk's 16
        double dcw[CH + 2 * R], dpr[G::NW];
        const int oi_base = tp.chunk * CH - 2 * R, tnrows = FT_NROWS(tp);
        const int gx = FT_X0(tp) + col;
        // output stores: buffer addressing -- the frame's plane as the resource, the row as a SCALAR byte offset, the column
        // as the lane's constant offset: no per-store 64-bit address arithmetic on the vector pipe (one frame < 2^32 bytes)
        const FtBuffer cbuf = ft_make_buffer(outp + (size_t)tp.frame * p.frame_stride, (unsigned)p.nx * (unsigned)p.ny * 4u);
        const unsigned crow0 = (unsigned)(FT_Y0(tp) + oi_base) * (unsigned)p.nx * 4u;  // wraps for rows above the segment: those stores are guarded
        if (DO_COL) {
#pragma unroll
            for (int i = 0; i < 2 * R; i++) {
                if (FT_DHIST) { dcw[i] = wo[i]; cw[i] = 0.f; }
                else cw[i] = (float)wo[i];
            }
#pragma unroll
            for (int r = 0; r < CH; r++) cw[2 * R + r] = nv[r];
            if (!FT_DHIST) {
#pragma unroll
                for (int i = 0; i < 2 * R; i++) wo[i] = nv[CH - 2 * R + i];
            }
        }
        float4 *rdst = reinterpret_cast<float4 *>(ring + (plane * CH + rr) * RP);
#pragma unroll
        for (int g = 0; g < NG; g++) {
            // Wave priority.  The SIMD's arbiter serves priority first, then age: left alone, the oldest of a SIMD's three waves runs
            // the big phase at full speed and idles at the barrier while the youngest finishes alone (FT_PROFILE, round 4: 22 % of
            // the wave cycles waited at barrier 1).  FT_PRIO 1 lets every wave lower its own priority as it advances (column
            // groups 3, 3, 2, 2, row groups 1, 1, 0, 0): the barrier wait falls to 13 % -- and the row pass grows from 15 % to 29 %
            // of the wave cycles: the time is conserved, because the f64 pipe is busy either way (profiles/r04/k3_phase_split.txt,
            // ubench7.txt: 26-28 T lane-op/s is what three waves per SIMD issue of ANY f64 instruction; the kernel runs 23.7).
            // FT_PRIO 2 (kept, -2 %): column pass at 1, row pass at 0 -- a wave still storing outranks the ones already in their
            // row pass, so the stores of the chunk leave early.
            if (FT_PRIO == 1) ft_setprio((DO_COL ? 3 : 1) - (2 * g) / NG);
            else if (FT_PRIO == 2 && g == 0) ft_setprio(DO_COL ? 1 : 0);
            if (DO_COL) {
                float o[ILP];
                ft_group<R, FMA, ILP, CH + 2 * R, FT_DHIST ? 2 * R : 0>(cw, dcw, ILP * g, p.B, o);
#pragma unroll
                for (int e = 0; e < ILP; e++) {
                    const int r = ILP * g + e;
                    if (OUT != 0) obuf[(plane * CH + r) * RP + scol] = o[e];
                    else if (FULL) ft_buffer_store(cbuf, (unsigned)gx * 4u, crow0 + (unsigned)r * (unsigned)p.nx * 4u, o[e]);
                    else if (gx < p.nx && oi_base + r >= 0 && oi_base + r < tnrows)
                        ft_buffer_store(cbuf, (unsigned)gx * 4u, crow0 + (unsigned)r * (unsigned)p.nx * 4u, o[e]);
                }
            }
            if (DO_ROW) {
                float o[ILP];
                ft_group<R, FMA, ILP, G::NW>(pr, dpr, ILP * g, p.B, o);
#pragma unroll
                for (int h = 0; h < ILP / 4; h++) rdst[ft_ring_slot4(rs, g * (ILP / 4) + h)] = make_float4(o[4 * h], o[4 * h + 1], o[4 * h + 2], o[4 * h + 3]);
            }
        }
        if (DO_COL && FT_DHIST) {
#pragma unroll
            for (int i = 0; i < 2 * R; i++) wo[i] = dcw[CH + i];
        }
    };

    TensorPos cur = next_segment();  // chunk of the row pass in this step
    if (cur.frame >= p.n_frames) return;  // workgroup-uniform (the host launches no worker without rows)
    TensorPos prev = cur, prev2 = cur;    // chunk of the column pass / of the output phase in this step
    bool have_cur = true, have_prev = false, have_prev2 = false;
#ifdef FT_PROFILE
    unsigned long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tlast = __builtin_amdgcn_s_memtime();
#endif
    prefetch(cur);
    while (have_cur || have_prev || (OUT != 0 && have_prev2)) {
        FT_T(0);
        __syncthreads();  // barrier 1: ring = row-filtered chunk `prev`, output buffer = chunk `prev2`, raw tile consumed
        FT_T(1);
        if (have_prev) {
            const float *rcol = ring + plane * CH * RP + scol;
#pragma unroll
            for (int r = 0; r < CH; r++) nv[r] = rcol[r * RP];
        }
        if (have_cur) {
            commit();
            if (VEC && !(FT_X0(cur) - HALO >= 0 && FT_X0(cur) - HALO + G::W <= p.nx)) {
                __syncthreads();
                patch_borders(cur);
            }
        }
        if (OUT != 0 && have_prev2) output_rows(prev2);
        FT_T(2);
        __syncthreads();  // barrier 2: ring and output buffer may be overwritten, raw tile of `cur` complete
        // big phase.  Order: the fetch of the next tile first (it lands while the phase computes), then the column pass --
        // its stores are then old by the time the next small phase waits for the fetched tile -- then the row pass.
        FT_T(3);
        TensorPos nxt = cur;
        bool have_nxt = false;
        if (have_cur) {
            nxt = next_pos(cur);
            have_nxt = nxt.frame < p.n_frames;
        }
        FT_T(4);
        float pr[G::NW];
        {
            using T = std::true_type;
            using F = std::false_type;
            const int oib = prev.chunk * CH - 2 * R;
            const bool full = OUT != 0 || (oib >= 0 && oib + CH <= FT_NROWS(prev) && FT_X0(prev) + TW <= p.nx);
            // column pass first (its stores drain while the row pass computes), then the row pass
            if (have_nxt) prefetch(nxt);  // lands while the phase computes; issued later it was not there in time (+5..9 %)
            if (have_prev) {
                if (full) big_phase(F(), T(), T(), prev, pr);
                else big_phase(F(), T(), F(), prev, pr);
            }
            FT_T(5);
            if (have_cur) {
                if (plane == 0) ft_row_products<R, TW, 0>(raw4, rr, rs, pr);
                else if (plane == 1) ft_row_products<R, TW, 1>(raw4, rr, rs, pr);
                else ft_row_products<R, TW, 2>(raw4, rr, rs, pr);
                big_phase(T(), F(), T(), prev, pr);
            }
        }
        FT_T(6);
        prev2 = prev; have_prev2 = have_prev;
        prev = cur; have_prev = have_cur;
        cur = nxt; have_cur = have_nxt;
    }
#ifdef FT_PROFILE
    if ((threadIdx.x & 63) == 0)
        for (int i = 0; i < 8; i++) atomicAdd(&g_ft_prof[i], prof[i]);
#endif
}

// ------------------------------------------------------------------ host side
bool tensor_fast_path(int R) { return R == 7 || R == 3 || R == 1; }

template <int R, int TW, int OUT>
static imgfd_status launch_tensor_r(imgfd_ctx *ctx, TensorParams &p, int n_frames, bool vec)
{
    using G = TensorGeom<R, TW>;
    const int strips = ceil_div(p.nx, G::TW);
    const size_t lds = G::lds_bytes(OUT);
    // Workers = the workgroups the chip holds at once (persistent), each with an equal share of the line of chunks (see the
    // kernel).  One persistent workgroup per CU: 12 waves at 168 registers fill a CU's register file, and the hardware
    // admitted one workgroup even where LDS (68 KB for the 128-column instance) and the occupancy API promised two
    // (profiles/r02/k3_log.txt).
    const int per_cu = 1;
    const long slots = (long)per_cu * ctx->num_cu;
    p.nstrips = strips;
    p.n_frames = n_frames;
    p.units_per_column = ceil_div(p.ny + 2 * R, G::CH);
    p.total_units = (long)strips * n_frames * p.units_per_column;
    long workers = std::min<long>(p.total_units, slots);
    if (ctx->tune.tensor_workers > 0) workers = std::min<long>(p.total_units, ctx->tune.tensor_workers);  // tests: few workers, several segments each
    p.units_per_worker = (p.total_units + workers - 1) / workers;
    workers = (p.total_units + p.units_per_worker - 1) / p.units_per_worker;
    dim3 grid((unsigned)workers);
    p.xcd_remap = 1;
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

// out_mode 0: A, B, C (direct stores), 2: Harris response only (d_A receives R).  (OUT 1 of the kernel -- A, B, C as float4
// rows through LDS -- measured no faster than the direct stores and is not instantiated.)
// Returns IMGFD_ERR_UNSUPPORTED when no specialised kernel serves the radius / alignment (the caller falls back).
imgfd_status launch_tensor_march(imgfd_ctx *ctx, const float *d_Ix, const float *d_Iy, float *d_A, float *d_B, float *d_C,
                                 int nx, int ny, int n_frames, int R, const double *B, float k, int out_mode,
                                 unsigned char *d_tq, float Th)
{
    if (!tensor_fast_path(R)) return IMGFD_ERR_UNSUPPORTED;
    // the kernel addresses a frame's f32 plane with 32-bit BYTE offsets (buffer resources, row offsets in scalar registers): a
    // plane of 4 GiB or more is not its business (frame_fits admits up to 2^31 - 2^24 ELEMENTS; the caller falls back to the
    // generic two-pass FIR, which indexes with size_t)
    if ((size_t)nx * (size_t)ny * sizeof(float) >= ((size_t)1 << 32)) return IMGFD_ERR_UNSUPPORTED;
    TensorParams p;
    memset(&p, 0, sizeof p);
    p.ix = d_Ix; p.iy = d_Iy; p.out0 = d_A; p.out1 = d_B; p.out2 = d_C; p.nx = nx; p.ny = ny;
    p.frame_stride = (long)nx * ny;
    p.k = k;
    p.tq = out_mode == 2 ? d_tq : nullptr; p.Th = Th;
    memcpy(p.B, B, sizeof(double) * (R + 1));
    // float4 tile loads / row stores need 16-byte aligned planes and whole quads per row (frames are nx*ny floats apart)
    const bool vec = nx % 4 == 0 && nx >= 4 && (size_t)d_Ix % 16 == 0 && (size_t)d_Iy % 16 == 0 && (size_t)d_A % 16 == 0 &&
                     (out_mode == 2 || ((size_t)d_B % 16 == 0 && (size_t)d_C % 16 == 0));
    if (out_mode != 0 && out_mode != 2) return IMGFD_ERR_UNSUPPORTED;
    if (out_mode != 0 && !vec) return IMGFD_ERR_UNSUPPORTED;
    // 256-column strips (12 waves: every SIMD carries three) unless the image is narrow
    const int tw = nx > 384 ? 256 : 128;
#define FT_GO(RR)                                                                                          \
    case RR:                                                                                               \
        if (tw == 256) {                                                                                   \
            if (out_mode == 2) return launch_tensor_r<RR, 256, 2>(ctx, p, n_frames, vec);                  \
            return launch_tensor_r<RR, 256, 0>(ctx, p, n_frames, vec);                                     \
        }                                                                                                  \
        if (out_mode == 2) return launch_tensor_r<RR, 128, 2>(ctx, p, n_frames, vec);                      \
        return launch_tensor_r<RR, 128, 0>(ctx, p, n_frames, vec);
    switch (R) {
        FT_GO(7)
        FT_GO(3)
        FT_GO(1)
    }
#undef FT_GO
    return IMGFD_ERR_UNSUPPORTED;
}
