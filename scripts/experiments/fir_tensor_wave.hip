// image_amd/csrc/fir_tensor_wave.hip -- the Harris structure-tensor pass (K3), wave-autonomous form, for gfx950.
//
// Same arithmetic as fir_tensor.hip (compute_autocorrelation_matrix(), image.CornerDetectionHarris/src/harris.cpp:44-70:
// float products Ix*Ix, Ix*Iy, Iy*Iy, :57-62, then per plane a horizontal and a vertical 1-D Gaussian pass, each
// accumulated in double in the reference's order with ONE rounding to float per pass, gaussian.cpp:351-359, 382-390;
// borders :345-349; OUT = 2: the Harris measure of harris.cpp:100-103 on the smoothed planes, only R is written).
//
// What differs is who waits for whom.  fir_tensor.hip runs 12 waves per CU in lock step: two workgroup barriers per
// 16-row chunk, and the counters say a third of the wave cycles wait there (profiles/r03/e_k3_pmc_summary.txt:
// SQ_WAIT_INST_ANY 32 %, VALU issue 0.68).  Here NO wave ever waits for another wave:
//
//   * a wave owns a 64-column strip of one (frame, segment) tile and marches down it in 16-row chunks, doing all three
//     planes itself, one after the other: row pass of the plane (lane = (row, 16-pixel piece): 16 rows x 4 pieces) ->
//     the wave's own 16 x 64 ring in LDS -> column pass (lane = column, the column's last 2R row-filtered values of
//     each plane stay in registers).  The ring is written and read by the same wave, so program order is the only
//     synchronisation: no s_barrier, no flags, no spinning.
//   * the raw Ix/Iy tile of the NEXT chunk is fetched into registers (five 16-byte buffer loads per plane and lane:
//     lane = (row, quarter of the row)) while the current chunk is computed, and written to the wave's own raw tile in
//     LDS at the top of the next step.
//   * eight waves per CU (two per SIMD: amdgpu_waves_per_eu(2, 2), up to 256 registers each -- the three planes'
//     column histories, the register window of the pass and the prefetched tile live side by side), 15 KB of LDS per
//     wave.  The two waves of a SIMD drift apart by themselves: while one reads LDS or stores, the other issues f64.
//   * the response variant needs no exchange either: the lane that owns column c holds A, B and C of its 16 rows in
//     registers when the third column pass ends, evaluates R there and stores it (one dword per lane and row, 256 B per
//     wave and row); the threshold quads come from two quad-permute DPP moves, three compares and a ballot per row.
//
// Workers are waves: worker w walks tiles w, w + workers, ... of the (strip, segment, frame) list as ONE sequence of
// chunks; the waves of a workgroup start on neighbouring strips (their halo columns meet in the CU's L1 / the XCD's L2).
#include "common.h"
#include "fir_tensor_device.h"

#include <algorithm>
#include <type_traits>

#ifndef FTW_ILP
#define FTW_ILP 4
#endif
#ifndef FTW_WAVES
#define FTW_WAVES 8
#endif
#ifndef FTW_DHIST
#define FTW_DHIST 0  // 1: the column histories stay in double (2R conversions fewer per column pass, 2R registers more per plane)
#endif

template <int R>
struct WaveGeom {
    static constexpr int TW = 64, CH = 16, PX = 16, WAVES = FTW_WAVES, NT = 64 * WAVES;
    static constexpr int HALO = (R + 3) / 4 * 4;     // tile halo in whole float4 slots: x0 - HALO is 16-byte aligned
    static constexpr int W = TW + 2 * HALO, W4 = W / 4;
    static constexpr int P4 = W4 | 1;                // raw row pitch in float4 slots, odd: with lane = (row = lane / 4, piece = lane % 4) every
                                                     // 16-lane group of the ds_read_b128 pattern sees 16 distinct slot banks
    static constexpr int OFF = HALO - R;             // window start inside a piece's first slot
    static constexpr int NW = PX + 2 * R;            // window length
    static constexpr int NW4 = (OFF + NW + 3) / 4;
    static constexpr int NLQ = (W4 + 3) / 4;         // slots a lane fetches per plane: lane = (row, quarter), slots quarter * NLQ + l
    static constexpr int RP = TW + 4;                // ring row pitch in floats (17 slots: odd)
    static constexpr int RAW4 = 2 * CH * P4;         // float4 slots of a wave's raw tile (Ix rows, then Iy rows)
    static constexpr int WAVE4 = RAW4 + CH * RP / 4; // float4 slots of LDS per wave (raw tile + ring)
    static constexpr size_t lds_bytes = sizeof(float4) * WAVE4 * WAVES;
    static_assert(2 * R <= CH, "the column pass reaches 2R rows back into the previous chunk");
    static_assert(4 * 3 + NW4 <= W4, "row-pass window reads stay inside the tile row");
    static_assert(4 * NLQ >= W4, "the four quarters cover the row");
    static_assert(FTW_ILP % 4 == 0, "the row pass stores its groups as float4s");
};

// the waves of a workgroup never meet.  Inside a wave the LDS pipe executes in program order; this keeps the compiler from
// moving one lane's read over another lane's write (wavefront-scope fences emit no instruction)
__device__ __forceinline__ void ftw_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct WavePos {  // chunk `chunk` of tile `tile` = (strip, seg, frame), strip fastest
    long tile;
    int strip, seg, frame, chunk;
};

// row pass, first half: the 16+2R float products of one (row, piece) of plane PL from the wave's raw Ix/Iy tile
template <int R, int PL>
__device__ __forceinline__ void ftw_row_products(const float4 *raw4, int r, int s, float (&pr)[WaveGeom<R>::NW])
{
    using G = WaveGeom<R>;
    constexpr int NW = G::NW, NW4 = G::NW4, OFF = G::OFF;
    float wx[NW4 * 4], wy[NW4 * 4];
    // volatile LDS pointers: each window slot stays ONE ds_read_b128 (see fir_tensor.hip)
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

template <int R, bool FMA, int OUT>
__global__ void __launch_bounds__(64 * FTW_WAVES) IMGFD_WAVES_PER_EU(FTW_WAVES / 4, FTW_WAVES / 4) fir_tensor_wave(TensorParams p)
{
    using G = WaveGeom<R>;
    constexpr int CH = G::CH, P4 = G::P4, HALO = G::HALO, NLQ = G::NLQ, RP = G::RP, TW = G::TW, ILP = FTW_ILP, NG = CH / ILP;

    HIP_DYNAMIC_SHARED(float4, smem4)
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    float4 *raw4 = smem4 + wave * G::WAVE4;                            // [2][CH][P4]
    float *ring = reinterpret_cast<float *>(raw4 + G::RAW4);          // [CH][RP]

    // worker = wave.  Workgroup ids are dealt round-robin to the 8 XCDs; let XCD x own a contiguous run of workgroups, so
    // that the tiles in flight on one XCD are neighbours (shared halo columns / rows come from its own L2)
    int wg = blockIdx.x;
    const int wgs = gridDim.x;
    if (p.xcd_remap) {
        const int q = wgs >> 3, rem = wgs & 7;
        const int xcd = wg & 7, local = wg >> 3;
        wg = xcd * q + min(xcd, rem) + local;
    }
    const long workers = (long)wgs * G::WAVES;
    const long tiles = (long)p.nstrips * p.nseg * p.n_frames;

    auto at_tile = [&](long tile) __attribute__((always_inline)) -> WavePos {
        WavePos t;
        t.tile = tile;
        const long rest = tile / p.nstrips;
        t.strip = (int)(tile - rest * p.nstrips);
        t.frame = (int)(rest / p.nseg);
        t.seg = (int)(rest - (long)t.frame * p.nseg);
        t.chunk = 0;
        return t;
    };
#define FTW_X0(t) ((t).strip * TW)
#define FTW_Y0(t) ((t).seg * p.seg_rows)
#define FTW_NROWS(t) (min(p.ny, FTW_Y0(t) + p.seg_rows) - FTW_Y0(t))
#define FTW_NCHUNKS(t) ((FTW_NROWS(t) + 2 * R + CH - 1) / CH)

    // ---- tile staging: lane = (row of the chunk, quarter of the row); slots quarter * NLQ + l of both planes
    const int frow = lane >> 2, fq = (lane & 3) * NLQ;
    ft_v4f pre[2][NLQ];
    auto prefetch = [&](const WavePos &t) __attribute__((always_inline)) {
        const int yc = FTW_Y0(t) - R + t.chunk * CH, tx0 = FTW_X0(t);
        const unsigned plane_bytes = (unsigned)p.nx * (unsigned)p.ny * 4u;
        const FtBuffer bx = ft_make_buffer(const_cast<float *>(p.ix) + (size_t)t.frame * p.frame_stride, plane_bytes);
        const FtBuffer by = ft_make_buffer(const_cast<float *>(p.iy) + (size_t)t.frame * p.frame_stride, plane_bytes);
        const int gy = fir_reflect(yc + frow, p.ny);
        const int rowbase = gy * p.nx;
#pragma unroll
        for (int l = 0; l < NLQ; l++) {
            if (4 * NLQ == G::W4 || fq + l < G::W4) {
                // slots that hang over the left / right image border fetch a neighbouring quad and are rebuilt in LDS (patch_borders)
                const int xo = min(max(tx0 - HALO + 4 * (fq + l), 0), p.nx - 4);
                const unsigned off = (unsigned)(rowbase + xo) * 4u;
                const auto a = __builtin_amdgcn_raw_buffer_load_b128(bx.r, (int)off, 0, 0);
                const auto b = __builtin_amdgcn_raw_buffer_load_b128(by.r, (int)off, 0, 0);
                pre[0][l] = __builtin_bit_cast(ft_v4f, a);
                pre[1][l] = __builtin_bit_cast(ft_v4f, b);
            }
        }
    };
    auto commit = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int pl = 0; pl < 2; pl++)
#pragma unroll
            for (int l = 0; l < NLQ; l++)
                if (4 * NLQ == G::W4 || fq + l < G::W4) reinterpret_cast<ft_v4f *>(raw4)[(pl * CH + frow) * P4 + fq + l] = pre[pl][l];
    };
    // border strips only (wave-uniform): rebuild the reflected halo columns from the columns of the same LDS row.
    // left: x = -k -> k;  right: x = nx-1+k -> nx-k  (gaussian.cpp:345-349)
    auto patch_borders = [&](const WavePos &t) __attribute__((always_inline)) {
        float *rawf = reinterpret_cast<float *>(raw4);
        const int tx0 = FTW_X0(t);
        for (int i = lane; i < 2 * CH * 2 * HALO; i += 64) {
            const int h = i % (2 * HALO), rr_ = i / (2 * HALO);  // rr_ = plane * CH + row
            int c, x;
            if (h < HALO) { c = h; x = tx0 - HALO + c; if (x >= 0) continue; }
            else { x = p.nx + (h - HALO); c = x - tx0 + HALO; if (c >= G::W) continue; }
            const int sc = fir_reflect(x, p.nx) - tx0 + HALO;
            if (sc < 0 || sc >= G::W) continue;
            rawf[rr_ * P4 * 4 + c] = rawf[rr_ * P4 * 4 + sc];
        }
    };

    // ---- per-lane roles.  Row pass: (row of the chunk, 16-pixel piece) = (lane / 4, lane % 4); column pass: column = lane
    const int rr = lane >> 2, rs = lane & 3;
    using hist_t = std::conditional_t<FTW_DHIST != 0, double, float>;
    hist_t hist[3][2 * R];  // per plane: the column's last 2R row-filtered values of the previous chunk
#pragma unroll
    for (int pl = 0; pl < 3; pl++)
#pragma unroll
        for (int i = 0; i < 2 * R; i++) hist[pl][i] = 0;

    // one plane of one chunk: row pass -> ring -> column pass; o[] = the 16 output rows of this lane's column
    auto plane_pass = [&](auto pl_tag, float (&o)[CH]) __attribute__((always_inline)) {
        constexpr int PL = decltype(pl_tag)::value;
        {
            float pr[G::NW];
            double dpr[G::NW];
            ftw_row_products<R, PL>(raw4, rr, rs, pr);
            float4 *rdst = reinterpret_cast<float4 *>(ring + rr * RP + 16 * rs);
#pragma unroll
            for (int g = 0; g < NG; g++) {
                float og[ILP];
                ft_group<R, FMA, ILP, G::NW>(pr, dpr, ILP * g, p.B, og);
#pragma unroll
                for (int h = 0; h < ILP / 4; h++) rdst[g * (ILP / 4) + h] = make_float4(og[4 * h], og[4 * h + 1], og[4 * h + 2], og[4 * h + 3]);
            }
        }
        ftw_wave_sync();
        float cw[CH + 2 * R];  // column window: the previous chunk's last 2R rows, then this chunk's 16
        double dcw[CH + 2 * R];
#pragma unroll
        for (int i = 0; i < 2 * R; i++) {
            if (FTW_DHIST) { dcw[i] = hist[PL][i]; cw[i] = 0.f; }
            else cw[i] = (float)hist[PL][i];
        }
#pragma unroll
        for (int k = 0; k < CH; k++) cw[2 * R + k] = ring[k * RP + lane];
        ftw_wave_sync();
#pragma unroll
        for (int g = 0; g < NG; g++) {
            float og[ILP];
            ft_group<R, FMA, ILP, CH + 2 * R, FTW_DHIST ? 2 * R : 0>(cw, dcw, ILP * g, p.B, og);
#pragma unroll
            for (int e = 0; e < ILP; e++) o[ILP * g + e] = og[e];
        }
#pragma unroll
        for (int i = 0; i < 2 * R; i++) hist[PL][i] = FTW_DHIST ? (hist_t)dcw[CH + i] : (hist_t)cw[CH + i];
    };

    WavePos cur = at_tile((long)wg * G::WAVES + wave);
    if (cur.tile >= tiles) return;  // wave-uniform; nobody waits for this wave
    prefetch(cur);
    for (;;) {
        commit();
        ftw_wave_sync();
        if (!(FTW_X0(cur) - HALO >= 0 && FTW_X0(cur) - HALO + G::W <= p.nx)) {
            patch_borders(cur);
            ftw_wave_sync();
        }
        WavePos nxt = cur;
        if (cur.chunk + 1 < FTW_NCHUNKS(cur)) nxt.chunk++;
        else nxt = at_tile(cur.tile + workers);
        const bool have_nxt = nxt.tile < tiles;
        if (have_nxt) prefetch(nxt);  // lands while this chunk computes

        const int oi_base = cur.chunk * CH - 2 * R, tnrows = FTW_NROWS(cur);
        const int gx = FTW_X0(cur) + lane;
        const unsigned plane_bytes = (unsigned)p.nx * (unsigned)p.ny * 4u;
        const unsigned crow0 = (unsigned)(FTW_Y0(cur) + oi_base) * (unsigned)p.nx * 4u;  // wraps for rows above the segment: those rows are skipped
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>;
        if (OUT == 0) {
            // A, B, C straight from the column pass: one dword per lane and row (256 B per wave), the plane as a hardware
            // buffer, the row as a scalar byte offset
            auto store_plane = [&](float *plane, const float (&o)[CH]) __attribute__((always_inline)) {
                const FtBuffer cbuf = ft_make_buffer(plane + (size_t)cur.frame * p.frame_stride, plane_bytes);
                if (gx < p.nx) {
#pragma unroll
                    for (int r = 0; r < CH; r++)
                        if (oi_base + r >= 0 && oi_base + r < tnrows)  // wave-uniform
                            ft_buffer_store(cbuf, (unsigned)gx * 4u, crow0 + (unsigned)r * (unsigned)p.nx * 4u, o[r]);
                }
            };
            float o[CH];
            plane_pass(I0(), o);
            store_plane(p.out0, o);
            plane_pass(I1(), o);
            store_plane(p.out1, o);
            plane_pass(I2(), o);
            store_plane(p.out2, o);
        } else {
            float oa[CH], ob[CH], oc[CH];
            plane_pass(I0(), oa);
            plane_pass(I1(), ob);
            plane_pass(I2(), oc);
            const FtBuffer cbuf = ft_make_buffer(p.out0 + (size_t)cur.frame * p.frame_stride, plane_bytes);
            unsigned char *tqf = p.tq ? p.tq + (((size_t)cur.frame * p.frame_stride) >> 2) : nullptr;
#pragma unroll
            for (int r = 0; r < CH; r++) {
                if (oi_base + r >= 0 && oi_base + r < tnrows) {  // wave-uniform
                    const float v = harris_response_value<0>(oa[r], ob[r], oc[r], p.k);
                    if (gx < p.nx) ft_buffer_store(cbuf, (unsigned)gx * 4u, crow0 + (unsigned)r * (unsigned)p.nx * 4u, v);
                    if (tqf) {  // kernel-uniform
                        // harris_quad_bits() with the quad's four responses in four lanes: the neighbours inside the quad by
                        // quad-permute DPP ([1,2,3,3] / [0,0,1,2]), the three tests as lane masks; the lanes at the quad's
                        // ends have no test on that side
                        const unsigned vb = __builtin_bit_cast(unsigned, v);
                        const float right = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(vb, 0xF9, 0xf, 0xf, true));
                        const float left = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(vb, 0x90, 0xf, 0xf, true));
                        const unsigned long long keep = __ballot(!(v < p.Th)) & (__ballot(!(right >= v)) | 0x8888888888888888ull) &
                                                        (__ballot(!(left > v)) | 0x1111111111111111ull);
                        if ((lane & 3) == 0 && gx < p.nx)
                            tqf[((unsigned)(FTW_Y0(cur) + oi_base + r) * (unsigned)p.nx + (unsigned)gx) >> 2] = (unsigned char)((keep >> lane) & 0xfull);
                    }
                }
            }
        }
        if (!have_nxt) break;
        cur = nxt;
    }
}

// ------------------------------------------------------------------ host side
template <int R, int OUT>
static imgfd_status launch_tensor_wave_r(imgfd_ctx *ctx, TensorParams &p, int n_frames)
{
    using G = WaveGeom<R>;
    const int strips = ceil_div(p.nx, G::TW);
    // Workers = the waves the chip holds at once (persistent: each walks its share of the tiles).  A tile is (rows + 2R)
    // rows in chunks of CH; the pass takes ceil(tiles / workers) tiles of (chunks per tile) steps per worker: pick the
    // segment count that minimises that product (ties: fewer, longer segments = less halo work).
    const long slots = (long)ctx->num_cu * G::WAVES;
    long best_cost = -1;
    int seg = p.ny;
    for (int nseg = 1; nseg <= ceil_div(p.ny, G::CH); nseg++) {
        int m = ceil_div(ceil_div(p.ny, nseg) + 2 * R, G::CH);
        if (m < 2) m = 2;
        const int sr = m * G::CH - 2 * R;  // (rows + 2R) fills whole chunks
        const long tiles = (long)strips * ceil_div(p.ny, sr) * n_frames;
        const long cost = ((tiles + slots - 1) / slots) * m;
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; seg = sr; }
    }
    p.seg_rows = seg;
    if (ctx->tune.tensor_seg > 0) p.seg_rows = ctx->tune.tensor_seg;
    p.nstrips = strips;
    p.nseg = ceil_div(p.ny, p.seg_rows);
    p.n_frames = n_frames;
    const long tiles = (long)p.nstrips * p.nseg * n_frames;
    long workers = std::min<long>(tiles, slots);
    if (ctx->tune.tensor_workers > 0) workers = std::min<long>(tiles, ctx->tune.tensor_workers);  // tests: several tiles per worker on small images
    dim3 grid((unsigned)((workers + G::WAVES - 1) / G::WAVES));
    p.xcd_remap = ctx->tune.xcd_remap;
    auto go = [&](auto kern) -> imgfd_status {
        IMGFD_HIP(ctx, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::lds_bytes));
        hipLaunchKernelGGL(kern, grid, dim3(G::NT), G::lds_bytes, ctx->stream, p);
        IMGFD_HIP(ctx, hipGetLastError());
        ctx->tensor_wave_launches++;
        return IMGFD_OK;
    };
    if (ctx->fir_mode) return go(fir_tensor_wave<R, true, OUT>);
    return go(fir_tensor_wave<R, false, OUT>);
}

// The wave-autonomous kernel serves 16-byte aligned planes whose rows are whole quads and at least one strip wide, of
// less than 2^32 bytes per plane (32-bit buffer offsets).  p arrives filled by launch_tensor_march (fir_tensor.hip).
imgfd_status launch_tensor_wave(imgfd_ctx *ctx, TensorParams &p, int n_frames, int R, int out_mode)
{
    if (p.nx < WaveGeom<7>::TW || (size_t)p.nx * p.ny * 4 >= ((size_t)1 << 32)) return IMGFD_ERR_UNSUPPORTED;
#define FTW_GO(RR)                                                               \
    case RR:                                                                     \
        if (out_mode == 2) return launch_tensor_wave_r<RR, 2>(ctx, p, n_frames); \
        return launch_tensor_wave_r<RR, 0>(ctx, p, n_frames);
    switch (R) {
        FTW_GO(7)
        FTW_GO(3)
        FTW_GO(1)
    }
#undef FTW_GO
    return IMGFD_ERR_UNSUPPORTED;
}
