// image_amd/csrc/gauss_grad_march.hip -- K1 + K2 for u8 frames as a marching kernel: the discrete Gaussian of sigma_d and
// the gradient of the smoothed image, every row filtered once.
//
// Same arithmetic as gauss_grad_tile (gauss_grad.hip; harris.cpp:511-514 = gaussian.cpp:289-395 + gradient.cpp:17-106),
// different shape.  The tile kernel owns 64 x 24 outputs and recomputes the row pass on 33 rows and the column pass on 27
// for them (1.44 x and 1.18 x the arithmetic, 3.5 conversions per output and pass); here a 256-thread workgroup owns a strip
// of 240 output columns and walks down a segment of the frame in chunks of 16 rows, like the structure-tensor kernel:
//   * row pass: thread (row, 16-column group) slides a register window of 16 + 2R bytes over its 16 outputs (1.4 conversions
//     per output), results as floats into an LDS ring of 16 rows;
//   * column pass, register-marching: thread c owns computed column c (256 of them: x0-8 .. x0+247) for the whole segment,
//     keeps the last 2R row-filtered values, reads 16 new ones per chunk and emits 16 smoothed rows into an 18-row LDS ring;
//   * gradient: a wave per row, a lane per quad of pixels, three 16-byte LDS reads per row of the stencil, float4 streaming
//     stores; border pixels take the gradient of the nearest interior pixel (gradient.cpp:40-55): the quad / row beside them
//     writes them too; the image borders of the Gaussian are the reference's reflections (gaussian.cpp:345-349, 376-380): rows through
//     the load address, columns rebuilt inside the LDS tile of the first / last strip.
// Step k:  [ fetch chunk k+1 -> registers | row pass chunk k | gradient rows of chunk k-1 ]  barrier
//          [ column pass chunk k | registers -> raw tile of chunk k+1 ]                      barrier
// LDS 39.7 KB: four workgroups per CU (71 registers).  HBM: 1 B read (+ 13 % strip halo, + segment halo rows) and 8 B written per pixel.
#include "common.h"
#include "fir_device.h"

#include <algorithm>

#define GM_TW 240   // output columns of a strip
#define GM_CW 256   // computed columns: x0-8 .. x0+247 (the gradient needs x0-1 .. x0+240)
#define GM_NT 256
#define GM_CH 16
#define GM_RAWQ 17  // 16-byte slots of a raw row: x0-16 .. x0+255
#define GM_RAWP 69  // raw row pitch in dwords (odd multiple pattern: the row pass's dword reads are conflict-free)
#define GM_RINGP 260  // row-filtered ring, row pitch in floats
#define GM_OBR 18   // smoothed rows kept: the 16 of a chunk and the last two of the one before
#define GM_OBP 260
#ifndef GM_ILP
#define GM_ILP 4
#endif

struct GaussMarchParams {
    const unsigned char *in;
    float *Ix, *Iy;
    int nx, ny, in_pitch;
    long in_frame_stride;
    int seg_rows, nstrips, nseg;
    int xcd_order;       // 1: XCD-aware order of the tiles (imgfd_xcd_tile): neighbouring strips march on one XCD
    unsigned *rowcount;  // optional: ny counters per frame, cleared by the workgroups of strip 0 (for the NMS kernel two launches on:
                         // a fill of its own, queued behind the structure-tensor kernel, sat 20 us on the chain's critical path)
    double B[8];
};

// ILP outputs o0 .. of one 1-D pass from the float window w (converted to double on first use), the reference's order per
// output (fir_window8), chains advanced together
template <int R, bool FMA, int ILP, int NWIN>
__device__ __forceinline__ void gm_group(const float (&w)[NWIN], double (&dw)[NWIN], int o0, const double *B, float (&out)[ILP])
{
#pragma unroll
    for (int k = (o0 == 0 ? 0 : o0 + 2 * R); k < o0 + ILP + 2 * R; k++) dw[k] = (double)w[k];
    double sum[ILP];
#pragma unroll
    for (int g = 0; g < ILP; g++) sum[g] = B[0] * dw[o0 + g + R];
#pragma unroll
    for (int j = 1; j <= R; j++) {
        double pair[ILP];
#pragma unroll
        for (int g = 0; g < ILP; g++) pair[g] = dw[o0 + g + R - j] + dw[o0 + g + R + j];
#pragma unroll
        for (int g = 0; g < ILP; g++) {
            if (FMA) sum[g] = __builtin_fma(B[j], pair[g], sum[g]);
            else sum[g] += B[j] * pair[g];
        }
    }
#pragma unroll
    for (int g = 0; g < ILP; g++) out[g] = (float)sum[g];
}

template <int R, int GRAD, bool FMA>
__global__ void __launch_bounds__(GM_NT) IMGFD_WAVES_PER_EU(4, 4) gauss_grad_march(GaussMarchParams p)
{
    static_assert(R >= 1 && R <= 7, "raw tile holds 8 columns either side of the computed ones; 2R rows of history fit a chunk");
    typedef float v4f __attribute__((vector_size(16)));
    __shared__ unsigned raw[GM_CH * GM_RAWP];
    __shared__ __attribute__((aligned(16))) float ring[GM_CH * GM_RINGP];
    __shared__ __attribute__((aligned(16))) float obuf[GM_OBR * GM_OBP];
    const int tid = threadIdx.x;
    // tile = (strip, segment, frame), strip fastest
    int t = (int)(p.xcd_order ? imgfd_xcd_tile(blockIdx.x, gridDim.x) : blockIdx.x);
    const int strip = t % p.nstrips; t /= p.nstrips;
    const int seg = t % p.nseg;
    const int frame = t / p.nseg;
    const int x0 = strip * GM_TW, y0 = seg * p.seg_rows;
    const int nrows = min(p.ny, y0 + p.seg_rows) - y0;
    const int yb = y0 - 1 - R;  // image row of row-filtered row 0 of chunk 0
    // output rows of a segment, v = y - y0, after chunk k: v <= 16 k + 13 - 2R (smoothed rows v .. v+2 exist)
    const int k_last = max(0, (nrows - 1 - (13 - 2 * R) + GM_CH - 1) / GM_CH);
    const unsigned char *inf = p.in + (size_t)frame * p.in_frame_stride;
    const bool border_strip = x0 == 0 || x0 - 16 + 16 * GM_RAWQ > p.nx;
    if (p.rowcount && strip == 0)
        for (int i = tid; i < nrows; i += GM_NT) p.rowcount[(size_t)frame * p.ny + y0 + i] = 0;

    // ---- staging: slot i = (row, 16-byte slot q) of a chunk's 16 x 17 slots; thread tid owns slot tid, the first 16 also 256 + tid
    const int s0_row = tid / GM_RAWQ, s0_q = tid - s0_row * GM_RAWQ;
    const int s1_row = (GM_NT + tid) / GM_RAWQ, s1_q = (GM_NT + tid) - s1_row * GM_RAWQ;
    const bool has_s1 = tid < GM_CH * GM_RAWQ - GM_NT;
    uint4 pre0 = make_uint4(0, 0, 0, 0), pre1 = make_uint4(0, 0, 0, 0);
    auto prefetch = [&](int k) __attribute__((always_inline)) {
        // every slot is ONE aligned 16-byte load from an in-range address; slots hanging over the left / right image border
        // fetch a neighbouring slot and are rebuilt in LDS (patch_borders)
        const int yc = yb + k * GM_CH;
        {
            const int gy = fir_reflect(yc + s0_row, p.ny);
            const int xq = min(max(x0 - 16 + 16 * s0_q, 0), p.nx - 16);
            pre0 = *reinterpret_cast<const uint4 *>(inf + (size_t)gy * p.in_pitch + xq);
        }
        if (has_s1) {
            const int gy = fir_reflect(yc + s1_row, p.ny);
            const int xq = min(max(x0 - 16 + 16 * s1_q, 0), p.nx - 16);
            pre1 = *reinterpret_cast<const uint4 *>(inf + (size_t)gy * p.in_pitch + xq);
        }
    };
    auto commit = [&]() __attribute__((always_inline)) {
        unsigned *d0 = raw + s0_row * GM_RAWP + 4 * s0_q;
        d0[0] = pre0.x; d0[1] = pre0.y; d0[2] = pre0.z; d0[3] = pre0.w;
        if (has_s1) {
            unsigned *d1 = raw + s1_row * GM_RAWP + 4 * s1_q;
            d1[0] = pre1.x; d1[1] = pre1.y; d1[2] = pre1.z; d1[3] = pre1.w;
        }
    };
    // first / last strip: columns left of x = 0 (-k -> k) and right of x = nx-1 (nx-1+k -> nx-k) from the columns of the same
    // LDS row (gaussian.cpp:345-349).  Columns whose source lies outside the tile feed only outputs beyond the image.
    auto patch_borders = [&]() __attribute__((always_inline)) {
        unsigned char *rb = reinterpret_cast<unsigned char *>(raw);
        constexpr int W = 16 * GM_RAWQ;
        for (int i = tid; i < GM_CH * 48; i += GM_NT) {
            const int row = i / 48, h = i - row * 48;
            int c;
            if (h < 16) { c = h; if (x0 - 16 + c >= 0) continue; }
            else { c = p.nx - (x0 - 16) + (h - 16); if (c >= W || c < 0) continue; }
            const int sc = fir_reflect(x0 - 16 + c, p.nx) - (x0 - 16);
            if (sc < 0 || sc >= W) continue;
            rb[row * GM_RAWP * 4 + c] = rb[row * GM_RAWP * 4 + sc];
        }
    };

    // ---- roles
    const int rr = tid >> 4, rs = tid & 15;  // row pass: row of the chunk, 16-column group
    const int col = tid;                     // column pass: computed column
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    float hist[2 * R];
#pragma unroll
    for (int i = 0; i < 2 * R; i++) hist[i] = 0.f;

    auto row_pass = [&]() __attribute__((always_inline)) {
        // window: raw bytes 8 - R + 16 rs .. of the row = computed columns 16 rs - R .. 16 rs + 15 + R
        constexpr int SH = (8 - R) & 3, D0 = (8 - R) >> 2, NWD = (SH + GM_CH + 2 * R + 3) / 4;
        const unsigned *row32 = raw + rr * GM_RAWP + D0 + 4 * rs;
        unsigned wvd[NWD];
#pragma unroll
        for (int k = 0; k < NWD; k++) wvd[k] = row32[k];
        float w[16 + 2 * R];
#pragma unroll
        for (int k = 0; k < 16 + 2 * R; k++) w[k] = (float)((wvd[(SH + k) >> 2] >> (8 * ((SH + k) & 3))) & 0xffu);  // v_cvt_f32_ubyteN: exact
        double dw[16 + 2 * R];
        // (four 16-byte writes 64 bytes apart per lane: 4-way bank conflicts, 48 extra LDS cycles per wave and step -- against
        // 3.8 KB of padding for a conflict-free layout, which costs the fourth resident workgroup)
        v4f *dst = reinterpret_cast<v4f *>(ring + rr * GM_RINGP) + 4 * rs;
#pragma unroll
        for (int g = 0; g < 16 / GM_ILP; g++) {
            float o[GM_ILP];
            gm_group<R, FMA, GM_ILP, 16 + 2 * R>(w, dw, GM_ILP * g, p.B, o);
#pragma unroll
            for (int h = 0; h < GM_ILP / 4; h++) dst[g * (GM_ILP / 4) + h] = v4f{o[4 * h], o[4 * h + 1], o[4 * h + 2], o[4 * h + 3]};
        }
    };
    // smoothed rows of chunk k: u = 16 k - 2R + i (u = 0 is image row y0 - 1), kept at ring row u mod 24
    auto col_pass = [&](int k, int ubase) __attribute__((always_inline)) {
        float cw[16 + 2 * R];
#pragma unroll
        for (int i = 0; i < 2 * R; i++) cw[i] = hist[i];
#pragma unroll
        for (int r = 0; r < GM_CH; r++) cw[2 * R + r] = ring[r * GM_RINGP + col];
#pragma unroll
        for (int i = 0; i < 2 * R; i++) hist[i] = cw[GM_CH + i];
        double dcw[16 + 2 * R];
#pragma unroll
        for (int g = 0; g < 16 / GM_ILP; g++) {
            float o[GM_ILP];
            gm_group<R, FMA, GM_ILP, 16 + 2 * R>(cw, dcw, GM_ILP * g, p.B, o);
#pragma unroll
            for (int e = 0; e < GM_ILP; e++) {
                const int i = GM_ILP * g + e;
                int pos = ubase + i;
                pos = pos >= GM_OBR ? pos - GM_OBR : pos;
                if (k > 0 || i >= 2 * R) obuf[pos * GM_OBP + col] = o[e];
            }
        }
    };
    // gradient rows v_lo .. v_hi of the segment: wave wv takes rows v_lo + wv + 4 i, lane j the quad x0 + 4 j .. + 3
    float *Ixf = p.Ix + (size_t)frame * p.nx * p.ny, *Iyf = p.Iy + (size_t)frame * p.nx * p.ny;
    auto grad_rows = [&](int v_lo, int v_hi) __attribute__((always_inline)) {
        const int x = x0 + 4 * lane;
        const bool live = lane < GM_TW / 4 && x < p.nx;
        for (int v = v_lo + wv; v <= v_hi; v += 4) {
            const int y = y0 + v;
            // the first and the last image row take the gradient of the row inside (gradient.cpp:40-55): written together with it
            if (y == 0 || y == p.ny - 1) continue;
            int pu = v % GM_OBR;                           // smoothed row y - 1 (u = v), then y, y + 1
            int pm = pu + 1; pm = pm >= GM_OBR ? pm - GM_OBR : pm;
            int pd = pm + 1; pd = pd >= GM_OBR ? pd - GM_OBR : pd;
            if (!live) continue;
            const v4f *ru = reinterpret_cast<const v4f *>(obuf + pu * GM_OBP) + 2 + lane;  // quad of computed columns 8 + 4 lane ..
            const v4f *rm = reinterpret_cast<const v4f *>(obuf + pm * GM_OBP) + 2 + lane;
            const v4f *rd = reinterpret_cast<const v4f *>(obuf + pd * GM_OBP) + 2 + lane;
            float m[6], up[6], dn[6];
            {
                const v4f a = rm[-1], b = rm[0], c = rm[1];
                m[0] = a[3]; m[1] = b[0]; m[2] = b[1]; m[3] = b[2]; m[4] = b[3]; m[5] = c[0];
            }
            if (GRAD == IMGFD_SOBEL_OPERATOR) {
                const v4f a = ru[-1], b = ru[0], c = ru[1];
                up[0] = a[3]; up[1] = b[0]; up[2] = b[1]; up[3] = b[2]; up[4] = b[3]; up[5] = c[0];
                const v4f d = rd[-1], e = rd[0], f = rd[1];
                dn[0] = d[3]; dn[1] = e[0]; dn[2] = e[1]; dn[3] = e[2]; dn[4] = e[3]; dn[5] = f[0];
            } else {
                const v4f b = ru[0], e = rd[0];
                up[0] = 0.f; up[1] = b[0]; up[2] = b[1]; up[3] = b[2]; up[4] = b[3]; up[5] = 0.f;
                dn[0] = 0.f; dn[1] = e[0]; dn[2] = e[1]; dn[3] = e[2]; dn[4] = e[3]; dn[5] = 0.f;
            }
            float gx[4], gy[4];
#pragma unroll
            for (int e = 0; e < 4; e++) {  // pixel x + e sits at index e + 1 of the six
                if (GRAD == IMGFD_SOBEL_OPERATOR) {  // gradient.cpp:80-87: float sums, double constants, double adds, float store
                    gx[e] = (float)(1. / 4. * (m[e + 2] - m[e]) + 1. / 8. * (up[e + 2] + dn[e + 2] - up[e] - dn[e]));
                    gy[e] = (float)(1. / 4. * (dn[e + 1] - up[e + 1]) + 1. / 8. * (dn[e + 2] + dn[e] - up[e + 2] - up[e]));
                } else {                             // gradient.cpp:34-35 (see gauss_grad.hip)
                    gx[e] = 0.5f * (m[e + 2] - m[e]);
                    gy[e] = 0.5f * (dn[e + 1] - up[e + 1]);
                }
            }
            // first / last image column: the gradient of the neighbour inside (whole quads: nx is a multiple of 16)
            if (x == 0) { gx[0] = gx[1]; gy[0] = gy[1]; }
            if (x + 4 == p.nx) { gx[3] = gx[2]; gy[3] = gy[2]; }
            const size_t o = (size_t)y * p.nx + x;
            const v4f gxv = {gx[0], gx[1], gx[2], gx[3]}, gyv = {gy[0], gy[1], gy[2], gy[3]};
            IMGFD_STREAM_STORE(gxv, reinterpret_cast<v4f *>(Ixf + o));
            IMGFD_STREAM_STORE(gyv, reinterpret_cast<v4f *>(Iyf + o));
            if (y == 1 || y == p.ny - 2) {  // wave-uniform: the border row beside it gets the same values
                const size_t ob = (y == 1 ? (size_t)0 : (size_t)(p.ny - 1) * p.nx) + x;
                IMGFD_STREAM_STORE(gxv, reinterpret_cast<v4f *>(Ixf + ob));
                IMGFD_STREAM_STORE(gyv, reinterpret_cast<v4f *>(Iyf + ob));
            }
        }
    };

    prefetch(0);
    commit();
    if (border_strip) { __syncthreads(); patch_borders(); }
    __syncthreads();
    int ubase = GM_OBR - 2 * R;  // (16 k - 2R) mod GM_OBR at k = 0
    for (int k = 0; k <= k_last; k++) {
        if (k < k_last) prefetch(k + 1);
        row_pass();
        if (k > 0) grad_rows(max(0, GM_CH * (k - 2) + 14 - 2 * R), min(nrows - 1, GM_CH * (k - 1) + 13 - 2 * R));  // rows completed by chunk k-1
        __syncthreads();  // ring = row-filtered chunk k; the smoothed rows of chunk k-1 have been read, the raw tile consumed
        col_pass(k, ubase);
        ubase += GM_CH; ubase = ubase >= GM_OBR ? ubase - GM_OBR : ubase;
        if (k < k_last) {
            commit();
            if (border_strip) { __syncthreads(); patch_borders(); }
        }
        __syncthreads();  // smoothed rows of chunk k complete, raw tile of chunk k+1 complete, ring free
    }
    grad_rows(max(0, GM_CH * k_last + 14 - 2 * R - GM_CH), nrows - 1);
}

bool gauss_grad_march_supported(const void *d_in, int in_is_u8, int in_pitch, size_t in_frame_stride, const float *d_Ix,
                                const float *d_Iy, int nx, int ny)
{
    return in_is_u8 && nx % 16 == 0 && nx >= 256 && ny >= 16 && (size_t)d_in % 16 == 0 && in_pitch % 16 == 0 && in_frame_stride % 16 == 0 &&
           (size_t)d_Ix % 16 == 0 && (size_t)d_Iy % 16 == 0;
}

imgfd_status launch_gauss_grad_march(imgfd_ctx *ctx, const void *d_in, int in_pitch, size_t in_frame_stride, float *d_Ix,
                                     float *d_Iy, int nx, int ny, int n_frames, const double *B, int grad_type, unsigned *d_rowcount)
{
    constexpr int R = 3;
    GaussMarchParams p;
    memset(&p, 0, sizeof p);
    p.in = (const unsigned char *)d_in; p.Ix = d_Ix; p.Iy = d_Iy; p.nx = nx; p.ny = ny; p.in_pitch = in_pitch;
    p.in_frame_stride = (long)in_frame_stride;
    p.rowcount = d_rowcount;
    memcpy(p.B, B, sizeof(double) * (R + 1));
    p.nstrips = ceil_div(nx, GM_TW);
    // Segment length: a segment of sr rows takes m = ceil((sr + 2R - 13) / 16) + 1 steps, so sr = 16 m - 2 - 2R fills them; the
    // launch takes ceil(workgroups / slots) rounds of m steps with three workgroups resident per CU: the segment count that
    // minimises the product -- and of the counts within 3 % of that minimum the one with the SHORTEST segments: the kernel runs beside
    // the other detectors' kernels, and many short workgroups interleave with them better than few long ones (32 4K frames, sustained:
    // segments of 1080 rows, the model's minimum, 71.2 Gpixel/s; 135-540 rows 72.5-72.8; profiles/r06/gauss_march_segments.txt)
    const long slots = 4L * ctx->num_cu;
    long best = -1;
    int seg = ny;
    for (int pass = 0; pass < 2; pass++)
        for (int m = 2; m <= ceil_div(ny + 2 * R + 2, GM_CH) + 1; m++) {
            const int sr = std::min(ny, GM_CH * m - 2 - 2 * R);
            if (sr < 1) continue;
            const long wgs = (long)p.nstrips * ceil_div(ny, sr) * n_frames;
            const long cost = ((wgs + slots - 1) / slots) * m;
            if (pass == 0) { if (best < 0 || cost < best) best = cost; }
            else if (100 * cost <= 103 * best) { seg = sr; break; }  // m ascending: the first hit is the shortest
            if (sr >= ny) break;
        }
    if (ctx->tune.gauss_march_seg > 0) seg = std::min(ny, std::max(2, ctx->tune.gauss_march_seg));
    // the last segment holds two rows at least: its last row is evaluated one row up (gradient.cpp:40-55) and the smoothed row
    // above that must be one the segment computes
    while (seg < ny && ny - (ceil_div(ny, seg) - 1) * seg < 2) seg++;
    p.seg_rows = seg;
    p.nseg = ceil_div(ny, seg);
    p.xcd_order = 1;
    const dim3 grid((unsigned)((long)p.nstrips * p.nseg * n_frames));
    const bool sobel = grad_type == IMGFD_SOBEL_OPERATOR;
#define GM_LAUNCH(G, F) hipLaunchKernelGGL((gauss_grad_march<R, G, F>), grid, dim3(GM_NT), 0, ctx->stream, p)
    if (ctx->fir_mode) { if (sobel) GM_LAUNCH(1, true); else GM_LAUNCH(0, true); }
    else { if (sobel) GM_LAUNCH(1, false); else GM_LAUNCH(0, false); }
#undef GM_LAUNCH
    IMGFD_HIP(ctx, hipGetLastError());
    ctx->gauss_march_launches++;
    return IMGFD_OK;
}
