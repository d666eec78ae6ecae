// image_amd/csrc/fhog_fused.hip -- fHOG histogram pass for cell_size 8 as ONE kernel (K13 + K14 of fhog.hip fused).
//
// Replaces the gradient / histogram loop of dlib's impl_extract_fhog_features,
// image.dlib/inst/dlib-19.20/dlib/image_transforms/fhog.h:821-968, for the R default cell_size = 8.
//
// The reference walks the image in raster order and adds every pixel's gradient magnitude into 4 histogram cells with
// `+=`: each float bin is a sum in a fixed order.  This kernel keeps that order and never writes a per-pixel plane:
//
//   phase 1  a workgroup stages the (v, o) of a window of 72 x 136 pixels in LDS: v = sqrt of the squared gradient length
//            of the strongest colour channel, o = its orientation bin.  A thread owns 4 consecutive pixels (12 bytes =
//            3 dwords) of a run of rows and slides a 3-row register window down the run: 5 dword loads per 4 pixels.
//            The orientation is a table look-up -- the bin is a pure function of the integer gradient (tx, ty), 511 x 511
//            possibilities, tabulated once per context by the reference's own float chain (fhog_best_orientation) --
//            instead of nine dot products per pixel.
//   phase 2  one thread per histogram cell (8 x 16 cells) walks the 16 x 16 pixels that vote into it in raster order
//            and adds wy * (wx * v) into bin o of its private LDS histogram with ds_add_f32.  LDS operations of one wave
//            execute in order and no other thread touches the cell, so the bin receives the reference's summands in the
//            reference's order; there is no read-modify-write round trip to wait for.  With cell_size 8 the bilinear
//            weights are the dyadic constants (k + 0.5) / 8, exact in float, so they are compile-time literals.
//   A workgroup marches down `bands` bands of 8 cell rows: the last 8 pixel rows of a band's window are the first 8 of
//   the next one and stay in the LDS ring.
//
// The 8-wide body / scalar tail split of the reference (different colour tie-break, different association of the
// weights; fhog.hip) is kept: workgroups whose window touches the image border or the tail columns run the general form
// of both phases (wave-uniform branch).
#include "fhog_device.h"

#include <algorithm>

namespace {

constexpr int FH_CS = 8;
constexpr int FH_CR = 8;                      // cell rows per band
constexpr int FH_CC = 16;                     // cell columns per workgroup
constexpr int FH_WX = FH_CS * FH_CC + FH_CS;  // 136 window columns
constexpr int FH_NEW = FH_CS * FH_CR;         // 64 new pixel rows per band
constexpr int FH_RING = FH_NEW + FH_CS;       // 72 rows in the LDS ring
constexpr int FH_PV = FH_WX;                  // dwords per row of V
// rows are stored in groups of 8 with 4 extra dwords between the groups: consecutive cell rows then sit an odd number
// of 16-byte slots apart and the ds_read_b128 lane groups of phase 2 see 16 distinct slots (lane <-> cell mapping below)
constexpr int FH_VGROUP = 8 * FH_PV + 4;
constexpr int FH_VDW = (FH_RING / 8) * FH_VGROUP;
constexpr int FH_PO = 144;                    // u16 per row of O (288 bytes: 8 rows = a whole number of 256-byte bank rows)
constexpr int FH_NCELL = FH_CR * FH_CC;       // 128 cells = 128 phase-2 threads
constexpr int FH_THREADS = 256;
constexpr size_t FH_LDS = (size_t)FH_VDW * 4 + (size_t)FH_RING * FH_PO * 2 + (size_t)18 * FH_NCELL * 4;

// bilinear weight of window row / column k (0..15) of a cell: fhog.h:823-826, :838-841 with cell_size 8
__host__ __device__ constexpr float fh_weight(int k) { return k < 8 ? (k + 0.5f) / 8 : (15.5f - k) / 8; }

// sqrt of an integer 0 .. 2 * 255^2, correctly rounded.  SQ 0: one v_rsq_f32 and a Newton step whose residual is an exact
// fma (verified against sqrtf for every possible argument: tests/test_fhog.py::test_fused_sqrt_exhaustive);
// SQ 1: the compiler's correctly rounded sqrtf.
template <int SQ>
__device__ __forceinline__ float fh_sqrt(int tl)
{
    const float f = (float)tl;
    if (SQ == 1) return sqrtf(f);
    const float r = __builtin_amdgcn_rsqf(fmaxf(f, 1.0f));
    const float y0 = f * r;
    const float h = 0.5f * r;
    const float e = __builtin_fmaf(-y0, y0, f);
    return __builtin_fmaf(e, h, y0);
}

struct FhRows {
    int x;      // image column of the group's first pixel (multiple of 4, may lie outside the image)
    int col;    // its window column
    int y;      // first image row
    int wr;     // its window row (ring slot = wr mod FH_RING)
    int nrows;
};

// phase 1 for one thread: `nrows` rows of one group of 4 pixels.  EDGE: loads clamped into the image, pixels outside
// [1, visible) vote 0, tail columns use the scalar colour rule.
template <bool EDGE, int SQ>
__device__ __forceinline__ void fh_phase1(const unsigned *__restrict__ img, const unsigned char *__restrict__ olut, float *V,
                                          unsigned short *O, const FhogGeom &g, int rd, const FhRows it)
{
    const int d0 = 3 * (it.x / 4);  // dword of the group's first byte in its row
    int di[5];
#pragma unroll
    for (int k = 0; k < 5; k++) di[k] = EDGE ? min(max(d0 - 1 + k, 0), rd - 1) : d0 - 1 + k;
    unsigned colvalid = 0xfu, body = 0xfu;
    if (EDGE) {
        colvalid = body = 0;
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const int xx = it.x + p;
            if (xx >= 1 && xx < g.visible_nc) colvalid |= 1u << p;
            if (xx < g.body_end) body |= 1u << p;
        }
    }
    auto load = [&](int y, unsigned(&w)[5]) {
        const int yy = EDGE ? min(max(y, 0), g.rows - 1) : y;
        const unsigned *row = img + (size_t)yy * rd;
#pragma unroll
        for (int k = 0; k < 5; k++) w[k] = row[di[k]];
    };
    auto byte_of = [](const unsigned(&w)[5], int b) -> int { return (int)((w[b >> 2] >> (8 * (b & 3))) & 0xffu); };

    unsigned up[5], cen[5], dn[5];
    load(it.y - 1, up);
    load(it.y, cen);
    load(it.y + 1, dn);
    int slot = it.wr % FH_RING;
    // the orientation bytes of a row are consumed one row later: the look-ups stay in flight behind the next row's
    // arithmetic
    float pv[4] = {0.f, 0.f, 0.f, 0.f};
    unsigned po[4] = {0u, 0u, 0u, 0u};
    int pslot = -1;
    auto flush = [&]() {
        float *vdst = V + (pslot >> 3) * FH_VGROUP + (pslot & 7) * FH_PV + it.col;
        *reinterpret_cast<float4 *>(vdst) = make_float4(pv[0], pv[1], pv[2], pv[3]);
        unsigned short *odst = O + pslot * FH_PO + it.col;
        *reinterpret_cast<uint2 *>(odst) = make_uint2((po[0] << 9) | (po[1] << 25), (po[2] << 9) | (po[3] << 25));
    };
#pragma unroll 1
    for (int r = 0; r < it.nrows; r++) {
        unsigned nxt[5] = {0u, 0u, 0u, 0u, 0u};
        if (r + 1 < it.nrows) load(it.y + r + 2, nxt);
        const bool rowvalid = !EDGE || (it.y + r >= 1 && it.y + r < g.visible_nr);
        int tl[4];
        unsigned no[4];
#pragma unroll
        for (int p = 0; p < 4; p++) {
            // simd8 get_gradient (:265-274) keeps the LATER channel on ties, the scalar one (:24-59) the EARLIER:
            // "later wins" is >= , i.e. > against (length - 1) on integers.  (Scalars, not arrays: a select between array
            // elements is turned into an indexed load from scratch.)
            const int b = EDGE ? (int)((body >> p) & 1u) : 1;
            int tx = 0, ty = 0, t = 0;
#pragma unroll
            for (int ch = 0; ch < 3; ch++) {
                const int gx = byte_of(cen, 4 + 3 * (p + 1) + ch) - byte_of(cen, 4 + 3 * (p - 1) + ch);
                const int gy = byte_of(dn, 4 + 3 * p + ch) - byte_of(up, 4 + 3 * p + ch);
                const int len = gx * gx + gy * gy;
                const bool take = ch == 0 || len > t - b;
                tx = take ? gx : tx; ty = take ? gy : ty; t = take ? len : t;
            }
            if (EDGE && !(rowvalid && ((colvalid >> p) & 1u))) t = 0;  // a zero vote leaves every sum unchanged
            tl[p] = t;
            no[p] = olut[(unsigned)((ty + 255) * 512 + (tx + 255))];
        }
        if (pslot >= 0) flush();
#pragma unroll
        for (int p = 0; p < 4; p++) { pv[p] = fh_sqrt<SQ>(tl[p]); po[p] = no[p]; }
        pslot = slot;
        slot = slot + 1 == FH_RING ? 0 : slot + 1;
#pragma unroll
        for (int k = 0; k < 5; k++) { up[k] = cen[k]; cen[k] = dn[k]; dn[k] = nxt[k]; }
    }
    flush();
}

// hist[(hr*HC + hc)*18 + o] for 1 <= hr <= cells_nr, 1 <= hc <= cells_nc (the layout of fhog_cell_hist) and the cell
// energies norm[(hr-1)*cells_nc + hc-1] (:959-968).  grid: (cell columns / 16, band groups, frames).
template <int SQ>
__global__ void __launch_bounds__(FH_THREADS) fhog_hist8(const unsigned char *__restrict__ rgb, size_t frame_stride,
                                                         const unsigned char *__restrict__ olut, float *__restrict__ hist,
                                                         float *__restrict__ norm, FhogGeom g, int bands_per_wg)
{
    HIP_DYNAMIC_SHARED(float, lds)
    float *V = lds;
    unsigned short *O = reinterpret_cast<unsigned short *>(lds + FH_VDW);
    float *H = reinterpret_cast<float *>(O + FH_RING * FH_PO);

    const int tid = threadIdx.x;
    const int f = blockIdx.z;
    const unsigned *img = reinterpret_cast<const unsigned *>(rgb + (size_t)f * frame_stride);
    const int rd = 3 * g.cols / 4;
    const int n_bands = (g.cells_nr + FH_CR - 1) / FH_CR;
    const int k0 = blockIdx.y * bands_per_wg, k1 = min(k0 + bands_per_wg, n_bands);
    const int hc0 = 1 + FH_CC * blockIdx.x;
    const int x0 = FH_CS * hc0 - 12;        // image column of window column 0 (128 bx - 4)
    const int Y0 = FH_CS * (1 + FH_CR * k0) - 12;  // image row of window row 0 (64 k0 - 4)
    // the two waves that run phase 2 alternate between workgroups (a workgroup's waves go to the four SIMDs in turn)
    const int role = (blockIdx.x + blockIdx.y) & 1;
    const bool cols_inside = x0 >= 4 && x0 + FH_WX + 1 <= min(g.visible_nc, g.body_end);
    const bool tail_window = x0 + FH_WX > g.body_end;

    for (int k = k0; k < k1; k++) {
        const int i = k - k0;
        // ---- phase 1: (v, o) of the window rows that are not in the ring yet
        const int wr_first = i ? FH_NEW * i + FH_CS : 0;
        const int seg_rows = i ? FH_CS : FH_CS + 1;  // 8 run segments of 8 rows (9 in the first band: 72 rows)
        const int y_first = Y0 + wr_first, y_end = Y0 + FH_NEW * i + FH_RING;
        const bool edge = !(cols_inside && y_first >= 1 && y_end <= g.visible_nr);
#pragma unroll 1
        for (int pass = 0; pass < 2; pass++) {
            FhRows it;
            if (pass == 0) {  // window columns 0..127: 32 groups x 8 run segments
                const int grp = tid & 31, seg = tid >> 5;
                it.x = x0 + 4 * grp; it.col = 4 * grp;
                it.wr = wr_first + seg * seg_rows; it.y = Y0 + it.wr; it.nrows = seg_rows;
            } else {          // window columns 128..135: one row per thread
                if (tid >= 2 * 8 * seg_rows) break;
                const int e = tid & 1, row = tid >> 1;
                it.x = x0 + 128 + 4 * e; it.col = 128 + 4 * e;
                it.wr = wr_first + row; it.y = Y0 + it.wr; it.nrows = 1;
            }
            if (edge) fh_phase1<true, SQ>(img, olut, V, O, g, rd, it);
            else fh_phase1<false, SQ>(img, olut, V, O, g, rd, it);
        }
        __syncthreads();
        // ---- phase 2: one thread per cell.  Lane <-> cell: lane bits 0,1 -> cell column bits 0,1; bit 3 -> column bit 2;
        // bit 2 -> column bit 3; bits 4.. -> cell row.  The four 16-lane groups of a ds_read_b128 ({0-3,12-15,20-27}, ...:
        // the lanes whose bits 2,3,4 have even / odd parity) then hold 8 distinct columns mod 8 x 2 cell rows of different
        // parity: 16 distinct 16-byte slots of V (column step 2 slots, cell-row step an odd number of slots) and of O.
        if ((tid >> 7) == role) {
            const int t = tid & 127;
            const int cc = (t & 3) | (((t >> 3) & 1) << 2) | (((t >> 2) & 1) << 3);
            const int cr = t >> 4;
            const int hr = 1 + FH_CR * k + cr, hc = hc0 + cc;
            float *bins = H + t;  // bin o at bins[o * 128]: lane <-> bank, whatever the orientations
#pragma unroll
            for (int o = 0; o < 18; o++) bins[o * FH_NCELL] = 0.f;
            int s8 = (8 * i) % 9 + cr;  // ring row group of the cell's first window row
            if (s8 >= 9) s8 -= 9;
            const unsigned tailbits = [&]() {
                unsigned m = 0;
                if (tail_window)
                    for (int kk = 0; kk < 16; kk++)
                        if (x0 + FH_CS * cc + kk >= g.body_end) m |= 1u << kk;
                return m;
            }();
#pragma unroll
            for (int half = 0; half < 2; half++) {
                const float *vrow = V + s8 * FH_VGROUP + FH_CS * cc;
                const unsigned short *orow = O + s8 * 8 * FH_PO + FH_CS * cc;
#pragma unroll
                for (int jj = 0; jj < 8; jj++) {
                    const float wy = fh_weight(8 * half + jj);
                    float v[16];
                    unsigned ow[8];
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const float4 a = *reinterpret_cast<const float4 *>(vrow + jj * FH_PV + 4 * q);
                        v[4 * q] = a.x; v[4 * q + 1] = a.y; v[4 * q + 2] = a.z; v[4 * q + 3] = a.w;
                    }
#pragma unroll
                    for (int q = 0; q < 2; q++) {
                        const uint4 a = *reinterpret_cast<const uint4 *>(orow + jj * FH_PO + 8 * q);
                        ow[4 * q] = a.x; ow[4 * q + 1] = a.y; ow[4 * q + 2] = a.z; ow[4 * q + 3] = a.w;
                    }
                    if (!tail_window) {
#pragma unroll
                        for (int kk = 0; kk < 16; kk++) {  // :863-870: vy * (vx * v)
                            const unsigned off = (kk & 1) ? (ow[kk >> 1] >> 16) : (ow[kk >> 1] & 0xffffu);
                            atomicAdd(reinterpret_cast<float *>(reinterpret_cast<char *>(bins) + off), wy * (fh_weight(kk) * v[kk]));
                        }
                    } else {
#pragma unroll
                        for (int kk = 0; kk < 16; kk++) {  // scalar tail, :951-954: (vy * vx) * v
                            const unsigned off = (kk & 1) ? (ow[kk >> 1] >> 16) : (ow[kk >> 1] & 0xffffu);
                            const float w = (tailbits >> kk) & 1u ? (wy * fh_weight(kk)) * v[kk] : wy * (fh_weight(kk) * v[kk]);
                            atomicAdd(reinterpret_cast<float *>(reinterpret_cast<char *>(bins) + off), w);
                        }
                    }
                }
                s8 = s8 + 1 == 9 ? 0 : s8 + 1;
            }
            if (hr <= g.cells_nr && hc <= g.cells_nc) {
                float b[18];
#pragma unroll
                for (int o = 0; o < 18; o++) b[o] = bins[o * FH_NCELL];
                float *dst = hist + (((size_t)f * (g.cells_nr + 2) + hr) * (g.cells_nc + 2) + hc) * 18;
#pragma unroll
                for (int o = 0; o < 18; o += 2) *reinterpret_cast<float2 *>(dst + o) = make_float2(b[o], b[o + 1]);
                float e = 0.f;
#pragma unroll
                for (int o = 0; o < 9; o++) e += (b[o] + b[o + 9]) * (b[o] + b[o + 9]);  // :959-968
                norm[((size_t)f * g.cells_nr + (hr - 1)) * g.cells_nc + (hc - 1)] = e;
            }
        }
        if (k + 1 < k1) __syncthreads();  // the next band overwrites ring rows this band's cells were reading
    }
}

__global__ void __launch_bounds__(512) fhog_build_olut(unsigned char *__restrict__ olut)
{
    const int tx = (int)threadIdx.x - 255, ty = (int)blockIdx.x - 255;
    olut[blockIdx.x * 512 + threadIdx.x] = (unsigned char)(tx <= 255 ? fhog_best_orientation(tx, ty) : 0);
}

template <int SQ>
__global__ void __launch_bounds__(256) fhog_sqrt_table(float *__restrict__ out, int n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = fh_sqrt<SQ>(i);
}

}  // namespace

bool fhog_fused_supported(const FhogGeom &g, const uint8_t *d_rgb, size_t frame_stride)
{
    return g.cs == FH_CS && g.cols % 4 == 0 && (size_t)d_rgb % 4 == 0 && frame_stride % 4 == 0;
}

imgfd_status fhog_fused_hist(imgfd_ctx *ctx, const uint8_t *d_rgb, size_t frame_stride, const FhogGeom &g, int nf, float *hist,
                             float *norm)
{
    if (!ctx->fhog_olut) {
        void *p = nullptr;
        if (hipMalloc(&p, FHOG_OLUT_BYTES) != hipSuccess) return imgfd_fail(ctx, IMGFD_ERR_OOM, "hipMalloc of the fHOG orientation table failed");
        ctx->fhog_olut = (unsigned char *)p;
        hipLaunchKernelGGL(fhog_build_olut, dim3(511), dim3(512), 0, ctx->stream, ctx->fhog_olut);
        IMGFD_HIP(ctx, hipFuncSetAttribute((const void *)fhog_hist8<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FH_LDS));
        IMGFD_HIP(ctx, hipFuncSetAttribute((const void *)fhog_hist8<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FH_LDS));
    }
    const int tiles_x = ceil_div(g.cells_nc, FH_CC), n_bands = ceil_div(g.cells_nr, FH_CR);
    int bpw = ctx->tune.fhog_bands;
    if (bpw <= 0) {  // march as far as the batch leaves >= 8 workgroups per CU
        const long band_tiles = (long)tiles_x * n_bands * nf;
        bpw = (int)std::min<long>(8, std::max<long>(1, band_tiles / (8L * ctx->num_cu)));
    }
    bpw = std::min(bpw, n_bands);
    const dim3 grid(tiles_x, ceil_div(n_bands, bpw), nf);
    if (ctx->tune.fhog_sqrt == 1)
        hipLaunchKernelGGL(fhog_hist8<1>, grid, dim3(FH_THREADS), FH_LDS, ctx->stream, d_rgb, frame_stride, ctx->fhog_olut, hist, norm, g, bpw);
    else
        hipLaunchKernelGGL(fhog_hist8<0>, grid, dim3(FH_THREADS), FH_LDS, ctx->stream, d_rgb, frame_stride, ctx->fhog_olut, hist, norm, g, bpw);
    IMGFD_HIP(ctx, hipGetLastError());
    return IMGFD_OK;
}

extern "C" {

// stage doorway (tests): the sqrt of phase 1 for every argument 0 .. n-1 (n <= 2 * 255^2 + 1); variant as tune.fhog_sqrt
imgfd_status imgfd_k_fhog_sqrt(imgfd_ctx *ctx, float *d_out, int n, int variant)
{
    if (!ctx || !d_out || n < 0) return IMGFD_ERR_INVALID;
    IMGFD_HIP(ctx, hipSetDevice(ctx->device));
    if (!n) return IMGFD_OK;
    if (variant == 1) hipLaunchKernelGGL(fhog_sqrt_table<1>, dim3(ceil_div(n, 256)), dim3(256), 0, ctx->stream, d_out, n);
    else hipLaunchKernelGGL(fhog_sqrt_table<0>, dim3(ceil_div(n, 256)), dim3(256), 0, ctx->stream, d_out, n);
    IMGFD_HIP(ctx, hipGetLastError());
    return IMGFD_OK;
}

}  // extern "C"
