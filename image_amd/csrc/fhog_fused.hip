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
//            Both are ONE table look-up: the packed (v, o) word is a pure function of the integer gradient (tx, ty), 511 x 511
//            possibilities, tabulated once per context by the reference's own float chain (fhog_best_orientation) and a
//            correctly rounded sqrtf -- instead of nine dot products and a square root per pixel.  Gradients of at most
//            16 grey levels per axis -- most pixels of a photograph -- are answered from a 4 KB copy of the table's centre
//            in LDS; the others gather from the 1 MB table in L2 (the gathers, not arithmetic, bound phase 1).
//   phase 2  one thread per histogram cell (8 x 16 cells) walks the 16 x 16 pixels that vote into it in raster order
//            and adds wy * (wx * v) into bin o of its private LDS histogram, four votes per LDS round trip (the bins of a
//            batch are read together; a vote whose bin an earlier vote of the batch touched continues from that sum).  LDS
//            operations of one wave execute in order and no other thread touches the cell, so every bin receives the
//            reference's summands in the reference's order.  With cell_size 8 the bilinear weights are the dyadic
//            constants (k + 0.5) / 8, exact in float, so they are compile-time literals.
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
constexpr int FH_NCELL = FH_CR * FH_CC;       // 128 cells = 128 phase-2 threads
constexpr size_t FH_LDS = (size_t)FH_VDW * 4;  // dynamic part (39 KB); + 9 KB of histograms + 4 KB of table: three workgroups per CU

// bilinear weight of window row / column k (0..15) of a cell: fhog.h:823-826, :838-841 with cell_size 8
__host__ __device__ constexpr float fh_weight(int k) { return k < 8 ? (k + 0.5f) / 8 : (15.5f - k) / 8; }

// A window pixel is ONE dword in LDS: v is 0 or lies in [1, 361), so its sign bit and the four high exponent bits are
// free: the word is v's float with the exponent field lowered by 126 (bits 23..26 hold 1..9; 0 for v = 0) and the
// orientation bin in bits 27..31.  Masked with FH_VMASK it reads as the float v * 2^-126 (a normal number, or +0);
// phase 2 multiplies it by column weights that carry the 2^126 -- a power of two moves no significand bit, so every
// product and sum is the reference's.  (6 bytes per pixel in two arrays allowed two workgroups per CU; the kernel is
// bound by how many cells a CU holds, profiles/r03.)
constexpr unsigned FH_VMASK = 0x07ffffffu, FH_EXP_SHIFT = 126u << 23;
constexpr float FH_2P126 = 0x1p126f;
constexpr int FH_CEN = 16;  // the LDS copy of the table covers gradients -16 .. 15 per axis (32 x 32 words)
__device__ __forceinline__ unsigned fh_pack(float v, unsigned o) { return (max(__float_as_uint(v), FH_EXP_SHIFT) - FH_EXP_SHIFT) | (o << 27); }

// The packed word of the gradient (tx, ty) without the table: the float chain of fhog_best_orientation decides like integer
// arithmetic does -- k = the number of the four boundaries between neighbouring directions of the first quadrant that
// (|tx|, |ty|) lies beyond (|ty| D_k > |tx| N_k with N_k / D_k = the differences of the 4-digit direction literals x 10^4),
// mirrored into the other quadrants; ties and zero gradients fall where the chain's strict comparisons put them.  Equal to
// the table for all 511 x 511 gradients (tests/test_fhog.py::test_fused_gradient_word_arithmetic_exhaustive).
__device__ __forceinline__ unsigned fh_word_arith(int tx, int ty, int len)
{
    const unsigned ax = (unsigned)abs(tx), ay = (unsigned)abs(ty);  // <= 255: the products fit 24 bits
    int k = (int)(ay * 3420u > ax * 603u) + (int)(ay * 3008u > ax * 1737u) + (int)(ay * 2232u > ax * 2660u) + (int)(ay * 1188u > ax * 3264u);
    const int o = ty >= 0 ? (tx >= 0 ? k : 9 - k) : (tx > 0 ? (k ? 18 - k : 0) : 9 + k);
    return fh_pack(sqrtf((float)len), (unsigned)o);
}

struct FhRows {
    int x;      // image column of the group's first pixel (multiple of 4, may lie outside the image)
    int col;    // its window column
    int y;      // first image row
    int wr;     // its window row (ring slot = wr mod FH_RING)
    int nrows;
    bool live;   // false: a lane that only keeps its wave whole (wave-wide vote below); it stores nothing
};

// phase 1 for one thread: `nrows` rows of one group of 4 pixels.  EDGE: loads clamped into the image, pixels outside
// [1, visible) vote 0, tail columns use the scalar colour rule.
template <bool EDGE>
__device__ __forceinline__ void fh_phase1(const unsigned *__restrict__ img, const unsigned *__restrict__ lut, const unsigned *lut_c,
                                          unsigned *V, const FhogGeom &g, int rd, const FhRows it, int arith_lanes)
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

    int slot = it.wr % FH_RING;
    // the words of a row are stored one row later: the look-ups stay in flight behind the next row's arithmetic
    unsigned pw[4] = {0u, 0u, 0u, 0u};
    int pslot = -1;
    auto flush = [&]() {
        unsigned *vdst = V + (pslot >> 3) * FH_VGROUP + (pslot & 7) * FH_PV + it.col;
        if (it.live) *reinterpret_cast<uint4 *>(vdst) = make_uint4(pw[0], pw[1], pw[2], pw[3]);
    };
    // one row of 4 pixels from the rows above / at / below it
    auto row = [&](const unsigned(&up)[5], const unsigned(&cen)[5], const unsigned(&dn)[5], int y) {
        const bool rowvalid = !EDGE || (y >= 1 && y < g.visible_nr);
        unsigned nw[4];
        int txs[4], tys[4], lens[4];
        unsigned span = 0;
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
            txs[p] = tx + FH_CEN; tys[p] = ty + FH_CEN; lens[p] = t;
            span |= (unsigned)txs[p] | (unsigned)tys[p];
        }
        // a lane whose four gradients lie inside the table's centre reads them from LDS (masked indices: always in bounds);
        // the other lanes -- edges, strong texture -- gather theirs from the table in L2.  ONE divergent region per row: a
        // gather's cost in the texture unit follows its number of active lanes, and the gathers bound phase 1.
#pragma unroll
        for (int p = 0; p < 4; p++)
            nw[p] = lut_c[((unsigned)tys[p] & (2u * FH_CEN - 1u)) * (2 * FH_CEN) + ((unsigned)txs[p] & (2u * FH_CEN - 1u))];
        const bool far = span >= 2u * FH_CEN && it.live;
        // A wave with at least arith_lanes such lanes (32) computes their words: the arithmetic (~40 VALU
        // instructions per pixel) costs the same for one lane or 64, a gather's cost in the texture unit follows its lanes.
        // Uniform noise, where every lane is outside: 112 -> 86.5 us per 4096^2 tile; the synthetic tile (75 us) takes the gathers.
        if (arith_lanes > 0 && (int)__popcll(__ballot(far)) >= arith_lanes) {
            if (far) {
#pragma unroll
                for (int p = 0; p < 4; p++) nw[p] = fh_word_arith(txs[p] - FH_CEN, tys[p] - FH_CEN, lens[p]);
            }
        } else if (far) {
#pragma unroll
            for (int p = 0; p < 4; p++) nw[p] = lut[(unsigned)((tys[p] + 255 - FH_CEN) * 512 + (txs[p] + 255 - FH_CEN))];
        }
        if (EDGE) {
#pragma unroll
            for (int p = 0; p < 4; p++)
                if (!(rowvalid && ((colvalid >> p) & 1u))) nw[p] = 0u;  // a zero vote leaves every sum unchanged
        }
        if (pslot >= 0) flush();
#pragma unroll
        for (int p = 0; p < 4; p++) pw[p] = nw[p];
        pslot = slot;
        slot = slot + 1 == FH_RING ? 0 : slot + 1;
    };
    // Four row buffers used cyclically: while row r is worked on (rows r-1, r, r+1), row r+3 is fetched into the buffer
    // row r-1 leaves -- two rows of arithmetic between a fetch and its first use, no register moves, no conditional fetch
    // (a fetch the compiler cannot count forces s_waitcnt vmcnt(0) on every row).  Fetches past the run re-read its last row.
    const int ylast = it.y + it.nrows;  // the row below the run's last row
    unsigned b0[5], b1[5], b2[5], b3[5];
    load(it.y - 1, b0);
    load(it.y, b1);
    load(it.y + 1, b2);
    load(min(it.y + 2, ylast), b3);
    int r = 0;
#pragma unroll 1
    for (; r + 4 <= it.nrows; r += 4) {
        row(b0, b1, b2, it.y + r);     load(min(it.y + r + 3, ylast), b0);
        row(b1, b2, b3, it.y + r + 1); load(min(it.y + r + 4, ylast), b1);
        row(b2, b3, b0, it.y + r + 2); load(min(it.y + r + 5, ylast), b2);
        row(b3, b0, b1, it.y + r + 3); load(min(it.y + r + 6, ylast), b3);
    }
#pragma unroll 1
    for (; r < it.nrows; r++) {
        row(b0, b1, b2, it.y + r);
#pragma unroll
        for (int k = 0; k < 5; k++) { b0[k] = b1[k]; b1[k] = b2[k]; b2[k] = b3[k]; }
        load(min(it.y + r + 3, ylast), b3);
    }
    flush();
}

// The 16 votes of one window row of a cell, in batches of 4.  A bin update is a read-modify-write of LDS (ds_add_f32 keeps
// the order too, but runs at ~3 cycles per LANE: 367 us per tile, profiles/r03): the four bins of a batch are read up
// front, a vote whose bin an earlier vote of the batch already updated continues from that vote's sum (the forwarding
// selects below) -- the reference's order of additions per bin -- and the four sums are stored in vote order, so the later
// of two stores to one bin stays.  One LDS round trip per 4 votes instead of per vote.
template <bool TAIL>
__device__ __forceinline__ void fh_vote_row(float *bins, const unsigned (&pw)[16], float wy, unsigned tailbits)
{
#pragma unroll
    for (int q = 0; q < 4; q++) {
        float *a[4];
        float w[4], r[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int kk = 4 * q + u;
            a[u] = reinterpret_cast<float *>(reinterpret_cast<char *>(bins) + ((pw[kk] >> 27) << 9));
            r[u] = *a[u];
            const float x = __uint_as_float(pw[kk] & FH_VMASK);  // v * 2^-126
            // :863-870: vy * (vx * v); scalar tail, :951-954: (vy * vx) * v
            w[u] = TAIL && ((tailbits >> kk) & 1u) ? (wy * fh_weight(kk) * FH_2P126) * x : wy * ((fh_weight(kk) * FH_2P126) * x);
        }
        const float s0 = r[0] + w[0];
        const float s1 = (a[1] == a[0] ? s0 : r[1]) + w[1];
        const float s2 = (a[2] == a[1] ? s1 : (a[2] == a[0] ? s0 : r[2])) + w[2];
        const float s3 = (a[3] == a[2] ? s2 : (a[3] == a[1] ? s1 : (a[3] == a[0] ? s0 : r[3]))) + w[3];
        *a[0] = s0; *a[1] = s1; *a[2] = s2; *a[3] = s3;
    }
}

// hist[(hr*HC + hc)*18 + o] for 1 <= hr <= cells_nr, 1 <= hc <= cells_nc (the layout of fhog_cell_hist) and the cell
// energies norm[(hr-1)*cells_nc + hc-1] (:959-968).  grid: (cell columns / 16, band groups, frames).
template <int NT>
__global__ void __launch_bounds__(NT) fhog_hist8(const unsigned char *__restrict__ rgb, size_t frame_stride,
                                                         const unsigned *__restrict__ lut, float *__restrict__ hist,
                                                         float *__restrict__ norm, FhogGeom g, int bands_per_wg, int tiles_x, int xcd_order,
                                                         int arith_lanes)
{
    HIP_DYNAMIC_SHARED(unsigned, lds)
    unsigned *V = lds;
    __shared__ unsigned lut_c[4 * FH_CEN * FH_CEN];
    for (int e = threadIdx.x; e < 4 * FH_CEN * FH_CEN; e += NT)
        lut_c[e] = lut[(e / (2 * FH_CEN) - FH_CEN + 255) * 512 + (e % (2 * FH_CEN) - FH_CEN + 255)];
    __syncthreads();
    __shared__ float H[18 * FH_NCELL];  // the cell histograms: an object of its own, so that the (v, o) reads of the next
                                        // window row may be scheduled across the bin stores

    const int tid = threadIdx.x;
    // grid = (windows of a frame, frames); XCD-aware order of the windows inside the frame (imgfd_xcd_tile): the windows left and
    // right of a window share the 128-byte lines at its edges with it
    const int in_frame = (int)(xcd_order ? imgfd_xcd_tile(blockIdx.x, gridDim.x) : blockIdx.x);
    const int wy = in_frame / tiles_x, wx = in_frame - wy * tiles_x;
    const int f = blockIdx.y;
    const unsigned *img = reinterpret_cast<const unsigned *>(rgb + (size_t)f * frame_stride);
    const int rd = 3 * g.cols / 4;
    const int n_bands = (g.cells_nr + FH_CR - 1) / FH_CR;
    const int k0 = wy * bands_per_wg, k1 = min(k0 + bands_per_wg, n_bands);
    const int hc0 = 1 + FH_CC * wx;
    const int x0 = FH_CS * hc0 - 12;        // image column of window column 0 (128 bx - 4)
    const int Y0 = FH_CS * (1 + FH_CR * k0) - 12;  // image row of window row 0 (64 k0 - 4)
    // the two waves that run phase 2 alternate between workgroups (a workgroup's waves go to the four SIMDs in turn)
    const int role = (wx + wy) & (NT / 128 - 1);
    const bool cols_inside = x0 >= 4 && x0 + FH_WX + 1 <= min(g.visible_nc, g.body_end);
    const bool tail_window = x0 + FH_WX > g.body_end;

    for (int k = k0; k < k1; k++) {
        const int i = k - k0;
        // ---- phase 1: (v, o) of the window rows that are not in the ring yet: rows 8..71 of the band's window (and rows 0..7
        // in the workgroup's first band).  The LDS window, not registers or wave slots, limits the workgroups per CU, so
        // the workgroup is wide: NT / 32 run segments of 64 * 32 / NT rows each over window columns 0..127, then one
        // row per thread for the leftovers (columns 128..135; rows 0..7 of a first band).
        constexpr int SEGS = NT / 32, SEG_ROWS = FH_NEW / SEGS;
        const int wr_main = FH_NEW * i + FH_CS;
        const int y_first = Y0 + (i ? wr_main : 0), y_end = Y0 + FH_NEW * i + FH_RING;
        const bool edge = !(cols_inside && y_first >= 1 && y_end <= g.visible_nr);
        {
            FhRows it;
            it.live = true;
            const int grp = tid & 31, seg = tid >> 5;
            it.x = x0 + 4 * grp; it.col = 4 * grp;
            it.wr = wr_main + seg * SEG_ROWS; it.y = Y0 + it.wr; it.nrows = SEG_ROWS;
            if (edge) fh_phase1<true>(img, lut, lut_c, V, g, rd, it, arith_lanes);
            else fh_phase1<false>(img, lut, lut_c, V, g, rd, it, arith_lanes);
            const int n_left = 2 * FH_NEW + (i ? 0 : FH_CS * (FH_WX / 4));
#pragma unroll 1
            for (int first = 0; first < n_left; first += NT) {
                it.live = first + tid < n_left;
                const int item = min(first + tid, n_left - 1);
                if (item < 2 * FH_NEW) { it.col = 128 + 4 * (item & 1); it.wr = wr_main + (item >> 1); }
                else { const int u = item - 2 * FH_NEW; it.col = 4 * (u % (FH_WX / 4)); it.wr = u / (FH_WX / 4); }
                it.x = x0 + it.col; it.y = Y0 + it.wr; it.nrows = 1;
                if (edge) fh_phase1<true>(img, lut, lut_c, V, g, rd, it, arith_lanes);
                else fh_phase1<false>(img, lut, lut_c, V, g, rd, it, arith_lanes);
            }
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
                const unsigned *vrow = V + s8 * FH_VGROUP + FH_CS * cc;
#pragma unroll
                for (int jj = 0; jj < 8; jj++) {
                    const float wy = fh_weight(8 * half + jj);
                    unsigned pw[16];
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const uint4 a = *reinterpret_cast<const uint4 *>(vrow + jj * FH_PV + 4 * q);
                        pw[4 * q] = a.x; pw[4 * q + 1] = a.y; pw[4 * q + 2] = a.z; pw[4 * q + 3] = a.w;
                    }
                    if (!tail_window) fh_vote_row<false>(bins, pw, wy, 0u);
                    else fh_vote_row<true>(bins, pw, wy, tailbits);
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

// lut[(ty + 255) * 512 + tx + 255] = the packed word of the gradient (tx, ty): v = sqrtf(tx^2 + ty^2) (the compiler's correctly
// rounded sqrtf, fhog.h:871 / :929), o by the reference's float chain
// (arith: the same words from fh_word_arith -- the doorway imgfd_k_fhog_lut_arith, for the exhaustive comparison)
__global__ void __launch_bounds__(512) fhog_build_lut(unsigned *__restrict__ lut, int arith)
{
    const int tx = (int)threadIdx.x - 255, ty = (int)blockIdx.x - 255;
    unsigned w = 0;
    if (tx <= 255) w = arith ? fh_word_arith(tx, ty, tx * tx + ty * ty) : fh_pack(sqrtf((float)(tx * tx + ty * ty)), (unsigned)fhog_best_orientation(tx, ty));
    lut[blockIdx.x * 512 + threadIdx.x] = w;
}

}  // namespace

bool fhog_fused_supported(const FhogGeom &g, const uint8_t *d_rgb, size_t frame_stride)
{
    return g.cs == FH_CS && g.cols % 4 == 0 && (size_t)d_rgb % 4 == 0 && frame_stride % 4 == 0;
}

// first use on a context: the gradient table and the kernels' LDS attribute.  The context keeps the table only when ALL of
// it succeeded -- a failure leaves the context as it was, and the next call tries again.
static imgfd_status fhog_fused_init(imgfd_ctx *ctx)
{
    if (ctx->fhog_lut) return IMGFD_OK;
    void *p = nullptr;
    if (hipMalloc(&p, FHOG_LUT_BYTES) != hipSuccess) return imgfd_fail(ctx, IMGFD_ERR_OOM, "hipMalloc of the fHOG gradient table failed");
    hipLaunchKernelGGL(fhog_build_lut, dim3(511), dim3(512), 0, ctx->stream, (unsigned *)p, 0);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipFuncSetAttribute((const void *)fhog_hist8<256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FH_LDS);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void *)fhog_hist8<512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FH_LDS);
    if (e != hipSuccess) {
        (void)hipStreamSynchronize(ctx->stream);  // the build kernel may be running on the table
        (void)hipFree(p);
        return imgfd_fail(ctx, IMGFD_ERR_HIP, (std::string("fHOG: building the gradient table failed: ") + hipGetErrorString(e)).c_str());
    }
    ctx->fhog_lut = (unsigned *)p;
    return IMGFD_OK;
}

imgfd_status fhog_fused_hist(imgfd_ctx *ctx, const uint8_t *d_rgb, size_t frame_stride, const FhogGeom &g, int nf, float *hist,
                             float *norm)
{
    IMGFD_TRY(fhog_fused_init(ctx));
    const int tiles_x = ceil_div(g.cells_nc, FH_CC), n_bands = ceil_div(g.cells_nr, FH_CR);
    int bpw = ctx->tune.fhog_bands;
    if (bpw <= 0) {  // march as far as the batch leaves >= 8 workgroups per CU, four bands at most (16 tiles of 4096^2: 1 / 2 / 4 / 8 / 16
                     // bands per workgroup 76.5 / 73.2 / 72.8 / 74.4 / 77.1 us per tile, profiles/r05/a_fhog_variants.txt; round 4 the same order)
        const long band_tiles = (long)tiles_x * n_bands * nf;
        bpw = (int)std::min<long>(4, std::max<long>(1, band_tiles / (8L * ctx->num_cu)));
    }
    bpw = std::min(bpw, n_bands);
    const dim3 grid((unsigned)tiles_x * (unsigned)ceil_div(n_bands, bpw), nf);
    // (512-thread workgroups: 83.8 against 77.6 us per tile, round 4; a wave with at least 32 lanes outside the table's LDS centre computes
    // its words instead of gathering them: thresholds 16-64 measure the same, profiles/r04/fhog_arith.txt)
    hipLaunchKernelGGL(fhog_hist8<256>, grid, dim3(256), FH_LDS, ctx->stream, d_rgb, frame_stride, ctx->fhog_lut, hist, norm, g, bpw, tiles_x, 1, 32);
    IMGFD_HIP(ctx, hipGetLastError());
    return IMGFD_OK;
}

extern "C" {

// stage doorway (tests): the gradient table of the fused kernel, 511 x 512 packed words (built on first use)
imgfd_status imgfd_k_fhog_lut(imgfd_ctx *ctx, uint32_t *d_out)
{
    if (!ctx || !d_out) return IMGFD_ERR_INVALID;
    IMGFD_HIP(ctx, hipSetDevice(ctx->device));
    IMGFD_TRY(fhog_fused_init(ctx));
    IMGFD_HIP(ctx, hipMemcpyAsync(d_out, ctx->fhog_lut, FHOG_LUT_BYTES, hipMemcpyDeviceToDevice, ctx->stream));
    return IMGFD_OK;
}

// the same 511 x 512 words computed by fh_word_arith (integer orientation rule): must equal imgfd_k_fhog_lut's bit for bit
imgfd_status imgfd_k_fhog_lut_arith(imgfd_ctx *ctx, uint32_t *d_out)
{
    if (!ctx || !d_out) return IMGFD_ERR_INVALID;
    IMGFD_HIP(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(fhog_build_lut, dim3(511), dim3(512), 0, ctx->stream, d_out, 1);
    IMGFD_HIP(ctx, hipGetLastError());
    return IMGFD_OK;
}

}  // extern "C"
