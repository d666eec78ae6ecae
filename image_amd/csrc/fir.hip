// image_amd/csrc/fir.hip -- separable double-accumulated Gaussian FIR for gfx950 (K1 and K3).
//
// Replaces discrete_gaussian(), image.CornerDetectionHarris/src/gaussian.cpp:289-395, and the pass
// that dominates the reference's run time, compute_autocorrelation_matrix(), harris.cpp:44-70
// (products Ix*Ix, Ix*Iy, Iy*Iy in float, then three in-place Gaussians).
//
// Numerics contract (SURVEY.md 7 "hard part 1", Appendix A):
//   - taps B[0..size-1] are computed on the host with the reference's literal expressions (fir_coeffs)
//   - each 1-D pass widens float -> double, forms  B[0]*R[i] + sum_j B[j]*(R[i-j]+R[i+j])  with j
//     ascending, pair added first, and rounds to float ONCE per pass (gaussian.cpp:351-359, 382-390)
//   - borders: left/top logical -k -> k, right/bottom logical n-1+k -> n-k (gaussian.cpp:345-349)
//   - FMA=false: the reference's exact operation sequence (the library is built -ffp-contract=off);
//     FMA=true: sum = fma(B[j], pair, sum) in the f64 accumulation only.
//
// Kernel shape (fir_march): one 256-thread workgroup owns a TW-column strip of one frame and marches
// down a segment of rows in chunks of CH rows.  Per chunk: (1) coalesced global loads of CH rows x
// (TW+2R) columns into LDS (the next chunk is prefetched into registers while this one computes);
// (2) row pass: a thread owns 8 consecutive pixels of one row, reads its 8+2R window with
// ds_read_b128 from a bank-swizzled tile, converts once to f64 and slides the window in registers;
// results (rounded to float) go to an LDS ring of row-filtered rows; (3) column pass: a thread owns
// one column and 8 consecutive output rows, reads its 8+2R window from the ring (lane <-> column:
// conflict-free), and stores 256-byte coalesced row segments.  Row-filtered rows are computed once per
// segment (2R halo rows per segment, not per tile), no intermediate plane ever leaves the CU: HBM
// traffic is the algorithmic 8 B read + 12 B written per pixel (plus the strip halo, served by L2).
#include "common.h"
#include "fir_device.h"

#include <math.h>


#ifdef FIR_PROFILE
// experiment build only (make EXTRA=-DFIR_PROFILE): per-phase shader-clock sums over all waves of fir_march
__device__ unsigned long long g_fir_prof[8];
#define FIR_T(i)                                                   \
    do {                                                           \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
        if ((threadIdx.x & 63) == 0) prof[i] += t_ - tlast;        \
        tlast = t_;                                                \
    } while (0)
extern "C" __attribute__((visibility("default"))) int imgfd_debug_fir_profile(unsigned long long *out, int reset)
{
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fir_prof), sizeof(unsigned long long) * 8) != hipSuccess) return 1;
    if (reset) {
        unsigned long long z[8] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_fir_prof), z, sizeof z) != hipSuccess) return 1;
    }
    return 0;
}
#else
#define FIR_T(i)
#endif

// native 16-byte vector (gcc and clang): unlike HIP's float4 struct it is always promoted to registers
typedef float fir_v4f __attribute__((vector_size(16)));

struct FirParams {
    const void *in0;
    const void *in1;
    float *out0;
    float *out1;
    float *out2;
    int nx, ny;
    int in_pitch;           // elements per input row
    long in_frame_stride;   // elements between input frames
    long out_frame_stride;  // elements between output frames (output pitch is nx)
    int seg_rows;           // output rows per workgroup segment
    int xcd_remap;          // 1: remap workgroup ids so an XCD (id % 8) owns a contiguous run of tiles
    int vec4;               // 1: input planes, pitch and frame stride are 16-byte aligned (float4 tile loads)
    double B[IMGFD_MAX_TAPS];
};

constexpr int fir_ring_size(int need)
{
    int r = 1;
    while (r < need) r *= 2;
    return r;
}
// float4 slots per LDS tile row: W4 data slots, +1 slot of skew per 16 slots, rounded to 16 slots
// (= 64 dwords) so that 16 lanes x ds_read_b128 at stride 32 B land on 16 distinct 4-bank groups.
constexpr int fir_rpitch4(int w4) { return ((w4 + ((w4 - 1) >> 4)) + 15) / 16 * 16; }
__device__ __forceinline__ int fir_swz4(int q) { return q + (q >> 4); }

// PX = consecutive outputs per thread and pass (4 or 8), NT = threads per workgroup (CH * TW / PX)
template <int R, int MODE, int TW, int CH, int FIR_PX = 8, int FIR_NT = 256>
struct FirGeom {
    static constexpr int NI = 1, NP = 1;  // input / output planes (the two-in three-out structure tensor lives in fir_tensor.hip)
    static constexpr int HALO = (R + 3) / 4 * 4;  // tile halo in whole float4 slots: x0-HALO is 16-byte aligned
    static constexpr int W = TW + 2 * HALO;
    static constexpr int W4 = W / 4;
    static constexpr int RP4 = fir_rpitch4(W4);
    static constexpr int RPITCH = RP4 * 4;
    static constexpr int RING = fir_ring_size(CH + 2 * R);
    static constexpr int NW = FIR_PX + 2 * R;            // window length
    static constexpr int S0 = (HALO - R) / 4;            // first float4 slot of strip 0's window
    static constexpr int OFF = (HALO - R) % 4;           // window start inside that slot
    static constexpr int NW4 = (OFF + NW + 3) / 4;       // float4 reads per window
    static constexpr int NL4 = (CH * W4 + FIR_NT - 1) / FIR_NT;  // float4 tile slots per thread per chunk
    static constexpr size_t LDS_BYTES = sizeof(float) * ((size_t)NI * CH * RPITCH + (size_t)NP * RING * TW);
};

// MODE 0: one f32 plane in -> one plane out;  MODE 1: one u8 plane in
template <int R, int MODE, int TW, int CH, bool FMA, int FIR_PX, int FIR_NT, bool VEC>
// LDS fixes the residency of this kernel at two workgroups per CU (IMGFD_WAVES_PER_EU, common.h)
__global__ void __launch_bounds__(FIR_NT) IMGFD_WAVES_PER_EU(FIR_NT == 192 ? 3 : FIR_NT / 128, FIR_NT == 192 ? 3 : FIR_NT / 128) fir_march(FirParams p)
{
    using G = FirGeom<R, MODE, TW, CH, FIR_PX, FIR_NT>;
    constexpr int NI = G::NI, NP = G::NP, W4 = G::W4, RPITCH = G::RPITCH, RING = G::RING, HALO = G::HALO;
    constexpr int NW = G::NW, NW4 = G::NW4, NL4 = G::NL4, S0 = G::S0, OFF = G::OFF;
    constexpr int STRIPS = TW / FIR_PX;

    HIP_DYNAMIC_SHARED(float4, smem4)
    float4 *raw4 = smem4;                                                   // [NI][CH][RP4], slots swizzled
    float *ring = reinterpret_cast<float *>(smem4) + NI * CH * RPITCH;      // [NP][RING][TW]

    const int tid = threadIdx.x;
    int bx = blockIdx.x, by = blockIdx.y;
    if (p.xcd_remap) {
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
    const int y1 = min(p.ny, y0 + p.seg_rows);
    const int nrows = y1 - y0;
    const int nchunks = (nrows + 2 * R + CH - 1) / CH;
    const int ybase = y0 - R;

    const float *in0f = reinterpret_cast<const float *>(p.in0) + (size_t)frame * p.in_frame_stride;
    const unsigned char *in0b = reinterpret_cast<const unsigned char *>(p.in0) + (size_t)frame * p.in_frame_stride;
    float *outp[1] = {p.out0 + (size_t)frame * p.out_frame_stride};

    // tile slot owned by this thread in load round l: row tr[l], float4 slot tq[l] (constant over chunks)
    int tr[NL4], tq[NL4];
    unsigned xoff[NL4];
#pragma unroll
    for (int l = 0; l < NL4; l++) {
        const int i = tid + l * FIR_NT;
        tr[l] = i / W4;
        tq[l] = i - tr[l] * W4;
        // VEC: every slot is fetched as one aligned float4 from an in-range address; slots that hang over the
        // left/right image border fetch a neighbouring quad and are rewritten in LDS by patch_borders()
        xoff[l] = (unsigned)min(max(x0 - HALO + 4 * tq[l], 0), max(p.nx - 4, 0));
    }
    const bool x_inside = x0 - HALO >= 0 && x0 - HALO + G::W <= p.nx;

    fir_v4f pre0[NL4];

    // ---- issue the global loads of one chunk into registers.  One straight-line sequence of loads: no
    // control-flow merge may touch pre0 before commit(), or the loads stop being asynchronous.
    auto prefetch = [&](int chunk) __attribute__((always_inline)) {
        const int yc = ybase + chunk * CH;
#pragma unroll
        for (int l = 0; l < NL4; l++) {
            if (l < NL4 - 1 || tid + l * FIR_NT < CH * W4) {
                const int gy = fir_reflect(yc + tr[l], p.ny);
                if (VEC) {
                    const unsigned off = (unsigned)gy * (unsigned)p.in_pitch + xoff[l];
                    pre0[l] = *reinterpret_cast<const fir_v4f *>(in0f + off);
                } else {
                    float v0[4];
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const int gx = fir_reflect(x0 - HALO + 4 * tq[l] + e, p.nx);
                        const size_t off = (size_t)gy * p.in_pitch + gx;
                        if (MODE == 1) v0[e] = (float)in0b[off];
                        else v0[e] = in0f[off];
                    }
                    pre0[l] = fir_v4f{v0[0], v0[1], v0[2], v0[3]};
                }
            }
        }
    };
    auto commit = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int l = 0; l < NL4; l++) {
            if (l < NL4 - 1 || tid + l * FIR_NT < CH * W4) {
                reinterpret_cast<fir_v4f *>(raw4)[(0 * CH + tr[l]) * G::RP4 + fir_swz4(tq[l])] = pre0[l];
            }
        }
        if (VEC && !x_inside) {
            // border strips only (workgroup-uniform): rebuild the reflected halo columns from the columns
            // of the same LDS row.  left: x = -k -> k;  right: x = nx-1+k -> nx-k  (gaussian.cpp:345-349)
            __syncthreads();
            float *rawf = reinterpret_cast<float *>(raw4);
            for (int i = tid; i < NI * CH * 2 * HALO; i += FIR_NT) {
                const int h = i % (2 * HALO), rr = i / (2 * HALO);  // rr = plane*CH + row
                int c, x;
                if (h < HALO) { c = h; x = x0 - HALO + c; if (x >= 0) continue; }
                else { x = p.nx + (h - HALO); c = x - x0 + HALO; if (c >= G::W) continue; }
                const int sx = fir_reflect(x, p.nx);
                const int sc = sx - x0 + HALO;
                if (sc < 0 || sc >= G::W) continue;
                rawf[(rr * G::RP4 + fir_swz4(c >> 2)) * 4 + (c & 3)] = rawf[(rr * G::RP4 + fir_swz4(sc >> 2)) * 4 + (sc & 3)];
            }
        }
    };

    // Software pipeline.  The loads of chunk c+1 are issued before the row pass of chunk c and written to LDS
    // right after it (the raw tile is free once every thread has passed the barrier that ends the row pass),
    // i.e. BEFORE the column pass issues its stores: s_waitcnt vmcnt counts stores too, so waiting for the
    // prefetched tile at the top of the next iteration would also wait for the stores just issued.
    prefetch(0);
    FIR_T(0);
    commit();
    for (int chunk = 0; chunk < nchunks; chunk++) {
        FIR_T(1);
        __syncthreads();
        FIR_T(2);
        prefetch(chunk + 1);  // unconditional (the tile past the last chunk is fetched from clamped rows and never used):
        FIR_T(3);             // a branch here would make the waitcnt bookkeeping of the loop conservative

        // ---- row pass: raw tile -> ring of row-filtered rows
        for (int item = tid; item < CH * STRIPS; item += FIR_NT) {
            const int r = item / STRIPS, s = item - r * STRIPS;
            float wx[NW4 * 4];
            const float4 *rx = raw4 + (0 * CH + r) * G::RP4;
#pragma unroll
            for (int q = 0; q < NW4; q++) {
                const float4 v = rx[fir_swz4((FIR_PX / 4) * s + S0 + q)];
                wx[4 * q] = v.x; wx[4 * q + 1] = v.y; wx[4 * q + 2] = v.z; wx[4 * q + 3] = v.w;
            }
            const int slot = (chunk * CH + r) & (RING - 1);
#pragma unroll
            for (int pl = 0; pl < NP; pl++) {
                double d[NW];
#pragma unroll
                for (int k = 0; k < NW; k++) {
                    d[k] = (double)wx[OFF + k];
                }
                float o[FIR_PX];
                fir_window8<R, FMA, FIR_PX>(d, p.B, o);
                float4 *dst = reinterpret_cast<float4 *>(ring + (pl * RING + slot) * TW + FIR_PX * s);
#pragma unroll
                for (int h = 0; h < FIR_PX / 4; h++) dst[h] = make_float4(o[4 * h], o[4 * h + 1], o[4 * h + 2], o[4 * h + 3]);
            }
        }
        FIR_T(4);
        __syncthreads();
        FIR_T(5);
        commit();

        // ---- column pass: ring -> global
        for (int item = tid; item < TW * (CH / FIR_PX); item += FIR_NT) {
            const int g = item / TW, col = item - g * TW;
            const int oi0 = chunk * CH - 2 * R + FIR_PX * g;
            const int gx = x0 + col;
            if (oi0 + FIR_PX <= 0 || oi0 >= nrows || gx >= p.nx) continue;
            const bool full = oi0 >= 0 && oi0 + FIR_PX <= nrows;
            const unsigned obase = (unsigned)(y0 + oi0) * (unsigned)p.nx + (unsigned)gx;  // fits: one frame < 2^32 px
#pragma unroll
            for (int pl = 0; pl < NP; pl++) {
                double d[NW];
#pragma unroll
                for (int k = 0; k < NW; k++)
                    d[k] = (double)ring[(pl * RING + ((oi0 + k) & (RING - 1))) * TW + col];
                float o[FIR_PX];
                fir_window8<R, FMA, FIR_PX>(d, p.B, o);
                if (full) {
#pragma unroll
                    for (int k = 0; k < FIR_PX; k++) outp[pl][obase + (unsigned)k * (unsigned)p.nx] = o[k];
                } else {
#pragma unroll
                    for (int k = 0; k < FIR_PX; k++) {
                        const int oi = oi0 + k;
                        if (oi >= 0 && oi < nrows) outp[pl][(size_t)(y0 + oi) * p.nx + gx] = o[k];
                    }
                }
            }
        }
        FIR_T(6);
        // the next row pass writes ring slots only after the barrier at the top of the loop, by which time
        // every thread has left this column pass.
    }
#ifdef FIR_PROFILE
    if ((threadIdx.x & 63) == 0)
        for (int i = 0; i < 8; i++) atomicAdd(&g_fir_prof[i], prof[i]);
#endif
}

// ------------------------------------------------------------------ generic fallback (any radius)
// One thread per output pixel, taps from kernarg memory; used for sigmas whose radius has no
// specialised instantiation.  Same operation order, same borders.
struct FirGenericParams {
    const float *in;
    float *out;
    int nx, ny, size;  // size = taps B[0..size-1]
    int horizontal;
    int fma;
    long frame_stride;
    const double *Bg;  // taps in memory when there are more than IMGFD_MAX_TAPS (nullptr: B below)
    double B[IMGFD_MAX_TAPS];
};

__global__ void __launch_bounds__(256) fir_generic_pass(FirGenericParams p)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= p.nx) return;
    const float *in = p.in + (size_t)blockIdx.z * p.frame_stride;
    float *out = p.out + (size_t)blockIdx.z * p.frame_stride;
    const int n = p.horizontal ? p.nx : p.ny;
    const int c = p.horizontal ? x : y;
    auto at = [&](int i) -> double {
        const int r = fir_reflect(i, n);
        return (double)(p.horizontal ? in[(size_t)y * p.nx + r] : in[(size_t)r * p.nx + x]);
    };
    const double *B = p.Bg ? p.Bg : p.B;
    double sum = B[0] * at(c);
    for (int j = 1; j < p.size; j++) {
        const double pair = at(c - j) + at(c + j);
        if (p.fma) sum = __builtin_fma(B[j], pair, sum);
        else sum += B[j] * pair;
    }
    out[(size_t)y * p.nx + x] = (float)sum;
}

__global__ void __launch_bounds__(256) tensor_products(const float *Ix, const float *Iy, float *A, float *B,
                                                       float *C, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float ix = Ix[i], iy = Iy[i];
    A[i] = ix * ix;
    B[i] = ix * iy;
    C[i] = iy * iy;
}

__global__ void __launch_bounds__(256) plane_copy(const void *in, int in_is_u8, int in_pitch,
                                                  size_t in_frame_stride, float *out, int nx, int ny)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= nx) return;
    const size_t src = (size_t)blockIdx.z * in_frame_stride + (size_t)y * in_pitch + x;
    const float v = in_is_u8 ? (float)reinterpret_cast<const unsigned char *>(in)[src]
                             : reinterpret_cast<const float *>(in)[src];
    out[((size_t)blockIdx.z * ny + y) * nx + x] = v;
}

// ------------------------------------------------------------------ host side
// taps exactly as gaussian.cpp:307-330 (den in float, integer -i*i, pi = 3.1415926)
int fir_size(float sigma, int precision) { return (int)(precision * sigma) + 1; }

static void fir_coeffs_n(float sigma, int size, double *B)
{
    double den = 2 * sigma * sigma;
    for (int i = 0; i < size; i++) B[i] = 1 / (sigma * sqrt(2.0 * 3.1415926)) * exp(-i * i / den);
    double norm = 0;
    for (int i = 0; i < size; i++) norm += B[i];
    norm *= 2;
    norm -= B[0];
    for (int i = 0; i < size; i++) B[i] /= norm;
}

int fir_coeffs(float sigma, int precision, double *B)
{
    const int size = fir_size(sigma, precision);
    if (size > IMGFD_MAX_TAPS) return -1;
    fir_coeffs_n(sigma, size, B);
    return size;
}

// more taps than a kernel argument holds (the reference stops only at size > xdim, gaussian.cpp:312): the taps go to a
// device buffer the context keeps.  Rare and not on any timed path: the copy is waited for.
static imgfd_status fir_taps_device(imgfd_ctx *ctx, float sigma, int size, const double **d_B)
{
    std::vector<double> B((size_t)size);
    fir_coeffs_n(sigma, size, B.data());
    if ((size_t)size > ctx->taps_cap) {
        IMGFD_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->taps_dev) (void)hipFree(ctx->taps_dev);
        ctx->taps_dev = nullptr; ctx->taps_cap = 0;
        void *p = nullptr;
        const size_t cap = align_up((size_t)size, 1024);
        if (hipMalloc(&p, cap * sizeof(double)) != hipSuccess) return imgfd_fail(ctx, IMGFD_ERR_OOM, "hipMalloc of the Gaussian taps failed");
        ctx->taps_dev = (double *)p; ctx->taps_cap = cap;
    }
    IMGFD_HIP(ctx, hipMemcpyAsync(ctx->taps_dev, B.data(), sizeof(double) * (size_t)size, hipMemcpyHostToDevice, ctx->stream));
    IMGFD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *d_B = ctx->taps_dev;
    return IMGFD_OK;
}

template <int R, int MODE, int TW, int CH, int FIR_PX = 8, int FIR_NT = 256>
static imgfd_status launch_march(imgfd_ctx *ctx, FirParams &p, int n_frames)
{
    using G = FirGeom<R, MODE, TW, CH, FIR_PX, FIR_NT>;
    const int strips = ceil_div(p.nx, TW);
    // Segment length: a workgroup walks (rows + 2R) rows in chunks of CH.  With `slots` workgroups resident on the
    // chip, the pass takes ceil(workgroups / slots) rounds of (chunks per segment) steps: pick the segment count
    // that minimises that product (ties: fewer, longer segments = less halo work).
    static int per_cu = 0;
    if (!per_cu) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void *)fir_march<R, MODE, TW, CH, true, FIR_PX, FIR_NT, MODE != 1>, FIR_NT,
                                                         G::LDS_BYTES) != hipSuccess || n < 1)
            n = 2;
        per_cu = n;
    }
    const long slots = (long)per_cu * ctx->num_cu;
    long best_cost = -1;
    int seg = p.ny;
    for (int nseg = 1; nseg <= ceil_div(p.ny, CH); nseg++) {
        int m = ceil_div(ceil_div(p.ny, nseg) + 2 * R, CH);
        if (m < 2) m = 2;
        const int sr = m * CH - 2 * R;  // (rows + 2R) fills whole chunks
        const long wgs = (long)strips * ceil_div(p.ny, sr) * n_frames;
        const long cost = ((wgs + slots - 1) / slots) * m;
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; seg = sr; }
    }
    p.seg_rows = seg;
    dim3 grid(strips, ceil_div(p.ny, seg), n_frames);
    p.xcd_remap = 1;
    // float4 tile loads need 16-byte aligned planes, pitch and frame stride, and whole quads per row
    p.vec4 = MODE != 1 && ((size_t)p.in0 % 16 == 0) && ((size_t)p.in1 % 16 == 0) && p.in_pitch % 4 == 0 &&
             p.in_frame_stride % 4 == 0 && p.nx % 4 == 0 && p.nx >= 4;
    const size_t lds = G::LDS_BYTES;
    auto go = [&](auto kern) -> imgfd_status {
        IMGFD_HIP(ctx, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, grid, dim3(FIR_NT), lds, ctx->stream, p);
        IMGFD_HIP(ctx, hipGetLastError());
        return IMGFD_OK;
    };
    if (ctx->fir_mode) {
        if (p.vec4) return go(fir_march<R, MODE, TW, CH, true, FIR_PX, FIR_NT, MODE != 1>);
        return go(fir_march<R, MODE, TW, CH, true, FIR_PX, FIR_NT, false>);
    }
    if (p.vec4) return go(fir_march<R, MODE, TW, CH, false, FIR_PX, FIR_NT, MODE != 1>);
    return go(fir_march<R, MODE, TW, CH, false, FIR_PX, FIR_NT, false>);
}

static imgfd_status launch_generic(imgfd_ctx *ctx, const float *in, float *tmp, float *out, int nx, int ny,
                                   int n_frames, int size, const double *B, const double *d_B = nullptr)
{
    FirGenericParams g;
    memset(&g, 0, sizeof g);
    g.nx = nx; g.ny = ny; g.size = size; g.fma = ctx->fir_mode; g.frame_stride = (long)nx * ny;
    g.Bg = d_B;
    if (!d_B) memcpy(g.B, B, sizeof(double) * size);
    dim3 grid(ceil_div(nx, 256), ny, n_frames);
    g.in = in; g.out = tmp; g.horizontal = 1;
    hipLaunchKernelGGL(fir_generic_pass, grid, dim3(256), 0, ctx->stream, g);
    g.in = tmp; g.out = out; g.horizontal = 0;
    hipLaunchKernelGGL(fir_generic_pass, grid, dim3(256), 0, ctx->stream, g);
    IMGFD_HIP(ctx, hipGetLastError());
    return IMGFD_OK;
}

size_t gaussian_tmp_bytes(int nx, int ny, int n_frames, float sigma, int type, int planes)
{
    (void)planes;
    if (type == IMGFD_FAST_GAUSSIAN) return sizeof(float) * sii_scratch_floats(nx, ny, sigma) * n_frames;  // cumulative sums
    if (type != IMGFD_STD_GAUSSIAN || sigma <= 0) return 0;
    return sizeof(float) * (size_t)nx * ny * n_frames;  // generic two-pass scratch (unused on the marching fast paths)
}

// gaussian(): gaussian.cpp:403-430.  d_in may be u8 or f32 with its own pitch; d_out is a packed
// f32 plane; d_tmp (nx*ny*n_frames floats) is needed for the generic/SII paths only.
imgfd_status launch_gaussian(imgfd_ctx *ctx, const void *d_in, int in_is_u8, int in_pitch,
                             size_t in_frame_stride, float *d_out, int nx, int ny, int n_frames,
                             float sigma, int type, float *d_tmp)
{
    dim3 cgrid(ceil_div(nx, 256), ny, n_frames);
    auto copy = [&]() -> imgfd_status {
        if (d_in == (const void *)d_out) return IMGFD_OK;
        hipLaunchKernelGGL(plane_copy, cgrid, dim3(256), 0, ctx->stream, d_in, in_is_u8, in_pitch,
                           in_frame_stride, d_out, nx, ny);
        IMGFD_HIP(ctx, hipGetLastError());
        return IMGFD_OK;
    };
    if (type == IMGFD_FAST_GAUSSIAN) {
        if (!(sigma > 0)) return copy();
        if (in_is_u8 || in_pitch != nx || in_frame_stride != (size_t)nx * ny) {
            // widen / repack into the output plane, then filter it in place (the cumulative sums live in d_tmp)
            hipLaunchKernelGGL(plane_copy, cgrid, dim3(256), 0, ctx->stream, d_in, in_is_u8, in_pitch,
                               in_frame_stride, d_out, nx, ny);
            return launch_sii_gaussian(ctx, d_out, d_out, nx, ny, n_frames, sigma, d_tmp);
        }
        return launch_sii_gaussian(ctx, (const float *)d_in, d_out, nx, ny, n_frames, sigma, d_tmp);
    }
    if (type != 0 || sigma <= 0) return copy();  // NO_GAUSSIAN / sigma<=0: gaussian.cpp:299-305, 424-429
    FirParams p;
    memset(&p, 0, sizeof p);
    const int size = fir_size(sigma, 3);
    if (size > nx) return copy();  // gaussian.cpp:312: output untouched == input (the reference works in place)
    const double *d_B = nullptr;
    if (size > IMGFD_MAX_TAPS) IMGFD_TRY(fir_taps_device(ctx, sigma, size, &d_B));
    else (void)fir_coeffs(sigma, 3, p.B);
    const int R = size - 1;
    p.in0 = d_in; p.in1 = nullptr; p.out0 = d_out; p.nx = nx; p.ny = ny; p.in_pitch = in_pitch;
    p.in_frame_stride = (long)in_frame_stride; p.out_frame_stride = (long)nx * ny;
    if (R == 3 && !(d_in == (const void *)d_out)) {
        if (in_is_u8) return launch_march<3, 1, 128, 16>(ctx, p, n_frames);
        return launch_march<3, 0, 128, 16>(ctx, p, n_frames);
    }
    if (!d_tmp) return imgfd_fail(ctx, IMGFD_ERR_INVALID, "generic gaussian needs a scratch plane");
    const float *src = (const float *)d_in;
    if (in_is_u8 || in_pitch != nx || in_frame_stride != (size_t)nx * ny) {
        hipLaunchKernelGGL(plane_copy, cgrid, dim3(256), 0, ctx->stream, d_in, in_is_u8, in_pitch,
                           in_frame_stride, d_out, nx, ny);
        src = d_out;
    }
    return launch_generic(ctx, src, d_tmp, d_out, nx, ny, n_frames, size, p.B, d_B);
}

// structure tensor + Harris response in one kernel: applies to the discrete Gaussian with a specialised radius on
// 16-byte aligned planes whose rows are whole quads
bool tensor_response_supported(const imgfd_ctx *ctx, int nx, int ny, float sigma, int gauss, int measure, const float *d_Ix, const float *d_Iy, const float *d_R)
{
    if (gauss != IMGFD_STD_GAUSSIAN || measure != IMGFD_HARRIS_MEASURE || !(sigma > 0)) return false;
    const int size = fir_size(sigma, 3);
    if (size > nx || !tensor_fast_path(size - 1)) return false;
    return nx % 4 == 0 && nx >= 4 && (size_t)d_Ix % 16 == 0 && (size_t)d_Iy % 16 == 0 && (size_t)d_R % 16 == 0;
}

// d_tq (optional, nx * ny * n_frames / 4 bytes): the kernel also publishes "R is not below Th", a byte per quad of pixels --
// the input of launch_harris_nms_sparse
imgfd_status launch_tensor_response(imgfd_ctx *ctx, const float *d_Ix, const float *d_Iy, float *d_R, int nx, int ny,
                                    int n_frames, float sigma, float k, unsigned char *d_tq, float Th)
{
    double B[IMGFD_MAX_TAPS];
    const int size = fir_coeffs(sigma, 3, B);
    if (size < 0) return imgfd_fail(ctx, IMGFD_ERR_UNSUPPORTED, "fused structure tensor + response: no such radius (callers ask tensor_response_supported first)");
    const imgfd_status st = launch_tensor_march(ctx, d_Ix, d_Iy, d_R, nullptr, nullptr, nx, ny, n_frames, size - 1, B, k, 2, d_tq, Th);
    if (st == IMGFD_ERR_UNSUPPORTED) return imgfd_fail(ctx, st, "fused structure tensor + response: unsupported shape");
    return st;
}

// compute_autocorrelation_matrix(): harris.cpp:44-70
imgfd_status launch_structure_tensor(imgfd_ctx *ctx, const float *d_Ix, const float *d_Iy, float *d_A,
                                     float *d_B, float *d_C, int nx, int ny, int n_frames, float sigma,
                                     int gauss, float *d_tmp)
{
    if (gauss == 2) gauss = 1;  // harris.cpp:64-65
    const size_t n = (size_t)nx * ny * n_frames;
    auto products = [&]() -> imgfd_status {
        hipLaunchKernelGGL(tensor_products, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                           d_Ix, d_Iy, d_A, d_B, d_C, n);
        IMGFD_HIP(ctx, hipGetLastError());
        return IMGFD_OK;
    };
    if (gauss == IMGFD_FAST_GAUSSIAN) {
        IMGFD_TRY(products());
        if (!(sigma > 0)) return IMGFD_OK;
        float *pl[3] = {d_A, d_B, d_C};
        for (int i = 0; i < 3; i++) IMGFD_TRY(launch_sii_gaussian(ctx, pl[i], pl[i], nx, ny, n_frames, sigma, d_tmp));  // in place, harris.cpp:67-69
        return IMGFD_OK;
    }
    if (sigma <= 0) return products();
    FirParams p;
    memset(&p, 0, sizeof p);
    const int size = fir_size(sigma, 3);
    if (size > nx) return products();
    const double *d_taps = nullptr;
    if (size > IMGFD_MAX_TAPS) IMGFD_TRY(fir_taps_device(ctx, sigma, size, &d_taps));
    else (void)fir_coeffs(sigma, 3, p.B);
    const int R = size - 1;
    p.in0 = d_Ix; p.in1 = d_Iy; p.out0 = d_A; p.out1 = d_B; p.out2 = d_C; p.nx = nx; p.ny = ny;
    p.in_pitch = nx; p.in_frame_stride = (long)nx * ny; p.out_frame_stride = (long)nx * ny;
    if (tensor_fast_path(R)) {  // radius 7, 3, 1: the marching structure-tensor kernel of fir_tensor.hip (any shape and alignment)
        const imgfd_status st = launch_tensor_march(ctx, d_Ix, d_Iy, d_A, d_B, d_C, nx, ny, n_frames, R, p.B, 0.f, 0);
        if (st != IMGFD_ERR_UNSUPPORTED) return st;
    }
    if (!d_tmp) return imgfd_fail(ctx, IMGFD_ERR_INVALID, "generic structure tensor needs a scratch plane");
    IMGFD_TRY(products());
    float *pl[3] = {d_A, d_B, d_C};
    for (int i = 0; i < 3; i++) IMGFD_TRY(launch_generic(ctx, pl[i], d_tmp, pl[i], nx, ny, n_frames, size, p.B, d_taps));
    return IMGFD_OK;
}
