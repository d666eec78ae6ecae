// image_amd/csrc/nms.hip -- Harris non-maximum suppression (K5).
//
// Replaces non_maximum_suppression(), image.CornerDetectionHarris/src/harris.cpp:141-255.  The
// reference walks each row with a Neubeck-style scan line and a shared skip[] array; what it decides
// is a pure window rule (checked against the compiled reference in the tests):
//   pixel (i,j), radius r, r <= i < ny-r, r <= j < nx-r, R(i,j) >= Th, is a corner iff every other
//   pixel q of its (2r+1)^2 window satisfies
//       R(q) <  R(i,j)   when q is in a row above, or to the right in the same row   (:188, :223-233)
//       R(q) <= R(i,j)   when q is in a row below, or to the left in the same row    (:201, :209-219)
// One thread per pixel, one wave per 64 consecutive pixels of a row, so that a single __ballot is the
// 64-bit word of the corner bit mask consumed by compact.hip; per-row corner counts are accumulated
// with one integer atomic per non-empty word.  A cheap 3x3 pre-test rejects almost every pixel above
// the threshold before the full window is read (reads hit L1/L2: R is streamed once from HBM).
#include "common.h"
#include "harris_device.h"

// The reference walks a row from column `radius` and first skips "the downhill at the beginning": every j with
// R[j] < Th or R[j-1] >= R[j] (harris.cpp:177).  A pixel inside that initial run is never a candidate, although the window
// rule would accept it when its left neighbour ties it exactly (left neighbours only have to be <=).  Callers test this
// for window maxima whose left neighbour is EQUAL (anything else already fails one of the two rules): true = the whole
// stretch radius..x belongs to the initial run, the reference emits nothing here.
__device__ __forceinline__ bool harris_row_start_blocks(const float *row, int x, int radius, float Th)
{
    for (int j = x; j >= radius; j--)
        if (!(row[j] < Th) && !(row[j - 1] >= row[j])) return false;  // the scan line would have stopped skipping at j
    return true;
}

__global__ void __launch_bounds__(256) harris_nms_kernel(const float *__restrict__ R, int nx, int ny, float Th,
                                                         int radius, unsigned long long *__restrict__ mask,
                                                         unsigned *__restrict__ rowcount, int words_per_row)
{
    const int lane = threadIdx.x & 63;
    const int x = blockIdx.x * 64 + lane;
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int frame = blockIdx.z;
    const float *Rf = R + (size_t)frame * nx * ny;
    bool corner = false;
    if (y < ny && x >= radius && x < nx - radius && y >= radius && y < ny - radius) {
        const float *c = Rf + (size_t)y * nx + x;
        const float v = *c;
        if (!(v < Th)) {  // skip[] = R < Th, harris.cpp:160-162
            // 3x3 pre-test with the window rule's own comparisons
            bool ok = !(c[-nx - 1] >= v) && !(c[-nx] >= v) && !(c[-nx + 1] >= v) && !(c[1] >= v) &&
                      !(c[-1] > v) && !(c[nx - 1] > v) && !(c[nx] > v) && !(c[nx + 1] > v);
            if (ok) {
                for (int dy = -radius; dy <= radius && ok; dy++) {
                    const float *row = c + (ptrdiff_t)dy * nx;
                    if (dy < 0) {
                        for (int dx = -radius; dx <= radius; dx++) ok = ok && !(row[dx] >= v);
                    } else if (dy > 0) {
                        for (int dx = -radius; dx <= radius; dx++) ok = ok && !(row[dx] > v);
                    } else {
                        for (int dx = -radius; dx < 0; dx++) ok = ok && !(row[dx] > v);
                        for (int dx = 1; dx <= radius; dx++) ok = ok && !(row[dx] >= v);
                    }
                }
            }
            // the scan line's start-of-row rule (harris.cpp:177): see harris_row_start_blocks
            if (ok && c[-1] == v) ok = !harris_row_start_blocks(Rf + (size_t)y * nx, x, radius, Th);
            corner = ok;
        }
    }
    const unsigned long long word = __ballot(corner);
    if (lane == 0 && y < ny && (int)blockIdx.x < words_per_row) {
        mask[((size_t)frame * ny + y) * words_per_row + blockIdx.x] = word;
        if (word) atomicAdd(&rowcount[(size_t)frame * ny + y], (unsigned)__popcll(word));
    }
}

imgfd_status launch_harris_nms(imgfd_ctx *ctx, const float *d_R, int nx, int ny, int n_frames, float Th,
                               int radius, const CompactBuffers &cb)
{
    // harris.cpp:151-152: nothing is detected on images not larger than the window; radius >= 1
    if (ny <= 2 * radius + 1 || nx <= 2 * radius + 1) radius = nx + ny;  // empties the search domain
    else if (radius < 1) radius = 1;
    dim3 grid(cb.words_per_row, ceil_div(ny, 4), n_frames);
    hipLaunchKernelGGL(harris_nms_kernel, grid, dim3(256), 0, ctx->stream, d_R, nx, ny, Th, radius, cb.mask,
                       cb.rowcount, cb.words_per_row);
    IMGFD_HIP(ctx, hipGetLastError());
    return IMGFD_OK;
}

// ------------------------------------------------------------------ K4 + K5 fused (batch path)
// The response plane is never materialised.  One workgroup owns a 64x64 tile: (1) R of exactly the tile is computed from
// A, B, C into LDS: 64 floats = 256 bytes per row and plane, i.e. whole cache lines only -- the kernel is bound by this
// fill, and a tile widened by a halo of even one pixel touches twice as many lines; (2) every pixel applies the threshold
// and the 3x3 part of the window rule, survivors (a few per tile) go to an LDS candidate list; (3) the waves take
// candidates in turn and test the full (2r+1)^2 window with all 64 lanes (up to three window positions per lane, __any as
// the verdict).  In (2) and (3) a neighbour outside the tile is computed from A, B, C on the spot: only pixels above the
// threshold ever look at neighbours, so this is rare; (4) keepers set their bit in the tile row's mask word (LDS),
// written out at the end.  HBM traffic: the 12 B/px of A, B, C.
#define RN_TX 64
#ifndef RN_TY
#define RN_TY 64
#endif
#define RN_HALO 6  // largest window radius served
#define RN_MAXC (RN_TX * RN_TY / 4)  // 3x3 local maxima cannot be denser than one per 2x2 block

// HC > 0: the window radius is the compile-time constant HC (index arithmetic by constants); HC == 0: any radius <= RN_HALO
// FROM_R: A is a materialised response plane (written by fir_tensor's response epilogue): the tile is copied, not computed
template <int MEASURE, int HC, bool FROM_R = false>
__global__ void __launch_bounds__(256) harris_resp_nms_kernel(const float *__restrict__ A, const float *__restrict__ B,
                                                              const float *__restrict__ C, int nx, int ny, float k, float Th,
                                                              int radius_rt, int vec4, unsigned long long *__restrict__ mask,
                                                              unsigned *__restrict__ rowcount, int words_per_row)
{
    const int radius = HC > 0 ? HC : radius_rt;
    // LDS tile of R: exactly the tile, columns x0 .. x0+63, rows y0 .. y0+TY-1
    constexpr int XO = 0, YO = 0, LW = RN_TX + 2 * XO, LH = RN_TY + 2 * YO, LP = LW + 4;
    __shared__ __attribute__((aligned(16))) float sR[LH][LP];
    __shared__ unsigned cand[RN_MAXC];            // (row << 8) | column, tile coordinates
    __shared__ unsigned long long rowmask[RN_TY];  // tile width = 64 = one mask word per tile row
    __shared__ unsigned ncand;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int x0 = blockIdx.x * RN_TX, y0 = blockIdx.y * RN_TY;
    const size_t fo = (size_t)blockIdx.z * nx * ny;
    const float *Af = A + fo, *Bf = B + fo, *Cf = C + fo;
    if (tid == 0) ncand = 0;
    for (int i = tid; i < RN_TY; i += 256) rowmask[i] = 0ull;
    const bool vec = vec4 && x0 - XO >= 0 && x0 - XO + LW <= nx;  // workgroup-uniform
    if (vec) {
        // all of a thread's loads are issued before the first use (straight-line, clamped instead of branched around):
        // the fill is latency-bound otherwise, one exposed round trip per 256 float4 triples
        typedef float v4f __attribute__((vector_size(16)));  // a native vector: arrays of HIP's float4 struct may land in scratch
        constexpr int NI = LH * (LW / 4), NR = (NI + 255) / 256;
        v4f a[NR], b[NR], c[NR];
#pragma unroll
        for (int u = 0; u < NR; u++) {
            const int i = min(tid + 256 * u, NI - 1);
            const int r = i / (LW / 4), q = i - r * (LW / 4);
            const int gy = min(max(y0 + r - YO, 0), ny - 1);
            const size_t p = (size_t)gy * nx + (x0 - XO + 4 * q);
            a[u] = *reinterpret_cast<const v4f *>(Af + p);
            if (!FROM_R) {
                b[u] = *reinterpret_cast<const v4f *>(Bf + p);
                c[u] = *reinterpret_cast<const v4f *>(Cf + p);
            }
        }
#pragma unroll
        for (int u = 0; u < NR; u++) {
            const int i = tid + 256 * u;
            if (i < NI) {
                const int r = i / (LW / 4), q = i - r * (LW / 4);
                const int gy = y0 + r - YO;
                v4f v = {0.f, 0.f, 0.f, 0.f};
                if (gy >= 0 && gy < ny) {
#pragma unroll
                    for (int e = 0; e < 4; e++) v[e] = FROM_R ? a[u][e] : harris_response_value<MEASURE>(a[u][e], b[u][e], c[u][e], k);
                }
                *reinterpret_cast<v4f *>(&sR[r][4 * q]) = v;
            }
        }
    } else {
        for (int i = tid; i < LH * LW; i += 256) {
            const int r = i / LW, c = i - r * LW;
            const int gx = x0 + c - XO, gy = y0 + r - YO;
            float v = 0.f;  // outside the image: never compared (the search domain stays `radius` away from the border)
            if (gx >= 0 && gx < nx && gy >= 0 && gy < ny) {
                const size_t p = (size_t)gy * nx + gx;
                v = FROM_R ? Af[p] : harris_response_value<MEASURE>(Af[p], Bf[p], Cf[p], k);
            }
            sR[r][c] = v;
        }
    }
    __syncthreads();
    // R at tile coordinates (ty, tx): from LDS inside the tile, from the planes beyond it (callers stay inside the image)
    auto R_at = [&](int ty, int tx) __attribute__((always_inline)) -> float {
        if (ty >= 0 && ty < RN_TY && tx >= 0 && tx < RN_TX) return sR[ty][tx];
        const size_t p = (size_t)(y0 + ty) * nx + (x0 + tx);
        return FROM_R ? Af[p] : harris_response_value<MEASURE>(Af[p], Bf[p], Cf[p], k);
    };
    // (2) threshold + 3x3 pre-test with the window rule's own comparisons
    for (int r = wv; r < RN_TY; r += 4) {
        const int x = x0 + lane, y = y0 + r;
        if (y < ny && x >= radius && x < nx - radius && y >= radius && y < ny - radius) {
            const float v = sR[r][lane];
            if (!(v < Th)) {  // skip[] = R < Th, harris.cpp:160-162
                const bool ok = !(R_at(r - 1, lane - 1) >= v) && !(R_at(r - 1, lane) >= v) && !(R_at(r - 1, lane + 1) >= v) &&
                                !(R_at(r, lane + 1) >= v) && !(R_at(r, lane - 1) > v) && !(R_at(r + 1, lane - 1) > v) &&
                                !(R_at(r + 1, lane) > v) && !(R_at(r + 1, lane + 1) > v);
                if (ok) {
                    const unsigned slot = atomicAdd(&ncand, 1u);
                    if (slot < RN_MAXC) cand[slot] = ((unsigned)r << 8) | (unsigned)lane;
                }
            }
        }
    }
    __syncthreads();
    // (3) full window, one candidate per wave at a time, window positions spread over the lanes
    const int n = min((int)ncand, RN_MAXC);
    const int side = 2 * radius + 1, npos = side * side;
    // every lane owns up to three window positions (lane, lane + 64, lane + 128 of the (2r+1)^2 <= 169): offsets and
    // the side of the tie rule are candidate-independent, so they are set up once
    int wdy[3], wdx[3];
    bool strict[3], live[3];
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const int pidx = lane + 64 * j;
        wdy[j] = pidx / side - radius;
        wdx[j] = pidx % side - radius;
        live[j] = pidx < npos && !(wdy[j] == 0 && wdx[j] == 0);
        strict[j] = wdy[j] < 0 || (wdy[j] == 0 && wdx[j] > 0);  // above, or to the right on the same row: must be <
    }
    for (int ci = wv; ci < n; ci += 4) {
        const int r = cand[ci] >> 8, c = cand[ci] & 255;
        const float v = sR[r][c];
        bool fail = false;
#pragma unroll
        for (int j = 0; j < 3; j++) {
            if (live[j]) {
                const float q = R_at(r + wdy[j], c + wdx[j]);
                fail = fail || (strict[j] ? (q >= v) : (q > v));
            }
        }
        if (!__any(fail) && lane == 0) {
            // start-of-row rule of the scan line (harris_row_start_blocks): only for an exact tie with the left neighbour
            bool blocked = false;
            if (R_at(r, c - 1) == v) {
                blocked = true;
                for (int j = x0 + c; j >= radius && blocked; j--) {
                    const float rj = R_at(r, j - x0), rl = R_at(r, j - 1 - x0);
                    if (!(rj < Th) && !(rl >= rj)) blocked = false;
                }
            }
            if (!blocked) atomicOr(&rowmask[r], 1ull << c);
        }
    }
    __syncthreads();
    // (4) mask words
    for (int r = tid; r < RN_TY; r += 256) {
        const int y = y0 + r;
        if (y >= ny) continue;
        const unsigned long long word = rowmask[r];
        mask[((size_t)blockIdx.z * ny + y) * words_per_row + blockIdx.x] = word;
        if (word) atomicAdd(&rowcount[(size_t)blockIdx.z * ny + y], (unsigned)__popcll(word));
    }
}

// false when the fused kernel cannot serve this radius (its LDS halo is RN_HALO): use response + NMS kernels instead
bool harris_resp_nms_supports(int nx, int ny, int radius)
{
    if (ny <= 2 * radius + 1 || nx <= 2 * radius + 1) return true;  // empty search domain: any radius works
    return (radius < 1 ? 1 : radius) <= RN_HALO;
}

imgfd_status launch_harris_resp_nms(imgfd_ctx *ctx, const float *d_A, const float *d_B, const float *d_C, int nx, int ny,
                                    int n_frames, int measure, float k, float Th, int radius, const CompactBuffers &cb)
{
    if (ny <= 2 * radius + 1 || nx <= 2 * radius + 1) {
        // harris.cpp:151-152: nothing is detected on images not larger than the window
        IMGFD_HIP(ctx, hipMemsetAsync(cb.mask, 0, sizeof(unsigned long long) * (size_t)cb.words_per_row * ny * n_frames, ctx->stream));
        return IMGFD_OK;
    }
    if (radius < 1) radius = 1;
    if (radius > RN_HALO) return imgfd_fail(ctx, IMGFD_ERR_UNSUPPORTED, "fused response+NMS: radius exceeds the LDS halo");
    dim3 grid(cb.words_per_row, ceil_div(ny, RN_TY), n_frames);
#define RN_LAUNCH(M, HC) hipLaunchKernelGGL((harris_resp_nms_kernel<M, HC>), grid, dim3(256), 0, ctx->stream, d_A, d_B, d_C, nx, ny, k, Th, radius, vec4, cb.mask, cb.rowcount, cb.words_per_row)
    // float4 tile loads: 16-byte aligned planes and whole quads per row (frames are nx*ny floats apart)
    const int vec4 = nx % 4 == 0 && (size_t)d_A % 16 == 0 && (size_t)d_B % 16 == 0 && (size_t)d_C % 16 == 0;
    if (measure == IMGFD_SHI_TOMASI_MEASURE) RN_LAUNCH(1, 0);
    else if (measure == IMGFD_HARMONIC_MEAN_MEASURE) RN_LAUNCH(2, 0);
    else if (radius == 5) RN_LAUNCH(0, 5);  // image_harris() defaults: sigma_i 2.5 -> radius 5
    else RN_LAUNCH(0, 0);
#undef RN_LAUNCH
    IMGFD_HIP(ctx, hipGetLastError());
    return IMGFD_OK;
}

// ---- NMS from threshold bits.  fir_tensor's response epilogue leaves, next to R, one bit per pixel "R is not below Th"
// (harris.cpp:160-162; a byte per quad of pixels, tq; the quad's own horizontal comparisons already applied): only those pixels -- corners and their
// surroundings, a fraction of a percent of a frame -- are ever candidates, and a neighbour below the threshold never beats
// one, so nothing of R is read but the 3x3 and (2r+1)^2 neighbourhoods of the candidates.  (The tiled kernel above streams
// the whole plane through LDS: 4 B/px and 25 VALU instructions per pixel to find them.)
//   a workgroup takes SN_WORDS consecutive mask words, a thread each (16 quad bytes -> 64 bits): bits outside the search
//   domain are dropped, the rest
//   become a dense list in LDS (popcount, block scan);
//   (2) the window rule's 3x3 pre-test, a thread per listed pixel; survivors move to the front of the list;
//   (3) the full window, a wave per survivor, window positions over the lanes; the scan line's start-of-row rule;
//   (4) the surviving bits go to `mask`, their popcounts to `rowcount`.
// The comparisons are those of harris_resp_nms_kernel, term by term.
#ifndef SN_WORDS
#define SN_WORDS 256
#endif
#ifndef SN_LIST
#define SN_LIST 4096  // list entries (8 KB)
#endif
template <int HC>
__global__ void __launch_bounds__(SN_WORDS) harris_nms_sparse(const float *__restrict__ Rp, const unsigned char *__restrict__ tq,
                                                            int nx, int ny, float Th, int radius_rt,
                                                            unsigned long long *__restrict__ mask, unsigned *__restrict__ rowcount,
                                                            int wpr, size_t nwords)
{
    // The list holds SN_LIST entries: a workgroup's words carry a few hundred candidate bits in practice (a fraction of a
    // percent of 16 384 pixels), and 32 KB for the worst case left four workgroups per CU where the kernel -- a chain of
    // four dependent memory round trips per workgroup -- wants as many resident as there are wave slots.  Words whose bits
    // do not fit are taken in several GROUPS of waves, one after the other (a wave's 64 words hold 4096 bits at the most).
    __shared__ unsigned short list[SN_LIST];  // (thread << 6) | bit
    __shared__ unsigned long long res[SN_WORDS];
    __shared__ unsigned wave_sum[SN_WORDS / 64], nsurv;
    static_assert(SN_LIST >= 4096, "one wave's words always fit");
    const int radius = HC > 0 ? HC : radius_rt;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const size_t w0 = (size_t)blockIdx.x * SN_WORDS, w = w0 + tid;
    // word w = (row, wx): row = frame * ny + y
    const size_t row = w / (size_t)wpr;
    const int wx = (int)(w - row * (size_t)wpr), y = (int)(row % (size_t)ny);
    unsigned long long word = 0ull;
    {   // the search domain keeps `radius` away from the border (harris.cpp:170-176)
        const int lo = radius - 64 * wx, hi = nx - radius - 64 * wx;  // valid bits: [lo, hi)
        if (w < nwords && !(y < radius || y >= ny - radius || hi <= 0 || lo >= 64)) {
            // the 16 quad bytes of the word (fewer at the end of a row), low nibbles packed into 64 bits
            const int qpr = nx >> 2, q0 = 16 * wx;
            const unsigned char *src = tq + row * (size_t)qpr + q0;
            unsigned d[4] = {0u, 0u, 0u, 0u};
            if (q0 + 16 <= qpr && ((size_t)src & 3) == 0) {
#pragma unroll
                for (int k = 0; k < 4; k++) d[k] = reinterpret_cast<const unsigned *>(src)[k];
            } else {
                for (int k = 0; k < 16 && q0 + k < qpr; k++) d[k >> 2] |= (unsigned)src[k] << (8 * (k & 3));
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                unsigned z = d[k] & 0x0f0f0f0fu;
                z = (z | (z >> 4)) & 0x00ff00ffu;
                z = (z | (z >> 8)) & 0x0000ffffu;
                word |= (unsigned long long)z << (16 * k);
            }
            if (lo > 0) word &= ~0ull << lo;
            if (hi < 64) word &= ~(~0ull << hi);
        }
    }
    res[tid] = 0ull;
    const unsigned cnt = (unsigned)__popcll(word);
    unsigned incl = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned t = __shfl_up(incl, d);
        if (lane >= d) incl += t;
    }
    if (lane == 63) wave_sum[wv] = incl;
    __syncthreads();
    unsigned all = 0;
#pragma unroll
    for (int k = 0; k < SN_WORDS / 64; k++) all += wave_sum[k];
    // pixel of a list entry: plane pointer of its frame, coordinates
    auto locate = [&](unsigned e, const float *&Rf, int &x, int &yy) __attribute__((always_inline)) {
        const size_t we = w0 + (e >> 6), rw = we / (size_t)wpr;
        const size_t frame = rw / (size_t)ny;
        yy = (int)(rw - frame * (size_t)ny);
        x = (int)(we - rw * (size_t)wpr) * 64 + (int)(e & 63u);
        Rf = Rp + frame * (size_t)nx * ny;
    };
    const int side = 2 * radius + 1, npos = side * side;
    // groups of waves whose bits fit the list: all four together (the case in practice), else one wave at a time
    const int gw = all <= SN_LIST ? SN_WORDS / 64 : 1;
    for (int g0 = 0; g0 < SN_WORDS / 64; g0 += gw) {  // workgroup-uniform
        const bool mine = wv >= g0 && wv < g0 + gw;
        unsigned pos = incl - cnt, total = 0;
#pragma unroll
        for (int k = 0; k < SN_WORDS / 64; k++) {
            const bool in = k >= g0 && k < g0 + gw;
            if (in && k < wv) pos += wave_sum[k];
            if (in) total += wave_sum[k];
        }
        if (tid == 0) nsurv = 0u;
        if (mine) {
            unsigned long long rest = word;
            while (rest) {
                const int bit = __ffsll((long long)rest) - 1;
                rest &= rest - 1;
                list[pos++] = (unsigned short)(tid << 6 | bit);
            }
        }
        __syncthreads();
        // ---- (2) threshold (already in the bits) + 3x3 pre-test with the window rule's own comparisons
        for (unsigned k0 = 0; k0 < total; k0 += SN_WORDS) {  // uniform trip count: barriers inside
            const unsigned k = k0 + tid;
            unsigned e = 0;
            bool ok = false;
            if (k < total) {
                e = list[k];
                const float *Rf; int x, yy;
                locate(e, Rf, x, yy);
                const float *c = Rf + (size_t)yy * nx + x;
                const float v = c[0];
                ok = !(c[-nx - 1] >= v) && !(c[-nx] >= v) && !(c[-nx + 1] >= v) && !(c[1] >= v) && !(c[-1] > v) &&
                     !(c[nx - 1] > v) && !(c[nx] > v) && !(c[nx + 1] > v);
            }
            __syncthreads();  // every entry of this trip has been read: survivors may overwrite the front of the list
            if (ok) list[atomicAdd(&nsurv, 1u)] = (unsigned short)e;
            __syncthreads();
        }
        // ---- (3) full window, one survivor per wave at a time, window positions spread over the lanes
        const int n = (int)nsurv;
        for (int ci = wv; ci < n; ci += SN_WORDS / 64) {
            const unsigned e = list[ci];
            const float *Rf; int x, yy;
            locate(e, Rf, x, yy);
            const float *c = Rf + (size_t)yy * nx + x;
            const float v = c[0];
            bool fail = false;
            for (int pidx = lane; pidx < npos; pidx += 64) {
                const int dy = pidx / side - radius, dx = pidx % side - radius;
                if (dy == 0 && dx == 0) continue;
                const float q = c[(long)dy * nx + dx];
                const bool strict = dy < 0 || (dy == 0 && dx > 0);  // above, or to the right on the same row: must be <
                fail = fail || (strict ? (q >= v) : (q > v));
            }
            if (!__any(fail) && lane == 0) {
                // start-of-row rule of the scan line (harris_row_start_blocks): only for an exact tie with the left neighbour
                bool blocked = false;
                if (c[-1] == v) {
                    blocked = true;
                    const float *rowp = Rf + (size_t)yy * nx;
                    for (int j = x; j >= radius && blocked; j--) {
                        const float rj = rowp[j], rl = rowp[j - 1];
                        if (!(rj < Th) && !(rl >= rj)) blocked = false;
                    }
                }
                if (!blocked) atomicOr(&res[e >> 6], 1ull << (e & 63u));
            }
        }
        __syncthreads();  // the list and nsurv are free for the next group
    }
    // ---- (4)
    if (w < nwords) {
        const unsigned long long out = res[tid];
        mask[w] = out;
        if (out) atomicAdd(&rowcount[row], (unsigned)__popcll(out));
    }
}

// threshold quads of a materialised R plane (what fir_tensor's response epilogue writes on the way): the stage doorway's
// way into harris_nms_sparse.  n_quads = nx / 4 * ny * n_frames
__global__ void __launch_bounds__(256) harris_threshold_quads(const float *__restrict__ R, unsigned char *__restrict__ tq, float Th, size_t n_quads)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_quads) return;
    const float *v = R + 4 * i;
    tq[i] = (unsigned char)harris_quad_bits(v[0], v[1], v[2], v[3], Th);
}
imgfd_status launch_harris_threshold_quads(imgfd_ctx *ctx, const float *d_R, unsigned char *d_tq, int nx, int ny, int n_frames, float Th)
{
    const size_t n_quads = (size_t)(nx / 4) * ny * n_frames;
    hipLaunchKernelGGL(harris_threshold_quads, dim3((unsigned)((n_quads + 255) / 256)), dim3(256), 0, ctx->stream, d_R, d_tq, Th, n_quads);
    IMGFD_HIP(ctx, hipGetLastError());
    return IMGFD_OK;
}

// NMS on the R plane and the threshold quads of fir_tensor's response epilogue (launch_tensor_response); nx % 4 == 0
imgfd_status launch_harris_nms_sparse(imgfd_ctx *ctx, const float *d_R, const unsigned char *d_tq, int nx, int ny, int n_frames,
                                      float Th, int radius, const CompactBuffers &cb)
{
    const size_t nwords = (size_t)cb.words_per_row * ny * n_frames;
    if (ny <= 2 * radius + 1 || nx <= 2 * radius + 1) {  // harris.cpp:151-152: nothing is detected on images not larger than the window
        IMGFD_HIP(ctx, hipMemsetAsync(cb.mask, 0, sizeof(unsigned long long) * nwords, ctx->stream));
        return IMGFD_OK;
    }
    if (radius < 1) radius = 1;
    const dim3 grid((unsigned)((nwords + SN_WORDS - 1) / SN_WORDS));
    if (radius == 5)
        hipLaunchKernelGGL(harris_nms_sparse<5>, grid, dim3(SN_WORDS), 0, ctx->stream, d_R, d_tq, nx, ny, Th, radius, cb.mask, cb.rowcount, cb.words_per_row, nwords);
    else
        hipLaunchKernelGGL(harris_nms_sparse<0>, grid, dim3(SN_WORDS), 0, ctx->stream, d_R, d_tq, nx, ny, Th, radius, cb.mask, cb.rowcount, cb.words_per_row, nwords);
    IMGFD_HIP(ctx, hipGetLastError());
    return IMGFD_OK;
}

imgfd_status launch_harris_nms_tiled(imgfd_ctx *ctx, const float *d_R, int nx, int ny, int n_frames, float Th, int radius,
                                     const CompactBuffers &cb)
{
    if (ny <= 2 * radius + 1 || nx <= 2 * radius + 1) {
        IMGFD_HIP(ctx, hipMemsetAsync(cb.mask, 0, sizeof(unsigned long long) * (size_t)cb.words_per_row * ny * n_frames, ctx->stream));
        return IMGFD_OK;
    }
    if (radius < 1) radius = 1;
    if (radius > RN_HALO) return launch_harris_nms(ctx, d_R, nx, ny, n_frames, Th, radius, cb);
    dim3 grid(cb.words_per_row, ceil_div(ny, RN_TY), n_frames);
    const int vec4 = nx % 4 == 0 && (size_t)d_R % 16 == 0;
    if (radius == 5)
        hipLaunchKernelGGL((harris_resp_nms_kernel<0, 5, true>), grid, dim3(256), 0, ctx->stream, d_R, d_R, d_R, nx, ny, 0.f, Th, radius, vec4,
                           cb.mask, cb.rowcount, cb.words_per_row);
    else
        hipLaunchKernelGGL((harris_resp_nms_kernel<0, 0, true>), grid, dim3(256), 0, ctx->stream, d_R, d_R, d_R, nx, ny, 0.f, Th, radius, vec4,
                           cb.mask, cb.rowcount, cb.words_per_row);
    IMGFD_HIP(ctx, hipGetLastError());
    return IMGFD_OK;
}
