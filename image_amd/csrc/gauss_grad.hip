// image_amd/csrc/gauss_grad.hip -- K1 + K2 fused for the path every image_harris() call takes: the discrete Gaussian
// of sigma_d (radius 3 for sigma_d = 1) and the gradient of the smoothed image in one kernel.
//
// Replaces gaussian(I, I, sigma_d) + gradient(I, Ix, Iy), image.CornerDetectionHarris/src/harris.cpp:511-514, i.e.
// discrete_gaussian (gaussian.cpp:289-395) followed by central_differences / sobel_operator (gradient.cpp:17-106).
// The smoothed image never reaches HBM: one workgroup owns a 64x24 tile of Ix, Iy; it stages the (64+3+2R)x(32+3+2R)
// input pixels in LDS (reflected like the reference's borders), runs the row pass and the column pass with the
// reference's arithmetic (f64 accumulate, one rounding to float per pass, fir_window8) on the tile plus a one-pixel
// ring, and differentiates from LDS.  The gradient's border rule -- a border pixel takes the gradient of the nearest
// interior pixel, gradient.cpp:40-55 -- is applied by clamping the evaluation point to [1, n-2].
// HBM traffic: 1 B (u8) or 4 B (f32) read + 8 B written per pixel, instead of (1|4)+4 and 4+8 for the two kernels.
#include "common.h"
#include "fir_device.h"

#include <algorithm>
#include <type_traits>

#define GG_TX 64
#ifndef GG_TY
#define GG_TY 24  // MI355X, 32 x 4K frames, u8: 8 -> 1004 us, 16 -> 757, 24 -> 696, 32 -> 755 (LDS 19 KB: 8 workgroups per CU)
#endif
#ifndef GG_PX
#define GG_PX 4  // outputs per thread and pass (register window of GG_PX + 2R values)
#endif

struct GaussGradParams {
    const void *in;
    float *Ix, *Iy;
    int nx, ny, in_pitch;
    int vec4;  // input base, pitch and frame stride allow aligned 4-pixel loads
    int vec16; // ... aligned 16-pixel loads (u8 frames)
    long in_frame_stride;
    double B[8];
    TileRuns runs;
};

template <int R, int GRAD, bool U8, bool FMA>
__global__ void __launch_bounds__(256) gauss_grad_tile(GaussGradParams p)
{
    // smoothed tile: the 64 x GG_TY outputs plus a ring of 2 pixels left/top and 1 right/bottom (a tile that starts on the
    // last image column/row evaluates its clamped gradient one pixel further inside)
    constexpr int SW = GG_TX + 3, SH = GG_TY + 3;
    // raw tile: u8 frames columns x0-16 .. x0+79 (first column 16-byte aligned: interior tiles load 16 pixels at a time),
    // f32 frames x0-8 .. x0+71 (float4s); rows y0-2-R .. y0+TY+R
    constexpr int XO = U8 ? 16 : 8, RW = GG_TX + 2 * XO, RH = SH + 2 * R, OFFX = XO - 2 - R;
    static_assert(OFFX >= 0 && SW + 2 * R + OFFX <= RW, "raw tile too narrow for this radius");
    static_assert(GG_PX % 4 == 0 && ((SW + GG_PX - 1) / GG_PX * GG_PX + OFFX + 2 * R + 3) / 4 * 4 <= RW + 4, "dword window reads stay inside the row");
    static_assert(OFFX % 4 == 3 || !U8, "byte window of the u8 row pass starts on byte 3 of a dword");
    // raw tile: bytes for u8 frames (a quarter of the LDS: the kernel is occupancy-bound, not bandwidth-bound), floats else
    constexpr int RP = U8 ? RW + 16 : RW + 4;  // row pitch in elements (u8: rows stay 16-byte aligned for ds_write_b128)
    using raw_t = typename std::conditional<U8, unsigned char, float>::type;
    __shared__ __attribute__((aligned(16))) raw_t raw[RH][RP];
    constexpr int SWP = (SW + GG_PX - 1) / GG_PX * GG_PX, SHP = (SH + GG_PX - 1) / GG_PX * GG_PX;  // whole groups of GG_PX
    __shared__ float rowf[SHP + 2 * R][SWP + 1];  // rows RH .. are padding: read by the last row group, never written
    __shared__ float is[SHP][SWP + 1];
    const int tid = threadIdx.x;
    // TileRuns, common.h.  No barrier is needed between two tiles: each of the three LDS arrays is next written one
    // barrier after its last readers
    {
    const unsigned run_id = blockIdx.x;  // one run of tiles per workgroup
    const int frame = (int)(run_id / (unsigned)(p.runs.runs_per_band * p.runs.bands));
    const int in_frame = (int)(run_id - (unsigned)frame * (unsigned)(p.runs.runs_per_band * p.runs.bands));
    const int band = in_frame / p.runs.runs_per_band, tile0 = (in_frame - band * p.runs.runs_per_band) * p.runs.run;
    for (int tile_x = tile0; tile_x < min(tile0 + p.runs.run, p.runs.tiles_x); tile_x++) {
    const int x0 = tile_x * GG_TX, y0 = band * GG_TY;
    const size_t fin = (size_t)frame * p.in_frame_stride;
    // ---- input tile; logical index outside the image -> the reference's reflection (gaussian.cpp:345-349, 376-380)
    if (U8 && p.vec16 && x0 - XO >= 0 && x0 - XO + RW <= p.nx) {  // workgroup-uniform: no column reflection in this tile
        for (int i = tid; i < RH * (RW / 16); i += 256) {
            const int r = i / (RW / 16), q = i - r * (RW / 16);
            const int gy = fir_reflect(y0 - 2 - R + r, p.ny);
            const size_t off = fin + (size_t)gy * p.in_pitch + (x0 - XO + 16 * q);
            *reinterpret_cast<uint4 *>(&raw[r][16 * q]) = *reinterpret_cast<const uint4 *>(reinterpret_cast<const unsigned char *>(p.in) + off);
        }
    } else if (p.vec4 && x0 - XO >= 0 && x0 - XO + RW <= p.nx) {
        for (int i = tid; i < RH * (RW / 4); i += 256) {
            const int r = i / (RW / 4), q = i - r * (RW / 4);
            const int gy = fir_reflect(y0 - 2 - R + r, p.ny);
            const size_t off = fin + (size_t)gy * p.in_pitch + (x0 - XO + 4 * q);
            if (U8)
                *reinterpret_cast<unsigned *>(&raw[r][4 * q]) =
                    *reinterpret_cast<const unsigned *>(reinterpret_cast<const unsigned char *>(p.in) + off);
            else
                *reinterpret_cast<float4 *>(&raw[r][4 * q]) = *reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(p.in) + off);
        }
    } else {
        for (int i = tid; i < RH * RW; i += 256) {
            const int r = i / RW, c = i - r * RW;
            const int gy = fir_reflect(y0 - 2 - R + r, p.ny), gx = fir_reflect(x0 - XO + c, p.nx);
            const size_t off = fin + (size_t)gy * p.in_pitch + gx;
            if (U8) raw[r][c] = (raw_t) reinterpret_cast<const unsigned char *>(p.in)[off];
            else raw[r][c] = (raw_t) reinterpret_cast<const float *>(p.in)[off];
        }
    }
    __syncthreads();
    // ---- row pass: rowf[r][c] for r < RH, c < SW; a thread takes GG_PX consecutive columns
    constexpr int RG = (SW + GG_PX - 1) / GG_PX;
    for (int i = tid; i < RH * RG; i += 256) {
        const int r = i / RG, c0 = (i - r * RG) * GG_PX;
        double d[GG_PX + 2 * R];
        if (U8) {
            // the window starts at byte c0 + OFFX of the row (c0 % 4 == 0): whole dwords, bytes picked at compile time
            constexpr int NW = (OFFX % 4 + GG_PX + 2 * R + 3) / 4;
            const unsigned *row32 = reinterpret_cast<const unsigned *>(&raw[r][0]) + ((c0 + OFFX) >> 2);
            unsigned wv[NW];
#pragma unroll
            for (int k = 0; k < NW; k++) wv[k] = row32[k];
#pragma unroll
            for (int k = 0; k < GG_PX + 2 * R; k++) {
                constexpr int sh = OFFX % 4;
                d[k] = (double)((wv[(sh + k) >> 2] >> (8 * ((sh + k) & 3))) & 0xffu);  // one v_cvt_f64_u32 (bytes are exact either way)
            }
        } else {
#pragma unroll
            for (int k = 0; k < GG_PX + 2 * R; k++) d[k] = (double)raw[r][min(c0 + OFFX + k, RW - 1)];
        }
        float o[GG_PX];
        fir_window8<R, FMA, GG_PX>(d, p.B, o);
        // the row is padded to whole groups (SWP): the last group writes its surplus outputs into the pad unconditionally
        // (a test per output costs an exec-mask branch each)
#pragma unroll
        for (int k = 0; k < GG_PX; k++) rowf[r][c0 + k] = o[k];
    }
    __syncthreads();
    // ---- column pass: is[r][c] for r < SH, c < SW; a thread takes GG_PX consecutive rows of one column
    constexpr int CG = (SH + GG_PX - 1) / GG_PX;
    for (int i = tid; i < CG * SW; i += 256) {
        const int g = i / SW, c = i - g * SW, r0 = g * GG_PX;
        double d[GG_PX + 2 * R];
#pragma unroll
        for (int k = 0; k < GG_PX + 2 * R; k++) d[k] = (double)rowf[r0 + k][c];
        float o[GG_PX];
        fir_window8<R, FMA, GG_PX>(d, p.B, o);
#pragma unroll
        for (int k = 0; k < GG_PX; k++) is[r0 + k][c] = o[k];  // rows SH .. SHP-1 are padding
    }
    __syncthreads();
    // ---- gradient.  Interior tiles (no pixel on the image border, whole quads, aligned planes): a thread takes 8 consecutive
    // pixels of a row -- 26 LDS words, 32 float operations, four 16-byte streaming stores; one pixel per thread spent more
    // instructions on addresses and the loop than on the two differences (22 of the kernel's 66 per pixel)
    float *Ix = p.Ix + (size_t)frame * p.nx * p.ny, *Iy = p.Iy + (size_t)frame * p.nx * p.ny;
    if (GRAD != IMGFD_SOBEL_OPERATOR && p.vec4 && x0 >= 1 && x0 + GG_TX <= p.nx - 1 && y0 >= 1 && y0 + GG_TY <= p.ny - 1 &&
        (reinterpret_cast<size_t>(Ix) & 15) == 0 && (reinterpret_cast<size_t>(Iy) & 15) == 0) {
        for (int item = tid; item < GG_TY * (GG_TX / 8); item += 256) {
            const int r = item / (GG_TX / 8), c = (item - r * (GG_TX / 8)) * 8;
            const int i = r + 2, j = c + 2;  // tile coordinates of the first pixel
            float mid[10], up[8], dn[8];
#pragma unroll
            for (int k = 0; k < 10; k++) mid[k] = is[i][j - 1 + k];
#pragma unroll
            for (int k = 0; k < 8; k++) { up[k] = is[i - 1][j + k]; dn[k] = is[i + 1][j + k]; }
            float gx[8], gy[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {  // gradient.cpp:34-35, as below
                gx[k] = 0.5f * (mid[k + 2] - mid[k]);
                gy[k] = 0.5f * (dn[k] - up[k]);
            }
            const size_t o = (size_t)(y0 + r) * p.nx + (x0 + c);
            typedef float v4f __attribute__((vector_size(16)));
            const v4f x0v = {gx[0], gx[1], gx[2], gx[3]}, x1v = {gx[4], gx[5], gx[6], gx[7]};
            const v4f y0v = {gy[0], gy[1], gy[2], gy[3]}, y1v = {gy[4], gy[5], gy[6], gy[7]};
            IMGFD_STREAM_STORE(x0v, reinterpret_cast<v4f *>(Ix + o));
            IMGFD_STREAM_STORE(x1v, reinterpret_cast<v4f *>(Ix + o + 4));
            IMGFD_STREAM_STORE(y0v, reinterpret_cast<v4f *>(Iy + o));
            IMGFD_STREAM_STORE(y1v, reinterpret_cast<v4f *>(Iy + o + 4));
        }
        continue;
    }
    // border tiles, Sobel, unaligned planes: lane = column (coalesced 256-byte row segments)
    const int lane = tid & 63;
    for (int r = tid >> 6; r < GG_TY; r += 4) {
        const int x = x0 + lane, y = y0 + r;
        if (x >= p.nx || y >= p.ny) continue;
        // border pixels take the gradient of the nearest interior pixel (gradient.cpp:40-55)
        const int j = min(max(x, 1), p.nx - 2) - x0 + 2, i = min(max(y, 1), p.ny - 2) - y0 + 2;  // tile coordinates
        float gx, gy;
        if (GRAD == IMGFD_SOBEL_OPERATOR) {  // gradient.cpp:80-87: float sums, double constants, double adds, float store
            gx = (float)(1. / 4. * (is[i][j + 1] - is[i][j - 1]) +
                         1. / 8. * (is[i - 1][j + 1] + is[i + 1][j + 1] - is[i - 1][j - 1] - is[i + 1][j - 1]));
            gy = (float)(1. / 4. * (is[i + 1][j] - is[i - 1][j]) +
                         1. / 8. * (is[i + 1][j + 1] + is[i + 1][j - 1] - is[i - 1][j + 1] - is[i - 1][j - 1]));
        } else {                             // gradient.cpp:34-35: (float)(0.5 * (a - b)) with a - b in float.  Halving a float in
            // double and rounding back is halving it in float (one exact product, one rounding): no conversions needed
            gx = 0.5f * (is[i][j + 1] - is[i][j - 1]);
            gy = 0.5f * (is[i + 1][j] - is[i - 1][j]);
        }
        // streaming stores: the 8 B/px written here would otherwise push the u8 lines a run of tiles shares out of the L2
        // before the next tile of the run reads them again
        IMGFD_STREAM_STORE(gx, &Ix[(size_t)y * p.nx + x]);
        IMGFD_STREAM_STORE(gy, &Iy[(size_t)y * p.nx + x]);
    }
    }
    }
}

bool gauss_grad_fused_supported(int nx, int ny, float sigma, int gauss_type)
{
    if (gauss_type != IMGFD_STD_GAUSSIAN || !(sigma > 0) || nx < 3 || ny < 3) return false;
    const int size = (int)(3 * sigma) + 1;  // gaussian.cpp:310 with the default precision 3
    return size == 4 && size <= nx;         // radius 3 (sigma_d in [1, 4/3)); gaussian.cpp:312: size > xdim leaves the image untouched
}

imgfd_status launch_gauss_grad_fused(imgfd_ctx *ctx, const void *d_in, int in_is_u8, int in_pitch, size_t in_frame_stride,
                                     float *d_Ix, float *d_Iy, int nx, int ny, int n_frames, float sigma, int grad_type,
                                     unsigned *d_rowcount, bool *rowcount_cleared)
{
    if (rowcount_cleared) *rowcount_cleared = false;
    GaussGradParams p;
    memset(&p, 0, sizeof p);
    if (fir_coeffs(sigma, 3, p.B) != 4) return imgfd_fail(ctx, IMGFD_ERR_UNSUPPORTED, "fused gaussian+gradient: radius is not 3");
    p.in = d_in; p.Ix = d_Ix; p.Iy = d_Iy; p.nx = nx; p.ny = ny; p.in_pitch = in_pitch; p.in_frame_stride = (long)in_frame_stride;
    if (ctx->tune.gauss_march && gauss_grad_march_supported(d_in, in_is_u8, in_pitch, in_frame_stride, d_Ix, d_Iy, nx, ny)) {
        if (rowcount_cleared) *rowcount_cleared = d_rowcount != nullptr;
        return launch_gauss_grad_march(ctx, d_in, in_pitch, in_frame_stride, d_Ix, d_Iy, nx, ny, n_frames, p.B, grad_type, d_rowcount);
    }
    const size_t esz = in_is_u8 ? 1 : 4;
    p.vec4 = ((size_t)d_in % (4 * esz) == 0) && in_pitch % 4 == 0 && in_frame_stride % 4 == 0 && nx % 4 == 0;
    p.vec16 = in_is_u8 && ((size_t)d_in % 16 == 0) && in_pitch % 16 == 0 && in_frame_stride % 16 == 0 && nx % 16 == 0;
    const bool sobel = grad_type == IMGFD_SOBEL_OPERATOR;
    const int tiles_x = ceil_div(nx, GG_TX), bands = ceil_div(ny, GG_TY);
    p.runs = tile_runs(tiles_x, bands, n_frames, tile_run_length(ctx, tiles_x, bands, n_frames));
#define GG_LAUNCH(G, U, F) hipLaunchKernelGGL((gauss_grad_tile<3, G, U, F>), dim3(p.runs.total), dim3(256), 0, ctx->stream, p)
    if (ctx->fir_mode) {
        if (in_is_u8) { if (sobel) GG_LAUNCH(1, true, true); else GG_LAUNCH(0, true, true); }
        else { if (sobel) GG_LAUNCH(1, false, true); else GG_LAUNCH(0, false, true); }
    } else {
        if (in_is_u8) { if (sobel) GG_LAUNCH(1, true, false); else GG_LAUNCH(0, true, false); }
        else { if (sobel) GG_LAUNCH(1, false, false); else GG_LAUNCH(0, false, false); }
    }
#undef GG_LAUNCH
    IMGFD_HIP(ctx, hipGetLastError());
    return IMGFD_OK;
}
