// image_amd/csrc/fhog.hip -- Felzenszwalb HOG (dlib fHOG) behind imgfd_fhog / imgfd_fhog_dev (K13-K15).
//
// Replaces dlib_fhog(), image.dlib/src/rcpp_fhog.cpp:10-46, i.e. dlib's extract_fhog_features
// (image.dlib/inst/dlib-19.20/dlib/image_transforms/fhog.h:1104-1113 -> impl_extract_fhog_features :702-1046)
// for interlaced 8-bit RGB input; cell_size 1 takes dlib's special case (:499-694, fhog_features_cs1).
//
// The reference adds every pixel's gradient magnitude into 4 histogram cells with `+=` while it walks the
// image in raster order (:821-956), so each float bin is a sum in a fixed order.  The device keeps that
// order without atomics by turning the scatter into a gather:
//   K13 fhog_grad_orient : one thread per pixel -> packed (squared gradient length : 27 bits | orientation : 5
//                          bits) of the colour channel with the strongest gradient (:24-59 / :147-275)
//   K14 fhog_cell_hist   : one thread per histogram cell walks the <= (2*cell+2)^2 pixels that vote into it,
//                          in raster order, with the reference's own bilinear weights; then the cell energy
//                          (:959-968).  Deterministic, and bit-identical to the sequential sums.
//   K15 fhog_features    : one thread per output cell: 4 block norms, 18 + 9 + 4 features (:970-1045) in the
//                          lane order of dlib's simd4f code (sum = (l0+l2)+(l1+l3), the SSE2 build CRAN ships).
// The reference handles columns in groups of 8 with float "SIMD" arithmetic and the remainder with a scalar
// tail whose rounding order and colour tie-break differ (:828-918 vs :920-955); both are reproduced, keyed on
// the pixel's column (fhog_is_body).  Library build flag -ffp-contract=off keeps a*b+c unfused like x86-64 -O2.
#include "fhog_device.h"

#include <math.h>
#include <stdlib.h>

#include <algorithm>


static bool fhog_geometry(int rows, int cols, int cs, int pad_r, int pad_c, FhogGeom *g)
{
    memset(g, 0, sizeof *g);
    g->rows = rows; g->cols = cols; g->cs = cs;
    if (cs == 1) {  // impl_extract_fhog_features_cell_size_1, fhog.h:536-560: every pixel is a cell
        if (rows <= 2 || cols <= 2) return false;
        g->cells_nr = rows; g->cells_nc = cols;
        g->hog_nr = rows - 2; g->hog_nc = cols - 2;
        g->out_nr = g->hog_nr + pad_r - 1; g->out_nc = g->hog_nc + pad_c - 1;
        g->off_r = (pad_r - 1) / 2; g->off_c = (pad_c - 1) / 2;
        g->visible_nr = rows - 1; g->visible_nc = cols - 1;
        int x1 = 1;
        while (x1 < g->visible_nc - 7) x1 += 8;
        g->body_end = x1;
        return true;
    }
    g->cells_nr = (int)((float)rows / (float)cs + 0.5);
    g->cells_nc = (int)((float)cols / (float)cs + 0.5);
    if (g->cells_nr == 0 || g->cells_nc == 0) return false;
    g->hog_nr = std::max(g->cells_nr - 2, 0);
    g->hog_nc = std::max(g->cells_nc - 2, 0);
    if (g->hog_nr == 0 || g->hog_nc == 0) return false;
    g->out_nr = g->hog_nr + pad_r - 1;
    g->out_nc = g->hog_nc + pad_c - 1;
    g->off_r = (pad_r - 1) / 2;
    g->off_c = (pad_c - 1) / 2;
    g->visible_nr = (int)std::min((long)g->cells_nr * cs, (long)rows) - 1;
    g->visible_nc = (int)std::min((long)g->cells_nc * cs, (long)cols) - 1;
    int x = 1;
    while (x < g->visible_nc - 7) x += 8;  // for (x = 1; x < visible_nc - 7; x += 8), :828
    g->body_end = x;
    return true;
}


// K13: packed[y*cols + x] = (len << 5) | best_o for 1 <= y < visible_nr, 1 <= x < visible_nc; 0 elsewhere
// (a zero-length gradient votes +0.0f, which leaves every sum unchanged)
__global__ void __launch_bounds__(256) fhog_grad_orient(const unsigned char *__restrict__ rgb, size_t frame_stride,
                                                        unsigned *__restrict__ packed, FhogGeom g)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= g.cols) return;
    unsigned out = 0;
    if (y >= 1 && y < g.visible_nr && x >= 1 && x < g.visible_nc) {
        const unsigned char *img = rgb + (size_t)blockIdx.z * frame_stride;
        const unsigned char *pc = img + 3 * ((size_t)y * g.cols + x);
        const int rs = 3 * g.cols;
        int gx[3], gy[3], len[3];
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            gx[ch] = (int)pc[3 + ch] - (int)pc[-3 + ch];
            gy[ch] = (int)pc[rs + ch] - (int)pc[-rs + ch];
            len[ch] = gx[ch] * gx[ch] + gy[ch] * gy[ch];
        }
        int tx, ty, tl;
        if (x < g.body_end) {  // simd8 get_gradient, :265-274: strict > keeps the LATER channel on ties
            if (len[0] > len[1]) { tx = gx[0]; ty = gy[0]; tl = len[0]; } else { tx = gx[1]; ty = gy[1]; tl = len[1]; }
            if (!(tl > len[2])) { tx = gx[2]; ty = gy[2]; tl = len[2]; }
        } else {               // scalar get_gradient, :24-59: strict > keeps the EARLIER channel on ties
            tx = gx[0]; ty = gy[0]; tl = len[0];
            if (len[1] > tl) { tl = len[1]; tx = gx[1]; ty = gy[1]; }
            if (len[2] > tl) { tl = len[2]; tx = gx[2]; ty = gy[2]; }
        }
        out = ((unsigned)tl << 5) | (unsigned)fhog_best_orientation(tx, ty);  // :846-859 and :929-943 decide identically
    }
    packed[((size_t)blockIdx.z * g.rows + y) * g.cols + x] = out;
}

// packed (len << 5 | orientation) of one pixel from its four neighbours' RGB (little-endian dword soup: byte k of the
// 12-byte group of 4 pixels is channel k%3 of pixel k/3)
__device__ __forceinline__ unsigned fhog_pack(const int (&l)[3], const int (&r)[3], const int (&u)[3], const int (&d)[3], bool body)
{
    int gx[3], gy[3], len[3];
#pragma unroll
    for (int ch = 0; ch < 3; ch++) {
        gx[ch] = r[ch] - l[ch];
        gy[ch] = d[ch] - u[ch];
        len[ch] = gx[ch] * gx[ch] + gy[ch] * gy[ch];
    }
    int tx, ty, tl;
    if (body) {
        if (len[0] > len[1]) { tx = gx[0]; ty = gy[0]; tl = len[0]; } else { tx = gx[1]; ty = gy[1]; tl = len[1]; }
        if (!(tl > len[2])) { tx = gx[2]; ty = gy[2]; tl = len[2]; }
    } else {
        tx = gx[0]; ty = gy[0]; tl = len[0];
        if (len[1] > tl) { tl = len[1]; tx = gx[1]; ty = gy[1]; }
        if (len[2] > tl) { tl = len[2]; tx = gx[2]; ty = gy[2]; }
    }
    return ((unsigned)tl << 5) | (unsigned)fhog_best_orientation(tx, ty);
}

// K13, fast form: one thread per 4 consecutive pixels (12 bytes = 3 aligned dwords per row when cols % 4 == 0 and the
// frame is 4-byte aligned): 11 dword loads and one 16-byte store per 4 pixels instead of 12 byte loads per pixel.
__global__ void __launch_bounds__(256) fhog_grad_orient4(const unsigned char *__restrict__ rgb, size_t frame_stride,
                                                         unsigned *__restrict__ packed, FhogGeom g)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;  // group of 4 pixels
    const int y = blockIdx.y;
    const int x = 4 * q;
    if (x >= g.cols) return;
    unsigned out[4] = {0u, 0u, 0u, 0u};
    if (y >= 1 && y < g.visible_nr) {
        const unsigned *img = reinterpret_cast<const unsigned *>(rgb + (size_t)blockIdx.z * frame_stride);
        const int rd = 3 * g.cols / 4;  // dwords per row
        const unsigned *c = img + (size_t)y * rd + 3 * q;
        unsigned cen[5], up[3], dn[3];
        cen[0] = q > 0 ? c[-1] : 0u;
        cen[1] = c[0]; cen[2] = c[1]; cen[3] = c[2];
        cen[4] = x + 4 < g.cols ? c[3] : 0u;
#pragma unroll
        for (int k = 0; k < 3; k++) { up[k] = c[k - rd]; dn[k] = c[k + rd]; }
        // byte b (0..19) of the 5-dword centre window; pixel p of the group starts at byte 4 + 3p
        auto cb = [&](int b) -> int { return (int)((cen[b >> 2] >> (8 * (b & 3))) & 0xffu); };
        auto ub = [&](int b) -> int { return (int)((up[b >> 2] >> (8 * (b & 3))) & 0xffu); };
        auto db = [&](int b) -> int { return (int)((dn[b >> 2] >> (8 * (b & 3))) & 0xffu); };
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const int xx = x + p;
            if (xx >= 1 && xx < g.visible_nc) {
                int l[3], r[3], u[3], d[3];
#pragma unroll
                for (int ch = 0; ch < 3; ch++) {
                    l[ch] = cb(4 + 3 * (p - 1) + ch);
                    r[ch] = cb(4 + 3 * (p + 1) + ch);
                    u[ch] = ub(3 * p + ch);
                    d[ch] = db(3 * p + ch);
                }
                out[p] = fhog_pack(l, r, u, d, xx < g.body_end);
            }
        }
    }
    *reinterpret_cast<uint4 *>(packed + ((size_t)blockIdx.z * g.rows + y) * g.cols + x) = make_uint4(out[0], out[1], out[2], out[3]);
}

// K14: hist[(hr*HC + hc)*18 + o] for 1 <= hr <= cells_nr, 1 <= hc <= cells_nc (HC = cells_nc+2), and the
// cell energy norm[(hr-1)*cells_nc + hc-1].  The column part of every vote (which of the two bilinear weights this
// cell receives from column x, :838-841 / :946-949) does not depend on the row: it is tabulated once per thread.
#define FHOG_MAXW 24  // candidate columns per cell: 2*cs + 3 <= 19 for cs <= 8, plus up to 3 for the 16-byte alignment of the row fetch
__global__ void __launch_bounds__(64) fhog_cell_hist(const unsigned *__restrict__ packed, float *__restrict__ hist,
                                                     float *__restrict__ norm, FhogGeom g)
{
    __shared__ float bins[64][19];
    const int hc = blockIdx.x * 8 + (threadIdx.x & 7) + 1;
    const int hr = blockIdx.y * 8 + (threadIdx.x >> 3) + 1;
    if (hr > g.cells_nr || hc > g.cells_nc) return;
    float *b = bins[threadIdx.x];
#pragma unroll
    for (int o = 0; o < 18; o++) b[o] = 0.f;
    const unsigned *pk = packed + (size_t)blockIdx.z * g.rows * g.cols;
    const int cs = g.cs;
    // candidate rows / columns: every pixel whose bilinear footprint can touch this cell, widened by one
    const int y_lo = max(1, cs * (hr - 2) + cs / 2 - 1), y_hi = min(g.visible_nr - 1, cs * hr + cs / 2 + 1);
    const int x_lo = max(1, cs * (hc - 2) + cs / 2 - 1), x_hi = min(g.visible_nc - 1, cs * hc + cs / 2 + 1);
    // the row fetch starts on a 16-byte boundary when rows are 16-byte multiples: 6 dwordx4 loads instead of 20 dwords
    const bool vec = (g.cols & 3) == 0;
    const int xc = vec ? (x_lo & ~3) : x_lo;
    if (x_hi - xc + 1 > FHOG_MAXW) {
        // large cells: the column table does not fit; evaluate the column weights per vote (same order, same values)
        for (int y = y_lo; y <= y_hi; y++) {
            const float yp = ((float)y + 0.5) / (float)cs - 0.5;
            const int iyp = (int)floor(yp);
            const float vy0 = yp - iyp;
            const float vy1 = 1.0 - vy0;
            float wy;
            if (iyp + 1 == hr) wy = vy1;
            else if (iyp + 2 == hr) wy = vy0;
            else continue;
            const unsigned *row = pk + (size_t)y * g.cols;
            for (int x = x_lo; x <= x_hi; x++) {
                const unsigned p = row[x];
                const float v = sqrtf((float)(p >> 5));
                float w;
                if (x < g.body_end) {
                    const float xp = ((float)x + 0.5f) / (float)cs + 0.5f;
                    const int ixp = (int)xp;
                    const float vx0 = xp - (float)ixp;
                    const float vx1 = 1.0f - vx0;
                    if (ixp == hc) w = wy * (vx1 * v);
                    else if (ixp + 1 == hc) w = wy * (vx0 * v);
                    else continue;
                } else {
                    const float xp = ((double)x + 0.5) / (double)cs - 0.5;
                    const int ixp = (int)floor(xp);
                    const float vx0 = xp - ixp;
                    const float vx1 = 1.0 - vx0;
                    if (ixp + 1 == hc) w = wy * vx1 * v;
                    else if (ixp + 2 == hc) w = wy * vx0 * v;
                    else continue;
                }
                b[p & 31] += w;
            }
        }
    } else {
        // per column: vx (-1 = this column does not vote into the cell) and whether it takes the 8-wide path
        float vx[FHOG_MAXW];
        bool body[FHOG_MAXW];
#pragma unroll
        for (int k = 0; k < FHOG_MAXW; k++) {
            const int x = xc + k;
            vx[k] = -1.f;
            body[k] = x < g.body_end;
            if (x >= x_lo && x <= x_hi) {
                if (body[k]) {  // :838-841: hist column ixp / ixp+1
                    const float xp = ((float)x + 0.5f) / (float)cs + 0.5f;
                    const int ixp = (int)xp;
                    const float vx0 = xp - (float)ixp;
                    const float vx1 = 1.0f - vx0;
                    if (ixp == hc) vx[k] = vx1;
                    else if (ixp + 1 == hc) vx[k] = vx0;
                } else {        // :946-949: hist column ixp+1 / ixp+2
                    const float xp = ((double)x + 0.5) / (double)cs - 0.5;
                    const int ixp = (int)floor(xp);
                    const float vx0 = xp - ixp;
                    const float vx1 = 1.0 - vx0;
                    if (ixp + 1 == hc) vx[k] = vx1;
                    else if (ixp + 2 == hc) vx[k] = vx0;
                }
            }
        }
        for (int y = y_lo; y <= y_hi; y++) {
            const float yp = ((float)y + 0.5) / (float)cs - 0.5;  // :823-826 (double arithmetic, float result)
            const int iyp = (int)floor(yp);
            const float vy0 = yp - iyp;
            const float vy1 = 1.0 - vy0;
            float wy;
            if (iyp + 1 == hr) wy = vy1;
            else if (iyp + 2 == hr) wy = vy0;
            else continue;
            const unsigned *row = pk + (size_t)y * g.cols + xc;
            // fetch the whole candidate row first (independent loads in flight), then vote in raster order
            unsigned pv[FHOG_MAXW];
            if (vec) {
#pragma unroll
                for (int m = 0; m < FHOG_MAXW / 4; m++) {  // a load clamped at the row end only feeds slots beyond x_hi
                    const uint4 q = *reinterpret_cast<const uint4 *>(row + min(4 * m, g.cols - 4 - xc));
                    pv[4 * m] = q.x; pv[4 * m + 1] = q.y; pv[4 * m + 2] = q.z; pv[4 * m + 3] = q.w;
                }
            } else {
#pragma unroll
                for (int k = 0; k < FHOG_MAXW; k++) pv[k] = row[min(k, x_hi - xc)];
            }
#pragma unroll
            for (int k = 0; k < FHOG_MAXW; k++) {
                if (vx[k] >= 0.f) {  // weights are in [0, 1]; -1 marks "no vote"
                    const unsigned p = pv[k];
                    const float v = sqrtf((float)(p >> 5));
                    // :863-870 weights vy*(vx*v)  vs  :951-954 (vy*vx)*v
                    b[p & 31] += body[k] ? wy * (vx[k] * v) : wy * vx[k] * v;
                }
            }
        }
    }
    const int HC = g.cells_nc + 2;
    float *dst = hist + (((size_t)blockIdx.z * (g.cells_nr + 2) + hr) * HC + hc) * 18;
    float e = 0.f;
#pragma unroll
    for (int o = 0; o < 18; o++) dst[o] = b[o];
#pragma unroll
    for (int o = 0; o < 9; o++) e += (b[o] + b[o + 9]) * (b[o] + b[o + 9]);  // :959-968
    norm[((size_t)blockIdx.z * g.cells_nr + (hr - 1)) * g.cells_nc + (hc - 1)] = e;
}

__device__ __forceinline__ float fhog_sum4(const float (&h)[4]) { return (h[0] + h[2]) + (h[1] + h[3]); }

// K15: out[feat][xx][yy] (feature-major, then column, rows fastest: the order rcpp_fhog.cpp:29-38 emits)
__global__ void __launch_bounds__(256) fhog_features(const float *__restrict__ hist, const float *__restrict__ norm,
                                                     float *__restrict__ out, FhogGeom g)
{
    const int y = blockIdx.x * 64 + (threadIdx.x & 63);  // rows fastest: coalesced stores
    const int x = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (y >= g.hog_nr || x >= g.hog_nc) return;
    const float *nm = norm + (size_t)blockIdx.z * g.cells_nr * g.cells_nc;
    const int NC = g.cells_nc;
#define NORM(r, c) nm[(size_t)(r) * NC + (c)]
    const float z1[4] = {NORM(y + 1, x + 1), NORM(y, x + 1), NORM(y + 1, x), NORM(y, x)};
    const float z2[4] = {NORM(y + 1, x + 2), NORM(y, x + 2), NORM(y + 1, x + 1), NORM(y, x + 1)};
    const float z3[4] = {NORM(y + 2, x + 1), NORM(y + 1, x + 1), NORM(y + 2, x), NORM(y + 1, x)};
    const float z4[4] = {NORM(y + 2, x + 2), NORM(y + 1, x + 2), NORM(y + 2, x + 1), NORM(y + 1, x + 1)};
#undef NORM
    const float eps = 0.0001;
    float nn[4], n[4], t[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int l = 0; l < 4; l++) {
        nn[l] = 0.2f * sqrtf(z1[l] + z2[l] + z3[l] + z4[l] + eps);  // :997
        n[l] = 0.1f / nn[l];                                        // :998
    }
    const float *h = hist + (((size_t)blockIdx.z * (g.cells_nr + 2) + (y + 2)) * (g.cells_nc + 2) + (x + 2)) * 18;
    float hv[18];
#pragma unroll
    for (int o = 0; o < 18; o++) hv[o] = h[o];
    const size_t plane = (size_t)g.out_nr * g.out_nc;
    float *dst = out + (size_t)blockIdx.z * plane * 31 + (size_t)(x + g.off_c) * g.out_nr + (y + g.off_r);
#pragma unroll
    for (int o = 0; o < 18; o += 3) {  // contrast-sensitive, :1005-1017
        float h0[4], h1[4], h2[4];
#pragma unroll
        for (int l = 0; l < 4; l++) {
            h0[l] = fminf(hv[o], nn[l]) * n[l];
            h1[l] = fminf(hv[o + 1], nn[l]) * n[l];
            h2[l] = fminf(hv[o + 2], nn[l]) * n[l];
            t[l] += h0[l] + h1[l] + h2[l];
        }
        dst[(size_t)o * plane] = fhog_sum4(h0);
        dst[(size_t)(o + 1) * plane] = fhog_sum4(h1);
        dst[(size_t)(o + 2) * plane] = fhog_sum4(h2);
    }
#pragma unroll
    for (int l = 0; l < 4; l++) t[l] *= (float)(2 * 0.2357);  // :1019
#pragma unroll
    for (int o = 0; o < 9; o += 3) {   // contrast-insensitive, :1022-1033
        const float t0 = hv[o] + hv[o + 9], t1 = hv[o + 1] + hv[o + 10], t2 = hv[o + 2] + hv[o + 11];
        float h0[4], h1[4], h2[4];
#pragma unroll
        for (int l = 0; l < 4; l++) {
            h0[l] = fminf(t0, nn[l]) * n[l];
            h1[l] = fminf(t1, nn[l]) * n[l];
            h2[l] = fminf(t2, nn[l]) * n[l];
        }
        dst[(size_t)(o + 18) * plane] = fhog_sum4(h0);
        dst[(size_t)(o + 19) * plane] = fhog_sum4(h1);
        dst[(size_t)(o + 20) * plane] = fhog_sum4(h2);
    }
#pragma unroll
    for (int l = 0; l < 4; l++) dst[(size_t)(27 + l) * plane] = t[l];  // texture, :1040-1043
}

// cell_size == 1 (fhog.h:499-694): norm = squared gradient length of the pixel (0 on the image border), one
// contrast-sensitive and one contrast-insensitive feature per pixel (its orientation), texture features from the four
// 2x2 block norms.  out[feat][xx][yy]; the other 25 planes of a pixel are zero (init_hog_zero_everything).
__global__ void __launch_bounds__(256) fhog_features_cs1(const unsigned *__restrict__ packed, float *__restrict__ out, FhogGeom g)
{
    const int y = blockIdx.x * 64 + (threadIdx.x & 63);  // rows fastest: coalesced stores
    const int x = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (y >= g.hog_nr || x >= g.hog_nc) return;
    const unsigned *pk = packed + (size_t)blockIdx.z * g.rows * g.cols;
#define NORM(r, c) ((float)(pk[(size_t)(r) * g.cols + (c)] >> 5))
    const float z1[4] = {NORM(y + 1, x + 1), NORM(y, x + 1), NORM(y + 1, x), NORM(y, x)};
    const float z2[4] = {NORM(y + 1, x + 2), NORM(y, x + 2), NORM(y + 1, x + 1), NORM(y, x + 1)};
    const float z3[4] = {NORM(y + 2, x + 1), NORM(y + 1, x + 1), NORM(y + 2, x), NORM(y + 1, x)};
    const float z4[4] = {NORM(y + 2, x + 2), NORM(y + 1, x + 2), NORM(y + 2, x + 1), NORM(y + 1, x + 1)};
#undef NORM
    const unsigned centre = pk[(size_t)(y + 1) * g.cols + (x + 1)];
    const int a = centre & 31;
    const float temp0 = sqrtf((float)(centre >> 5));  // :651
    const float eps = 0.0001;
    float h0[4], t[4];
#pragma unroll
    for (int l = 0; l < 4; l++) {
        const float nn = 0.2f * sqrtf(z1[l] + z2[l] + z3[l] + z4[l] + eps);
        const float n = 0.1f / nn;
        h0[l] = fminf(temp0, nn) * n;
        t[l] = (0.0f + h0[l]) * (float)(2 * 0.2357);
    }
    const float vv = fhog_sum4(h0);
    const size_t plane = (size_t)g.out_nr * g.out_nc;
    float *dst = out + (size_t)blockIdx.z * plane * 31 + (size_t)(x + g.off_c) * g.out_nr + (y + g.off_r);
#pragma unroll
    for (int f = 0; f < 27; f++) dst[(size_t)f * plane] = (f == a || f == a % 9 + 18) ? vv : 0.f;
#pragma unroll
    for (int l = 0; l < 4; l++) dst[(size_t)(27 + l) * plane] = t[l];
}

namespace {

size_t fhog_ws_bytes(const FhogGeom &g, int nf, bool fused)
{
    if (g.cs == 1) return align_up(sizeof(unsigned) * (size_t)g.rows * g.cols * nf, 256) + 4096;  // no histograms
    return (fused ? 0 : align_up(sizeof(unsigned) * (size_t)g.rows * g.cols * nf, 256)) +  // fhog_hist8 needs no pixel plane
           align_up(sizeof(float) * (size_t)(g.cells_nr + 2) * (g.cells_nc + 2) * 18 * nf, 256) +
           align_up(sizeof(float) * (size_t)g.cells_nr * g.cells_nc * nf, 256) + 4096;
}

// d_rgb: nf interlaced RGB frames; d_out: nf * 31 * out_nc * out_nr floats
imgfd_status fhog_device(imgfd_ctx *ctx, const uint8_t *d_rgb, size_t frame_stride, const FhogGeom &g, int nf, float *d_out)
{
    const bool fused = ctx->tune.fhog_fused && fhog_fused_supported(g, d_rgb, frame_stride);
    unsigned *packed = fused ? nullptr : (unsigned *)ws_alloc(ctx, sizeof(unsigned) * (size_t)g.rows * g.cols * nf);
    float *hist = g.cs == 1 ? nullptr : (float *)ws_alloc(ctx, sizeof(float) * (size_t)(g.cells_nr + 2) * (g.cells_nc + 2) * 18 * nf);
    float *norm = g.cs == 1 ? nullptr : (float *)ws_alloc(ctx, sizeof(float) * (size_t)g.cells_nr * g.cells_nc * nf);
    if ((!fused && !packed) || (g.cs != 1 && (!hist || !norm))) return imgfd_fail(ctx, IMGFD_ERR_OOM, "workspace reservation too small");
    const size_t out_n = (size_t)31 * g.out_nr * g.out_nc * nf;
    if (g.out_nr != g.hog_nr || g.out_nc != g.hog_nc)  // init_hog: zero border of the padded output
        IMGFD_HIP(ctx, hipMemsetAsync(d_out, 0, out_n * sizeof(float), ctx->stream));
    if (fused) {  // K13 + K14 in one kernel: no per-pixel plane between them
        IMGFD_TRY(fhog_fused_hist(ctx, d_rgb, frame_stride, g, nf, hist, norm));
        hipLaunchKernelGGL(fhog_features, dim3(ceil_div(g.hog_nr, 64), ceil_div(g.hog_nc, 4), nf), dim3(256), 0, ctx->stream,
                           hist, norm, d_out, g);
        IMGFD_HIP(ctx, hipGetLastError());
        return IMGFD_OK;
    }
    if (g.cols % 4 == 0 && (size_t)d_rgb % 4 == 0 && frame_stride % 4 == 0 && (size_t)packed % 16 == 0)
        hipLaunchKernelGGL(fhog_grad_orient4, dim3(ceil_div(g.cols / 4, 256), g.rows, nf), dim3(256), 0, ctx->stream, d_rgb,
                           frame_stride, packed, g);
    else
        hipLaunchKernelGGL(fhog_grad_orient, dim3(ceil_div(g.cols, 256), g.rows, nf), dim3(256), 0, ctx->stream, d_rgb,
                           frame_stride, packed, g);
    if (g.cs == 1) {
        hipLaunchKernelGGL(fhog_features_cs1, dim3(ceil_div(g.hog_nr, 64), ceil_div(g.hog_nc, 4), nf), dim3(256), 0, ctx->stream,
                           packed, d_out, g);
        IMGFD_HIP(ctx, hipGetLastError());
        return IMGFD_OK;
    }
    hipLaunchKernelGGL(fhog_cell_hist, dim3(ceil_div(g.cells_nc, 8), ceil_div(g.cells_nr, 8), nf), dim3(64), 0,
                       ctx->stream, packed, hist, norm, g);
    hipLaunchKernelGGL(fhog_features, dim3(ceil_div(g.hog_nr, 64), ceil_div(g.hog_nc, 4), nf), dim3(256), 0, ctx->stream,
                       hist, norm, d_out, g);
    IMGFD_HIP(ctx, hipGetLastError());
    return IMGFD_OK;
}

}  // namespace

extern "C" {

imgfd_status imgfd_fhog_size(int rows, int cols, int cell_size, int filter_rows_padding, int filter_cols_padding,
                             int *hog_nr, int *hog_nc)
{
    if (!hog_nr || !hog_nc) return IMGFD_ERR_INVALID;
    *hog_nr = *hog_nc = 0;
    if (rows < 0 || cols < 0 || cell_size < 1 || filter_rows_padding < 1 || filter_cols_padding < 1) return IMGFD_ERR_INVALID;
    FhogGeom g;
    if (fhog_geometry(rows, cols, cell_size, filter_rows_padding, filter_cols_padding, &g)) { *hog_nr = g.out_nr; *hog_nc = g.out_nc; }
    return IMGFD_OK;
}

// hog: library-allocated floats (imgfd_fhog, imgfd_fhog_i32), or -- hog_f64 -- the caller's vector of hog_cap doubles
// (imgfd_fhog_f64out: what rcpp_fhog.cpp:29-38 fills element by element)
static imgfd_status fhog_host(imgfd_ctx *ctx, const void *rgb, int kind, int rows, int cols, int cell_size, int filter_rows_padding,
                              int filter_cols_padding, float **hog, int *hog_nr, int *hog_nc, double *hog_f64 = nullptr, int64_t hog_cap = 0)
{
    if (!ctx) return IMGFD_ERR_INVALID;
    float *unused_hog = nullptr;
    if (hog_f64) hog = &unused_hog;
    if (!rgb || !hog || !hog_nr || !hog_nc || rows < 0 || cols < 0 || cell_size < 1 || filter_rows_padding < 1 || filter_cols_padding < 1 ||
        !frame_fits(rows, cols, 3))
        return imgfd_fail(ctx, IMGFD_ERR_INVALID, "imgfd_fhog: bad argument (DLIB_ASSERT of fhog.h:712-720)");
    *hog = nullptr; *hog_nr = 0; *hog_nc = 0;
    FhogGeom g;
    if (!fhog_geometry(rows, cols, cell_size, filter_rows_padding, filter_cols_padding, &g)) return IMGFD_OK;  // hog.clear()
    IMGFD_HIP(ctx, hipSetDevice(ctx->device));
    const size_t in_bytes = (size_t)3 * rows * cols, out_n = (size_t)31 * g.out_nr * g.out_nc;
    if (hog_f64 && (int64_t)out_n > hog_cap) return imgfd_fail(ctx, IMGFD_ERR_INVALID, "imgfd_fhog_f64out: the output vector is too short (imgfd_fhog_size gives 31 * hog_nr * hog_nc)");
    IMGFD_TRY(ws_reserve(ctx, fhog_ws_bytes(g, 1, false) + align_up(in_bytes, 256) + align_up(out_n * sizeof(float), 256) +
                              upload_stage_bytes(kind, in_bytes) + (hog_f64 ? align_up(out_n * sizeof(double), 256) : 0)));
    uint8_t *d_in = (uint8_t *)ws_alloc(ctx, in_bytes);
    float *d_out = (float *)ws_alloc(ctx, out_n * sizeof(float));
    double *d_wide = hog_f64 ? (double *)ws_alloc(ctx, out_n * sizeof(double)) : nullptr;
    if (!d_in || !d_out || (hog_f64 && !d_wide)) return imgfd_fail(ctx, IMGFD_ERR_OOM, "workspace reservation too small");
    IMGFD_TRY(upload_image(ctx, rgb, kind, in_bytes, d_in));
    IMGFD_TRY(fhog_device(ctx, d_in, in_bytes, g, 1, d_out));
    if (hog_f64) {
        IMGFD_TRY(download_widened(ctx, d_out, false, out_n, d_wide, hog_f64));
        IMGFD_HIP(ctx, hipStreamSynchronize(ctx->stream));
        *hog_nr = g.out_nr; *hog_nc = g.out_nc;
        return IMGFD_OK;
    }
    float *h = (float *)malloc(out_n * sizeof(float));
    if (!h) return imgfd_fail(ctx, IMGFD_ERR_OOM, "malloc of the fhog output failed");
    IMGFD_HIP(ctx, hipMemcpyAsync(h, d_out, out_n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    IMGFD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *hog = h; *hog_nr = g.out_nr; *hog_nc = g.out_nc;
    return IMGFD_OK;
}

imgfd_status imgfd_fhog(imgfd_ctx *ctx, const uint8_t *rgb, int rows, int cols, int cell_size, int filter_rows_padding,
                        int filter_cols_padding, float **hog, int *hog_nr, int *hog_nc)
{
    return imgfd_guard(ctx, [&] { return fhog_host(ctx, rgb, IMGFD_SRC_U8, rows, cols, cell_size, filter_rows_padding, filter_cols_padding, hog, hog_nr, hog_nc); });
}

imgfd_status imgfd_fhog_i32(imgfd_ctx *ctx, const int32_t *x, int rows, int cols, int cell_size, int filter_rows_padding,
                            int filter_cols_padding, float **hog, int *hog_nr, int *hog_nc)
{
    return imgfd_guard(ctx, [&] { return fhog_host(ctx, x, IMGFD_SRC_I32, rows, cols, cell_size, filter_rows_padding, filter_cols_padding, hog, hog_nr, hog_nc); });
}

imgfd_status imgfd_fhog_f64out(imgfd_ctx *ctx, const int32_t *x, int rows, int cols, int cell_size, int filter_rows_padding,
                               int filter_cols_padding, double *hog, int64_t hog_cap, int *hog_nr, int *hog_nc)
{
    if (!hog) return imgfd_fail(ctx, IMGFD_ERR_INVALID, "imgfd_fhog_f64out: no output vector");
    return imgfd_guard(ctx, [&] { return fhog_host(ctx, x, IMGFD_SRC_I32, rows, cols, cell_size, filter_rows_padding, filter_cols_padding, nullptr, hog_nr, hog_nc, hog, hog_cap); });
}

imgfd_status imgfd_fhog_dev(imgfd_ctx *ctx, const uint8_t *d_rgb, int n_frames, int rows, int cols, size_t frame_stride_bytes,
                            int cell_size, int filter_rows_padding, int filter_cols_padding, float *d_hog)
{
    if (!ctx) return IMGFD_ERR_INVALID;
    if (!d_rgb || !d_hog || n_frames < 0 || rows < 0 || cols < 0 || cell_size < 1 || filter_rows_padding < 1 || filter_cols_padding < 1 ||
        !frame_fits(rows, cols, 3))
        return imgfd_fail(ctx, IMGFD_ERR_INVALID, "imgfd_fhog_dev: bad argument");
    FhogGeom g;
    if (!n_frames || !fhog_geometry(rows, cols, cell_size, filter_rows_padding, filter_cols_padding, &g)) return IMGFD_OK;
    IMGFD_HIP(ctx, hipSetDevice(ctx->device));
    const bool fused = ctx->tune.fhog_fused && fhog_fused_supported(g, d_rgb, frame_stride_bytes);
    const size_t per_frame = fhog_ws_bytes(g, 1, fused);
    const int chunk = (int)std::max<size_t>(1, std::min<size_t>((size_t)n_frames, ((size_t)2 << 30) / per_frame));
    IMGFD_TRY(ws_reserve(ctx, fhog_ws_bytes(g, chunk, fused)));
    const size_t out_n = (size_t)31 * g.out_nr * g.out_nc;
    for (int f0 = 0; f0 < n_frames; f0 += chunk) {
        const int nf = std::min(chunk, n_frames - f0);
        ctx->ws_used = 0;
        IMGFD_TRY(fhog_device(ctx, d_rgb + (size_t)f0 * frame_stride_bytes, frame_stride_bytes, g, nf, d_hog + (size_t)f0 * out_n));
    }
    return IMGFD_OK;
}

}  // extern "C"
