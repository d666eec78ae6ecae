/*
 * oracle/fhog_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of dlib's Felzenszwalb HOG as the reference R package calls it:
 *   dlib_fhog()                       image.dlib/src/rcpp_fhog.cpp:10-46
 *   extract_fhog_features (interlaced) image.dlib/inst/dlib-19.20/dlib/image_transforms/fhog.h:1104-1113
 *   impl_extract_fhog_features        fhog.h:702-1046   (cell_size > 1)
 *   impl_extract_fhog_features_cell_size_1  fhog.h:499-694 (orc_fhog_cs1 below)
 *   get_gradient  rgb scalar :24-59, rgb simd8 :147-275 ; init_hog :448-471
 * in the arithmetic of the build CRAN makes (x86-64 -O2: dlib's SSE2 paths, no FMA): the histogram pass
 * handles groups of 8 columns in float "SIMD" arithmetic (:828-918) and the remaining columns in the scalar
 * tail (:920-955), which differ in rounding order and in the colour-channel tie break; sum(simd4f) is
 * (i0+i2)+(i1+i3) (dlib/simd/simd4f.h:549-566, SSE2 branch).
 * Pinned by oracle/_ref/libref_dlib.so (the reference headers compiled in place) in tests/test_oracle_dlib.py,
 * which is itself pinned by dlib's own known-answer test (dlib/test/fhog.cpp:34-53,156-214).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

static const float DIRS[9][2] = {{1.0000f, 0.0000f}, {0.9397f, 0.3420f}, {0.7660f, 0.6428f}, {0.500f, 0.8660f},
                                 {0.1736f, 0.9848f}, {-0.1736f, 0.9848f}, {-0.5000f, 0.8660f}, {-0.7660f, 0.6428f},
                                 {-0.9397f, 0.3420f}};

/* fhog.h:779-812 */
ORC_API int orc_fhog_dims(int rows, int cols, int cs, int pad_r, int pad_c, int *hog_nr, int *hog_nc)
{
    *hog_nr = *hog_nc = 0;
    if (cs < 1 || pad_r < 1 || pad_c < 1) return -1;
    if (cs == 1) { /* fhog.h:536-551 */
        if (rows <= 2 || cols <= 2) return 0;
        *hog_nr = rows - 2 + pad_r - 1;
        *hog_nc = cols - 2 + pad_c - 1;
        return 0;
    }
    const int cells_nr = (int)((float)rows / (float)cs + 0.5);
    const int cells_nc = (int)((float)cols / (float)cs + 0.5);
    if (cells_nr == 0 || cells_nc == 0) return 0;
    const int nr = cells_nr - 2 > 0 ? cells_nr - 2 : 0, nc = cells_nc - 2 > 0 ? cells_nc - 2 : 0;
    if (nr == 0 || nc == 0) return 0;
    *hog_nr = nr + pad_r - 1;
    *hog_nc = nc + pad_c - 1;
    return 0;
}

static inline const unsigned char *px(const unsigned char *rgb, int cols, int r, int c) { return rgb + 3 * ((size_t)r * cols + c); }

/* impl_extract_fhog_features_cell_size_1, fhog.h:499-694: every pixel is its own cell; norm = squared gradient length,
 * one sensitive + one insensitive feature per pixel, texture features from the 4 block norms. */
static int orc_fhog_cs1(const unsigned char *rgb, int rows, int cols, int pad_r, int pad_c, float *hog)
{
    int hnr, hnc;
    orc_fhog_dims(rows, cols, 1, pad_r, pad_c, &hnr, &hnc);
    if (hnr == 0 || hnc == 0) return 0;
    const int hog_nr = rows - 2, hog_nc = cols - 2;
    const int off_r = (pad_r - 1) / 2, off_c = (pad_c - 1) / 2;
    memset(hog, 0, sizeof(float) * (size_t)hnr * hnc * 31); /* init_hog_zero_everything */
    float *norm = (float *)calloc((size_t)rows * cols, sizeof(float)); /* zero_border_pixels(norm,1,1); interior overwritten */
    unsigned char *angle = (unsigned char *)calloc((size_t)rows * cols, 1);
    const int visible_nr = rows - 1, visible_nc = cols - 1;
    for (int y = 1; y < visible_nr; y++) {
        int x;
        for (x = 1; x < visible_nc - 7; x += 8)
            for (int k = 0; k < 8; k++) { /* simd8 lanes: later channel wins ties */
                const int xx = x + k;
                int g[3][2], len[3];
                for (int ch = 0; ch < 3; ch++) {
                    g[ch][0] = (int)px(rgb, cols, y, xx + 1)[ch] - (int)px(rgb, cols, y, xx - 1)[ch];
                    g[ch][1] = (int)px(rgb, cols, y + 1, xx)[ch] - (int)px(rgb, cols, y - 1, xx)[ch];
                    len[ch] = g[ch][0] * g[ch][0] + g[ch][1] * g[ch][1];
                }
                int tx, ty, tl;
                if (len[0] > len[1]) { tx = g[0][0]; ty = g[0][1]; tl = len[0]; } else { tx = g[1][0]; ty = g[1][1]; tl = len[1]; }
                if (!(tl > len[2])) { tx = g[2][0]; ty = g[2][1]; tl = len[2]; }
                const float gx = (float)tx, gy = (float)ty;
                float best_dot = 0;
                int best_o = 0;
                for (int o = 0; o < 9; o++) {
                    float dot = gx * DIRS[o][0] + gy * DIRS[o][1];
                    if (dot > best_dot) { best_dot = dot; best_o = o; }
                    dot *= -1;
                    if (dot > best_dot) { best_dot = dot; best_o = o + 9; }
                }
                norm[(size_t)y * cols + xx] = (float)tl;
                angle[(size_t)y * cols + xx] = (unsigned char)best_o;
            }
        for (; x < visible_nc; x++) { /* scalar tail: earlier channel wins ties */
            int g[3][2];
            float len[3];
            for (int ch = 0; ch < 3; ch++) {
                g[ch][0] = (int)px(rgb, cols, y, x + 1)[ch] - (int)px(rgb, cols, y, x - 1)[ch];
                g[ch][1] = (int)px(rgb, cols, y + 1, x)[ch] - (int)px(rgb, cols, y - 1, x)[ch];
                len[ch] = (float)g[ch][0] * (float)g[ch][0] + (float)g[ch][1] * (float)g[ch][1];
            }
            float gx = (float)g[0][0], gy = (float)g[0][1], v = len[0];
            if (len[1] > v) { v = len[1]; gx = (float)g[1][0]; gy = (float)g[1][1]; }
            if (len[2] > v) { v = len[2]; gx = (float)g[2][0]; gy = (float)g[2][1]; }
            float best_dot = 0;
            int best_o = 0;
            for (int o = 0; o < 9; o++) {
                const float dot = DIRS[o][0] * gx + DIRS[o][1] * gy;
                if (dot > best_dot) { best_dot = dot; best_o = o; }
                else if (-dot > best_dot) { best_dot = -dot; best_o = o + 9; }
            }
            norm[(size_t)y * cols + x] = v;
            angle[(size_t)y * cols + x] = (unsigned char)best_o;
        }
    }
#define N1(r, c) norm[(size_t)(r) * cols + (c)]
    const float eps = 0.0001;
    for (int y = 0; y < hog_nr; y++)
        for (int x = 0; x < hog_nc; x++) {
            float *out = hog + ((size_t)(y + off_r) * hnc + (x + off_c)) * 31;
            const float z1[4] = {N1(y + 1, x + 1), N1(y, x + 1), N1(y + 1, x), N1(y, x)};
            const float z2[4] = {N1(y + 1, x + 2), N1(y, x + 2), N1(y + 1, x + 1), N1(y, x + 1)};
            const float z3[4] = {N1(y + 2, x + 1), N1(y + 1, x + 1), N1(y + 2, x), N1(y + 1, x)};
            const float z4[4] = {N1(y + 2, x + 2), N1(y + 1, x + 2), N1(y + 2, x + 1), N1(y + 1, x + 1)};
            const float temp0 = sqrtf(N1(y + 1, x + 1));
            float h0[4], t[4];
            for (int l = 0; l < 4; l++) {
                const float nn = 0.2f * sqrtf(z1[l] + z2[l] + z3[l] + z4[l] + eps);
                const float n = 0.1f / nn;
                h0[l] = (temp0 < nn ? temp0 : nn) * n;
                t[l] = (0.0f + h0[l]) * (float)(2 * 0.2357);
            }
            const float vv = (h0[0] + h0[2]) + (h0[1] + h0[3]);
            const int a = angle[(size_t)(y + 1) * cols + (x + 1)];
            out[a] = vv;
            out[a % 9 + 18] = vv;
            out[27] = t[0]; out[28] = t[1]; out[29] = t[2]; out[30] = t[3];
        }
#undef N1
    free(norm); free(angle);
    return 0;
}

/* hog: (*hog_nr) x (*hog_nc) x 31 floats, row-major AoS like array2d<matrix<float,31,1>>.
 * hist_out (optional): (cells_nr+2) x (cells_nc+2) x 18 ; norm_out (optional): cells_nr x cells_nc */
ORC_API int orc_fhog(const unsigned char *rgb, int rows, int cols, int cs, int pad_r, int pad_c, float *hog,
                     float *hist_out, float *norm_out)
{
    int hnr, hnc;
    if (orc_fhog_dims(rows, cols, cs, pad_r, pad_c, &hnr, &hnc)) return -1;
    if (hnr == 0 || hnc == 0) return 0;
    if (cs == 1) return orc_fhog_cs1(rgb, rows, cols, pad_r, pad_c, hog);
    const int cells_nr = (int)((float)rows / (float)cs + 0.5);
    const int cells_nc = (int)((float)cols / (float)cs + 0.5);
    const int hog_nr = cells_nr - 2, hog_nc = cells_nc - 2;
    const int HR = cells_nr + 2, HC = cells_nc + 2;
    float *hist = (float *)calloc((size_t)HR * HC * 18, sizeof(float));
    float *norm = (float *)calloc((size_t)cells_nr * cells_nc, sizeof(float));
#define HIST(r, c, o) hist[(((size_t)(r)) * HC + (c)) * 18 + (o)]
    const int off_r = (pad_r - 1) / 2, off_c = (pad_c - 1) / 2;
    memset(hog, 0, sizeof(float) * (size_t)hnr * hnc * 31); /* init_hog zeroes the border; the interior is overwritten */
    const long vr = (long)cells_nr * cs < rows ? (long)cells_nr * cs : rows;
    const long vc = (long)cells_nc * cs < cols ? (long)cells_nc * cs : cols;
    const int visible_nr = (int)vr - 1, visible_nc = (int)vc - 1;

    for (int y = 1; y < visible_nr; y++) {
        const float yp = ((float)y + 0.5) / (float)cs - 0.5;
        const int iyp = (int)floor(yp);
        const float vy0 = yp - iyp;
        const float vy1 = 1.0 - vy0;
        int x;
        for (x = 1; x < visible_nc - 7; x += 8) {
            for (int k = 0; k < 8; k++) { /* the eight lanes are independent */
                const int xx = x + k;
                int g[3][2], len[3];
                for (int ch = 0; ch < 3; ch++) {
                    g[ch][0] = (int)px(rgb, cols, y, xx + 1)[ch] - (int)px(rgb, cols, y, xx - 1)[ch];
                    g[ch][1] = (int)px(rgb, cols, y + 1, xx)[ch] - (int)px(rgb, cols, y - 1, xx)[ch];
                    len[ch] = g[ch][0] * g[ch][0] + g[ch][1] * g[ch][1];
                }
                /* :265-274: red only if strictly longer than green, that only if strictly longer than blue */
                int tx, ty, tl;
                if (len[0] > len[1]) { tx = g[0][0]; ty = g[0][1]; tl = len[0]; } else { tx = g[1][0]; ty = g[1][1]; tl = len[1]; }
                if (!(tl > len[2])) { tx = g[2][0]; ty = g[2][1]; tl = len[2]; }
                const float gx = (float)tx, gy = (float)ty;
                float v = (float)tl;
                const float xp = ((float)xx + 0.5f) / (float)cs + 0.5f;
                const int ixp = (int)xp;
                float vx0 = xp - (float)ixp;
                float vx1 = 1.0f - vx0;
                v = sqrtf(v);
                float best_dot = 0;
                int best_o = 0;
                for (int o = 0; o < 9; o++) {
                    float dot = gx * DIRS[o][0] + gy * DIRS[o][1];
                    if (dot > best_dot) { best_dot = dot; best_o = o; }
                    dot *= -1;
                    if (dot > best_dot) { best_dot = dot; best_o = o + 9; }
                }
                vx1 *= v;
                vx0 *= v;
                HIST(iyp + 1, ixp, best_o) += vy1 * vx1;
                HIST(iyp + 2, ixp, best_o) += vy0 * vx1;
                HIST(iyp + 1, ixp + 1, best_o) += vy1 * vx0;
                HIST(iyp + 2, ixp + 1, best_o) += vy0 * vx0;
            }
        }
        for (; x < visible_nc; x++) { /* scalar tail, :920-955 */
            int g[3][2];
            float len[3];
            for (int ch = 0; ch < 3; ch++) {
                g[ch][0] = (int)px(rgb, cols, y, x + 1)[ch] - (int)px(rgb, cols, y, x - 1)[ch];
                g[ch][1] = (int)px(rgb, cols, y + 1, x)[ch] - (int)px(rgb, cols, y - 1, x)[ch];
                len[ch] = (float)g[ch][0] * (float)g[ch][0] + (float)g[ch][1] * (float)g[ch][1];
            }
            float gx = (float)g[0][0], gy = (float)g[0][1], v = len[0];
            if (len[1] > v) { v = len[1]; gx = (float)g[1][0]; gy = (float)g[1][1]; }
            if (len[2] > v) { v = len[2]; gx = (float)g[2][0]; gy = (float)g[2][1]; }
            float best_dot = 0;
            int best_o = 0;
            for (int o = 0; o < 9; o++) {
                const float dot = DIRS[o][0] * gx + DIRS[o][1] * gy;
                if (dot > best_dot) { best_dot = dot; best_o = o; }
                else if (-dot > best_dot) { best_dot = -dot; best_o = o + 9; }
            }
            v = sqrtf(v);
            const float xp = ((double)x + 0.5) / (double)cs - 0.5;
            const int ixp = (int)floor(xp);
            const float vx0 = xp - ixp;
            const float vx1 = 1.0 - vx0;
            HIST(iyp + 1, ixp + 1, best_o) += vy1 * vx1 * v;
            HIST(iyp + 2, ixp + 1, best_o) += vy0 * vx1 * v;
            HIST(iyp + 1, ixp + 2, best_o) += vy1 * vx0 * v;
            HIST(iyp + 2, ixp + 2, best_o) += vy0 * vx0 * v;
        }
    }
    /* energy, :959-968 */
    for (int r = 0; r < cells_nr; r++)
        for (int c = 0; c < cells_nc; c++)
            for (int o = 0; o < 9; o++)
                norm[(size_t)r * cells_nc + c] += (HIST(r + 1, c + 1, o) + HIST(r + 1, c + 1, o + 9)) *
                                                   (HIST(r + 1, c + 1, o) + HIST(r + 1, c + 1, o + 9));
#define NORM(r, c) norm[(size_t)(r) * cells_nc + (c)]
#define SUM4(h) (((h)[0] + (h)[2]) + ((h)[1] + (h)[3]))
    const float eps = 0.0001;
    for (int y = 0; y < hog_nr; y++) {
        const int yy = y + off_r;
        for (int x = 0; x < hog_nc; x++) {
            const int xx = x + off_c;
            float *out = hog + ((size_t)yy * hnc + xx) * 31;
            const float z1[4] = {NORM(y + 1, x + 1), NORM(y, x + 1), NORM(y + 1, x), NORM(y, x)};
            const float z2[4] = {NORM(y + 1, x + 2), NORM(y, x + 2), NORM(y + 1, x + 1), NORM(y, x + 1)};
            const float z3[4] = {NORM(y + 2, x + 1), NORM(y + 1, x + 1), NORM(y + 2, x), NORM(y + 1, x)};
            const float z4[4] = {NORM(y + 2, x + 2), NORM(y + 1, x + 2), NORM(y + 2, x + 1), NORM(y + 1, x + 1)};
            float nn[4], n[4], t[4] = {0, 0, 0, 0};
            for (int l = 0; l < 4; l++) {
                nn[l] = 0.2f * sqrtf(z1[l] + z2[l] + z3[l] + z4[l] + eps);
                n[l] = 0.1f / nn[l];
            }
            const float *h = &HIST(y + 2, x + 2, 0);
            for (int o = 0; o < 18; o += 3) {
                float h0[4], h1[4], h2[4];
                for (int l = 0; l < 4; l++) {
                    h0[l] = (h[o] < nn[l] ? h[o] : nn[l]) * n[l];
                    h1[l] = (h[o + 1] < nn[l] ? h[o + 1] : nn[l]) * n[l];
                    h2[l] = (h[o + 2] < nn[l] ? h[o + 2] : nn[l]) * n[l];
                }
                out[o] = SUM4(h0); out[o + 1] = SUM4(h1); out[o + 2] = SUM4(h2);
                for (int l = 0; l < 4; l++) t[l] += h0[l] + h1[l] + h2[l];
            }
            for (int l = 0; l < 4; l++) t[l] *= (float)(2 * 0.2357);
            for (int o = 0; o < 9; o += 3) {
                float h0[4], h1[4], h2[4];
                const float t0 = h[o] + h[o + 9], t1 = h[o + 1] + h[o + 9 + 1], t2 = h[o + 2] + h[o + 9 + 2];
                for (int l = 0; l < 4; l++) {
                    h0[l] = (t0 < nn[l] ? t0 : nn[l]) * n[l];
                    h1[l] = (t1 < nn[l] ? t1 : nn[l]) * n[l];
                    h2[l] = (t2 < nn[l] ? t2 : nn[l]) * n[l];
                }
                out[o + 18] = SUM4(h0); out[o + 19] = SUM4(h1); out[o + 20] = SUM4(h2);
            }
            out[27] = t[0]; out[28] = t[1]; out[29] = t[2]; out[30] = t[3];
        }
    }
    if (hist_out) memcpy(hist_out, hist, sizeof(float) * (size_t)HR * HC * 18);
    if (norm_out) memcpy(norm_out, norm, sizeof(float) * (size_t)cells_nr * cells_nc);
    free(hist); free(norm);
    return 0;
}
