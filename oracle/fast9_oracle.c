/*
 * oracle/fast9_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the reference FAST-9 detector,
 * image.CornerDetectionF9/src/f9.cpp (bnosac/image).  The reference's 5.7k-line
 * generated decision tree (f9.cpp:2953-5714 detect, :171-2941 score) is restated
 * as what it decides: "there exist 9 contiguous pixels on the 16-pixel radius-3
 * Bresenham ring that are all > cb or all < c_b".  Only tests/, smoke() and
 * bench.py's cpu_baseline leg may call this file.
 *
 * Pinning: the reference has no known-answer test for FAST-9; this restatement
 * is pinned against f9.cpp itself compiled in place (oracle/_ref/libref_f9.so)
 * in tests/test_oracle.py and against tests/golden/ vectors produced by
 * that build (chairs.pgm thr 80 -> 926 corners / 347 after NMS).
 */
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

/* makeOffsets f9.cpp:42-59 */
static const int RING_DX[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
static const int RING_DY[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};

/* is p a corner at threshold b?  cb/c_b saturate in unsigned char, f9.cpp:2962-2963 */
static int is_corner(const unsigned char *p, int stride, int b)
{
    int c = *p;
    unsigned char cb = (c > 255 - b) ? 255 : (unsigned char)(c + b);
    unsigned char c_b = (c < b) ? 0 : (unsigned char)(c - b);
    unsigned brighter = 0, darker = 0;
    for (int k = 0; k < 16; k++) {
        unsigned char v = p[RING_DX[k] + stride * RING_DY[k]];
        if (v > cb) brighter |= 1u << k;
        if (v < c_b) darker |= 1u << k;
    }
    brighter |= brighter << 16;
    darker |= darker << 16;
    for (int s = 0; s < 16; s++) {
        if (((brighter >> s) & 0x1FF) == 0x1FF) return 1;
        if (((darker >> s) & 0x1FF) == 0x1FF) return 1;
    }
    return 0;
}

/* cornerScore f9.cpp:171-2941: binary search bmin=b, bmax=255 */
static int corner_score(const unsigned char *p, int stride, unsigned char bstart)
{
    unsigned char bmin = bstart, bmax = 255;
    unsigned char b = ((int)bmax + (int)bmin) >> 1;
    for (;;) {
        if (is_corner(p, stride, b)) bmin = b; else bmax = b;
        if (bmin == bmax - 1 || bmin == bmax) return bmin;
        b = (bmin + bmax) >> 1;
    }
}

/*
 * detectCorners f9.cpp:66-82: detectAllCorners (:2953, raster order over
 * y in [3,h-3), x in [3,w-3)), optionally cornersScores (:2943-2951) and
 * nonMaxSuppression (:84-169: a corner is dropped iff one of its 8 neighbours
 * is a corner whose score is >= its own).  xy receives (x,y) int pairs; the
 * return value is the number of corners found (only the first cap are stored).
 * scores_out (optional, cap entries) receives the scores of the raw detections.
 */
ORC_API long orc_fast9(const unsigned char *im, int w, int h, int stride, int threshold,
                       int suppress_non_max, int *xy, long cap, int *scores_out)
{
    long n = 0, ncap = 1024;
    int *cx = (int *)malloc(sizeof(int) * 2 * ncap);
    unsigned char b = (unsigned char)threshold;
    for (int y = 3; y < h - 3; ++y)
        for (int x = 3; x < w - 3; ++x) {
            if (!is_corner(im + (long)y * stride + x, stride, b)) continue;
            if (n == ncap) { ncap *= 2; cx = (int *)realloc(cx, sizeof(int) * 2 * ncap); }
            cx[2 * n] = x; cx[2 * n + 1] = y; n++;
        }
    if (!suppress_non_max) {
        memcpy(xy, cx, sizeof(int) * 2 * (n < cap ? n : cap));
        free(cx);
        return n;
    }
    int *score = (int *)malloc(sizeof(int) * (n > 0 ? n : 1));
    int *map = (int *)malloc(sizeof(int) * (size_t)w * h);
    for (long i = 0; i < (long)w * h; i++) map[i] = -1;
    for (long i = 0; i < n; i++) {
        score[i] = corner_score(im + (long)cx[2 * i + 1] * stride + cx[2 * i], stride, b);
        map[(long)cx[2 * i + 1] * w + cx[2 * i]] = score[i];
        if (scores_out && i < cap) scores_out[i] = score[i];
    }
    long m = 0;
    for (long i = 0; i < n; i++) {
        int x = cx[2 * i], y = cx[2 * i + 1], s = score[i], keep = 1;
        for (int dy = -1; dy <= 1 && keep; dy++)
            for (int dx = -1; dx <= 1; dx++) {
                if (!dx && !dy) continue;
                int xx = x + dx, yy = y + dy;
                if (xx < 0 || yy < 0 || xx >= w || yy >= h) continue;
                if (map[(long)yy * w + xx] >= s) { keep = 0; break; }
            }
        if (keep) {
            if (m < cap) { xy[2 * m] = x; xy[2 * m + 1] = y; }
            m++;
        }
    }
    free(score); free(map); free(cx);
    return m;
}
