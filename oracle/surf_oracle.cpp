/*
 * oracle/surf_oracle.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of dlib's SURF as the reference R package calls it:
 *   dlib_surf_points()     image.dlib/src/rcpp_surf.cpp:10-53
 *   get_surf_points()      image.dlib/inst/dlib-19.20/dlib/image_keypoint/surf.h:237-288
 *   integral image         dlib/image_transforms/integral_image.h:33-96 (int32, wrapping), gray = (r+g+b)/3
 *                          (dlib/pixel.h:775-783), haar_x / haar_y :124-183
 *   hessian_pyramid        dlib/image_keypoint/hessian_pyramid.h:87-196 (build_pyramid(img,4,6,2)),
 *                          :324-446 (3x3x3 maximum test, quadratic interpolation), :453-506 (get_interest_points)
 *   dominant angle, 64-d descriptor   surf.h:75-232 ; centered_rect dlib/geometry/rectangle.h:363-376 ;
 *   double -> long rounding floor(v+0.5) dlib/geometry/vector.h:138-149 ; 3x3 inverse dlib/matrix/matrix_la.h:922-962
 * C++ only because get_surf_points orders the points with std::sort over reverse iterators (surf.h:268) and ties
 * must fall the same way.  PARITY: the reference has no test for surf.h / hessian_pyramid.h; this file is pinned
 * against dlib compiled in place (oracle/_ref/libref_dlib.so) in tests/test_oracle_dlib.py.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <vector>

#define ORC_API extern "C" __attribute__((visibility("default")))

namespace {

struct Integral {
    int rows, cols;
    std::vector<int32_t> v;
    int32_t at(long r, long c) const { return v[(size_t)r * cols + c]; }
    /* get_sum_of_area(rectangle(l,t,r,b)), integral_image.h:64-96; uint32 arithmetic = int32 wrap-around */
    int32_t sum(long l, long t, long r, long b) const
    {
        uint32_t tl = 0, tr = 0, bl = 0, br = (uint32_t)at(b, r);
        if (l - 1 >= 0 && t - 1 >= 0) { tl = at(t - 1, l - 1); bl = at(b, l - 1); tr = at(t - 1, r); }
        else if (l - 1 >= 0) bl = at(b, l - 1);
        else if (t - 1 >= 0) tr = at(t - 1, r);
        return (int32_t)(br - bl - tr + tl);
    }
    /* get_sum_of_area(centered_rect(x, y, w, h)) */
    int32_t csum(long x, long y, long w, long h) const
    {
        const long l = x - w / 2, t = y - h / 2;
        return sum(l, t, l + w - 1, t + h - 1);
    }
    int32_t haar_x(long x, long y, long width) const
    {
        const long ll = x - width / 2, lt = y - width / 2, lr = x - 1, lb = lt + width - 1;
        return (int32_t)((uint32_t)sum(x, lt, ll + width - 1, lb) - (uint32_t)sum(ll, lt, lr, lb));
    }
    int32_t haar_y(long x, long y, long width) const
    {
        const long tl = x - width / 2, tt = y - width / 2, tr = tl + width - 1, tb = y - 1;
        return (int32_t)((uint32_t)sum(tl, y, tr, tt + width - 1) - (uint32_t)sum(tl, tt, tr, tb));
    }
};

void load_integral(const unsigned char *rgb, int rows, int cols, Integral &I)
{
    I.rows = rows; I.cols = cols;
    I.v.assign((size_t)rows * cols, 0);
    for (long r = 0; r < rows; r++) {
        uint32_t temp = 0;
        for (long c = 0; c < cols; c++) {
            const unsigned char *p = rgb + 3 * ((size_t)r * cols + c);
            temp += ((unsigned)p[0] + (unsigned)p[1] + (unsigned)p[2]) / 3;
            I.v[(size_t)r * cols + c] = (int32_t)(temp + (r ? (uint32_t)I.v[(size_t)(r - 1) * cols + c] : 0u));
        }
    }
}

const int OCT = 4, INT = 6, STEP0 = 2;
long border_of(long i) { return (long)ceil((3 * (2.0 * (i + 1) + 1)) / 2.0); }
long step_of(long o) { return STEP0 * (long)(pow(2.0, (double)o) + 0.5); }

struct Pyramid {
    long nr[OCT], nc[OCT];
    std::vector<double> lev[OCT * INT];
    double val(long o, long i, long r, long c) const { return fabs(lev[o * INT + i][(size_t)r * nc[o] + c]); }
    double lap(long o, long i, long r, long c) const { return lev[o * INT + i][(size_t)r * nc[o] + c] > 0 ? +1 : -1; }
};

void build_pyramid(const Integral &I, Pyramid &P)
{
    for (long o = 0; o < OCT; o++) {
        const long step = step_of(o);
        P.nr[o] = I.rows / step; P.nc[o] = I.cols / step;
        for (long i = 0; i < INT; i++) {
            std::vector<double> &L = P.lev[o * INT + i];
            L.assign((size_t)P.nr[o] * P.nc[o], 0.0);  /* the reference leaves untouched cells uninitialised */
            const long border = border_of(i) * step;
            const long lobe = (long)(pow(2.0, o + 1.0) + 0.5) * (i + 1) + 1;
            const double area_inv = 1.0 / pow(3.0 * lobe, 2.0);
            const long off = lobe / 2 + 1;
            for (long r = border; r < I.rows - border; r += step)
                for (long c = border; c < I.cols - border; c += step) {
                    double Dxx = I.csum(c, r, lobe * 3, 2 * lobe - 1) - I.csum(c, r, lobe, 2 * lobe - 1) * 3.0;
                    double Dyy = I.csum(c, r, 2 * lobe - 1, lobe * 3) - I.csum(c, r, 2 * lobe - 1, lobe) * 3.0;
                    double Dxy = (int32_t)((uint32_t)I.csum(c - off, r + off, lobe, lobe) + (uint32_t)I.csum(c + off, r - off, lobe, lobe) -
                                           (uint32_t)I.csum(c - off, r - off, lobe, lobe) - (uint32_t)I.csum(c + off, r + off, lobe, lobe));
                    Dxx *= area_inv; Dyy *= area_inv; Dxy *= area_inv;
                    double sign = +1;
                    if (Dxx + Dyy < 0) sign = -1;
                    double det = Dxx * Dyy - 0.81 * Dxy * Dxy;
                    if (det < 0) det = 0;
                    L[(size_t)(r / step) * P.nc[o] + c / step] = sign * det;
                }
        }
    }
}

struct IP {
    double x, y, scale, score, laplacian;
    bool operator<(const IP &p) const { return score < p.score; }
};

bool is_max(const Pyramid &P, long o, long i, long r, long c)
{
    if (i <= 0 || i + 1 >= INT) return false;
    const double v = P.val(o, i, r, c);
    for (long ii = i - 1; ii <= i + 1; ii++)
        for (long rr = r - 1; rr <= r + 1; rr++)
            for (long cc = c - 1; cc <= c + 1; cc++)
                if (P.val(o, ii, rr, cc) > v) return false;
    return true;
}

IP interpolate(const Pyramid &P, long o, long i, long r, long c)
{
    const double val = P.val(o, i, r, c);
    const double g0 = (P.val(o, i, r, c + 1) - P.val(o, i, r, c - 1)) / 2.0;
    const double g1 = (P.val(o, i, r + 1, c) - P.val(o, i, r - 1, c)) / 2.0;
    const double g2 = (P.val(o, i + 1, r, c) - P.val(o, i - 1, r, c)) / 2.0;
    const double Dxx = (P.val(o, i, r, c + 1) + P.val(o, i, r, c - 1)) - 2 * val;
    const double Dyy = (P.val(o, i, r + 1, c) + P.val(o, i, r - 1, c)) - 2 * val;
    const double Dss = (P.val(o, i + 1, r, c) + P.val(o, i - 1, r, c)) - 2 * val;
    const double Dxy = (P.val(o, i, r + 1, c + 1) + P.val(o, i, r - 1, c - 1) - P.val(o, i, r - 1, c + 1) - P.val(o, i, r + 1, c - 1)) / 4.0;
    const double Dxs = (P.val(o, i + 1, r, c + 1) + P.val(o, i - 1, r, c - 1) - P.val(o, i - 1, r, c + 1) - P.val(o, i + 1, r, c - 1)) / 4.0;
    const double Dys = (P.val(o, i + 1, r + 1, c) + P.val(o, i - 1, r - 1, c) - P.val(o, i - 1, r + 1, c) - P.val(o, i + 1, r - 1, c)) / 4.0;
    /* m = [a b c; d e f; g h i] */
    const double a = Dxx, b = Dxy, cc = Dxs, d = Dxy, e = Dyy, f = Dys, g = Dxs, h = Dys, ii = Dss;
    double inv[3][3];
    double de = a * (e * ii - f * h) - b * (d * ii - f * g) + cc * (d * h - e * g);
    if (de != 0) {
        de = 1.0 / de;
        inv[0][0] = (e * ii - f * h) * de; inv[1][0] = (f * g - d * ii) * de; inv[2][0] = (d * h - e * g) * de;
        inv[0][1] = (cc * h - b * ii) * de; inv[1][1] = (a * ii - cc * g) * de; inv[2][1] = (b * g - a * h) * de;
        inv[0][2] = (b * f - cc * e) * de; inv[1][2] = (cc * d - a * f) * de; inv[2][2] = (a * e - b * d) * de;
    } else {
        for (int p = 0; p < 3; p++) for (int q = 0; q < 3; q++) inv[p][q] = p == q;
    }
    double ip[3];
    for (int p = 0; p < 3; p++) ip[p] = -(inv[p][0] * g0 + inv[p][1] * g1 + inv[p][2] * g2);
    IP t;
    const double m = std::max(fabs(ip[0]), std::max(fabs(ip[1]), fabs(ip[2])));
    if (m < 0.5) {
        const double step = (double)step_of(o);
        t.x = (c + ip[0]) * step; t.y = (r + ip[1]) * step;
        const double lobe = pow(2.0, o + 1.0) * (i + ip[2] + 1) + 1;
        t.scale = 1.2 / 9.0 * (3 * lobe);
        t.score = val;
        t.laplacian = P.lap(o, i, r, c);
    } else {
        t.x = t.y = t.scale = t.laplacian = 0;
        t.score = -1;
    }
    return t;
}

void interest_points(const Pyramid &P, double thr, std::vector<IP> &out)
{
    out.clear();
    for (long o = 0; o < OCT; o++)
        for (long i = 1; i < INT - 1; i++) {
            const long b = border_of(i + 1);
            for (long r = b + 1; r < P.nr[o] - b - 1; r++)
                for (long c = b + 1; c < P.nc[o] - b - 1; c++) {
                    if (P.val(o, i, r, c) >= thr && is_max(P, o, i, r, c)) {
                        IP sp = interpolate(P, o, i, r, c);
                        if (sp.score >= thr) out.push_back(sp);
                    }
                }
        }
}

inline long rnd(double v) { return (long)floor(v + 0.5); }

double gaussian(double x, double y, double sig)
{
    const double sqrt_2_pi = 2.5066282746310002416123552393401041626930;
    return 1.0 / (sig * sqrt_2_pi) * exp(-(x * x + y * y) / (2 * sig * sig));
}

double dominant_angle(const Integral &I, double cx, double cy, double scale)
{
    const double pi = 3.1415926535897932384626433832795;
    std::vector<double> ang, sx, sy;
    const long sc = (long)(scale + 0.5);
    for (long r = -6; r <= 6; r++)
        for (long c = -6; c <= 6; c++)
            if (r * r + c * c < 36) {
                const double gs = gaussian((double)c, (double)r, 2.5);
                const long px = rnd((double)(sc * c) + cx), py = rnd((double)(sc * r) + cy);
                const double vx = gs * I.haar_x(px, py, 4 * sc), vy = gs * I.haar_y(px, py, 4 * sc);
                sx.push_back(vx); sy.push_back(vy);
                ang.push_back(atan2(vy, vx));
            }
    double max_length = 0, best_ang = 0;
    const long slices = 45;
    const double ang_step = (2 * pi) / slices;
    for (long k = 0; k < slices; k++) {
        const double ang1 = ang_step * k - pi, ang2 = ang1 + pi / 3;
        double vx = 0, vy = 0;
        for (size_t i = 0; i < ang.size(); i++) {
            if (ang1 <= ang[i] && ang[i] <= ang2) { vx += sx[i]; vy += sy[i]; }
            else if (ang2 > pi && (ang[i] >= ang1 || ang[i] <= (-2 * pi + ang2))) { vx += sx[i]; vy += sy[i]; }
        }
        if (vx * vx + vy * vy > max_length) { max_length = vx * vx + vy * vy; best_ang = atan2(vy, vx); }
    }
    return best_ang;
}

void descriptor(const Integral &I, double cx, double cy, double scale, double angle, double *des)
{
    const double sa = sin(angle), ca = cos(angle), isa = sin(-angle), ica = cos(-angle);
    const long sc = (long)(scale + 0.5);
    long count = 0;
    for (long r = -10; r < 10; r += 5)
        for (long c = -10; c < 10; c += 5) {
            double vx = 0, vy = 0, ax = 0, ay = 0;
            for (long y = r - 1; y < r + 5 + 1; y++) {
                if (y < -10 || y >= 10) continue;
                for (long x = c - 1; x < c + 5 + 1; x++) {
                    if (x < -10 || x >= 10) continue;
                    const double qx = x * scale, qy = y * scale;
                    const long px = rnd((ca * qx - sa * qy) + cx), py = rnd((sa * qx + ca * qy) + cy);
                    const double weight = 1.0 / (4 + labs(r + 2 - y) + labs(c + 2 - x));
                    const double tx = weight * I.haar_x(px, py, 2 * sc), ty = weight * I.haar_y(px, py, 2 * sc);
                    const double rx = ica * tx - isa * ty, ry = isa * tx + ica * ty;
                    vx += rx; vy += ry; ax += fabs(rx); ay += fabs(ry);
                }
            }
            des[count++] = vx; des[count++] = vy; des[count++] = ax; des[count++] = ay;
        }
    double s = 0;
    for (int i = 0; i < 64; i++) s += des[i] * des[i];
    const double inv_len = 1.0 / (sqrt(s) + 1e-7);
    for (int i = 0; i < 64; i++) des[i] = des[i] * inv_len;
}

}  // namespace

ORC_API int orc_surf_integral(const unsigned char *rgb, int rows, int cols, int32_t *out)
{
    Integral I;
    load_integral(rgb, rows, cols, I);
    memcpy(out, I.v.data(), sizeof(int32_t) * I.v.size());
    return 0;
}

/* level (o,i): signed determinant values; untouched cells are 0 here */
ORC_API int orc_surf_pyramid_level(const unsigned char *rgb, int rows, int cols, int o, int i, double *out, int *nr, int *nc, int *border)
{
    Integral I; Pyramid P;
    load_integral(rgb, rows, cols, I);
    build_pyramid(I, P);
    *nr = (int)P.nr[o]; *nc = (int)P.nc[o]; *border = (int)border_of(i);
    if (out) memcpy(out, P.lev[o * INT + i].data(), sizeof(double) * P.lev[o * INT + i].size());
    return 0;
}

/* records of 6 doubles (x, y, scale, score, laplacian, 0) in emission order (octave, interval, row, column) */
ORC_API long orc_surf_interest_points(const unsigned char *rgb, int rows, int cols, double thr, double *out, long cap)
{
    Integral I; Pyramid P; std::vector<IP> pts;
    load_integral(rgb, rows, cols, I);
    build_pyramid(I, P);
    interest_points(P, thr, pts);
    for (size_t k = 0; k < pts.size() && (long)k < cap; k++) {
        double *q = out + 6 * k;
        q[0] = pts[k].x; q[1] = pts[k].y; q[2] = pts[k].scale; q[3] = pts[k].score; q[4] = pts[k].laplacian; q[5] = 0;
    }
    return (long)pts.size();
}

/* records of 71 doubles (x, y, angle, scale, score, laplacian, 0, des[64]) */
ORC_API long orc_surf(const unsigned char *rgb, int rows, int cols, long max_points, double thr, double *out, long cap)
{
    Integral I; Pyramid P; std::vector<IP> pts;
    load_integral(rgb, rows, cols, I);
    build_pyramid(I, P);
    interest_points(P, thr, pts);
    std::sort(pts.rbegin(), pts.rend());
    long n = 0;
    for (size_t k = 0; k < std::min((size_t)max_points, pts.size()); k++) {
        const unsigned long bs = (unsigned long)(32.0 * pts[k].scale);
        const long px = rnd(pts[k].x), py = rnd(pts[k].y);
        const long l = px - (long)bs / 2, t = py - (long)bs / 2, r = l + (long)bs - 1, b = t + (long)bs - 1;
        if (!(l >= 0 && t >= 0 && r <= cols - 1 && b <= rows - 1)) continue;  /* get_rect(int_img).contains(rect) */
        if (n < cap) {
            double *q = out + 71 * n;
            const double ang = dominant_angle(I, pts[k].x, pts[k].y, pts[k].scale);
            q[0] = pts[k].x; q[1] = pts[k].y; q[2] = ang; q[3] = pts[k].scale; q[4] = pts[k].score; q[5] = pts[k].laplacian; q[6] = 0;
            descriptor(I, pts[k].x, pts[k].y, pts[k].scale, ang, q + 7);
        }
        n++;
    }
    return n;
}
