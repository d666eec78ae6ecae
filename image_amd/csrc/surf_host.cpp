// image_amd/csrc/surf_host.cpp -- host stage of imgfd_surf (K19): dominant orientation and the 64-d SURF descriptor
// of one interest point, from the int32 integral image the device computed.
//
// Mirrors compute_dominant_angle / compute_surf_descriptor, image.dlib/inst/dlib-19.20/dlib/image_keypoint/surf.h:75-232,
// with haar_x / haar_y of dlib/image_transforms/integral_image.h:124-183, the float->integer point rounding
// floor(v+0.5) of dlib/geometry/vector.h:138-149 and point_rotator (dlib/geometry/point_transforms.h:22-49).
// Per-point work on <= max_points points (R default 1000) built on libm's atan2/exp/sin/cos: it runs where the
// reference runs it, on the host's glibc, so orientation-bucket membership tests see the same bits.
#include <math.h>
#include <stdint.h>

#include <vector>

namespace {

struct Table {
    const int32_t *v;
    int rows, cols;
    uint32_t at(long r, long c) const { return (uint32_t)v[(size_t)r * cols + c]; }
    // get_sum_of_area(rectangle(l,t,r,b)); unsigned arithmetic reproduces the wrapping int32 of the reference
    uint32_t box(long l, long t, long r, long b) const
    {
        uint32_t tl = 0, tr = 0, bl = 0;
        const uint32_t br = at(b, r);
        if (l >= 1 && t >= 1) { tl = at(t - 1, l - 1); bl = at(b, l - 1); tr = at(t - 1, r); }
        else if (l >= 1) bl = at(b, l - 1);
        else if (t >= 1) tr = at(t - 1, r);
        return br - bl - tr + tl;
    }
    // right half minus left half of the width x width box whose top-left is (x - width/2, y - width/2)
    int32_t haar_x(long x, long y, long width) const
    {
        const long left = x - width / 2, top = y - width / 2, bottom = top + width - 1;
        return (int32_t)(box(x, top, left + width - 1, bottom) - box(left, top, x - 1, bottom));
    }
    // bottom half minus top half
    int32_t haar_y(long x, long y, long width) const
    {
        const long left = x - width / 2, top = y - width / 2, right = left + width - 1;
        return (int32_t)(box(left, y, right, top + width - 1) - box(left, top, right, y - 1));
    }
};

inline long to_long(double v) { return (long)floor(v + 0.5); }

double gauss2(double x, double y, double sig)
{
    const double sqrt_2_pi = 2.5066282746310002416123552393401041626930;
    return 1.0 / (sig * sqrt_2_pi) * exp(-(x * x + y * y) / (2 * sig * sig));
}

}  // namespace

void surf_describe_host(const int32_t *I, int rows, int cols, double x, double y, double scale, double *angle, double *des)
{
    const Table T{I, rows, cols};
    const double pi = 3.1415926535897932384626433832795;
    const long sc = (long)(scale + 0.5);

    // ---- dominant angle: 109 Gaussian-weighted Haar samples on a radius-6 disc, 45 sliding pi/3 windows
    double sx[169], sy[169], sa[169];
    int ns = 0;
    for (long r = -6; r <= 6; r++)
        for (long c = -6; c <= 6; c++) {
            if (r * r + c * c >= 36) continue;
            const double w = gauss2((double)c, (double)r, 2.5);
            const long px = to_long((double)(sc * c) + x), py = to_long((double)(sc * r) + y);
            sx[ns] = w * T.haar_x(px, py, 4 * sc);
            sy[ns] = w * T.haar_y(px, py, 4 * sc);
            sa[ns] = atan2(sy[ns], sx[ns]);
            ns++;
        }
    double best_len = 0, best_ang = 0;
    const double ang_step = (2 * pi) / 45;
    for (long k = 0; k < 45; k++) {
        const double a1 = ang_step * k - pi, a2 = a1 + pi / 3;
        double vx = 0, vy = 0;
        for (int i = 0; i < ns; i++) {
            const bool in = (a1 <= sa[i] && sa[i] <= a2) || (a2 > pi && (sa[i] >= a1 || sa[i] <= (-2 * pi + a2)));
            if (in) { vx += sx[i]; vy += sy[i]; }
        }
        const double len = vx * vx + vy * vy;
        if (len > best_len) { best_len = len; best_ang = atan2(vy, vx); }
    }
    *angle = best_ang;

    // ---- descriptor: 4x4 buckets of 5x5 samples (+1 sample of padding), rotated by the dominant angle
    const double sn = sin(best_ang), cs = cos(best_ang), isn = sin(-best_ang), ics = cos(-best_ang);
    int k = 0;
    for (long r = -10; r < 10; r += 5)
        for (long c = -10; c < 10; c += 5) {
            double vx = 0, vy = 0, ax = 0, ay = 0;
            for (long yy = r - 1; yy < r + 6; yy++) {
                if (yy < -10 || yy >= 10) continue;
                for (long xx = c - 1; xx < c + 6; xx++) {
                    if (xx < -10 || xx >= 10) continue;
                    const double qx = xx * scale, qy = yy * scale;
                    const long px = to_long((cs * qx - sn * qy) + x), py = to_long((sn * qx + cs * qy) + y);
                    const double weight = 1.0 / (4 + labs(r + 2 - yy) + labs(c + 2 - xx));
                    const double hx = weight * T.haar_x(px, py, 2 * sc), hy = weight * T.haar_y(px, py, 2 * sc);
                    const double rx = ics * hx - isn * hy, ry = isn * hx + ics * hy;
                    vx += rx; vy += ry; ax += fabs(rx); ay += fabs(ry);
                }
            }
            des[k++] = vx; des[k++] = vy; des[k++] = ax; des[k++] = ay;
        }
    double ss = 0;
    for (int i = 0; i < 64; i++) ss += des[i] * des[i];
    const double inv_len = 1.0 / (sqrt(ss) + 1e-7);
    for (int i = 0; i < 64; i++) des[i] = des[i] * inv_len;
}
