// image_amd/csrc/surf_host.cpp -- the libm half of K19 (SURF orientation), kept on the host.
//
// compute_dominant_angle, image.dlib/inst/dlib-19.20/dlib/image_keypoint/surf.h:75-139, spends its time in 109 Haar
// samples per point (done on the device, surf_describe.hip) and decides with atan2 which of 45 sliding pi/3 windows
// each sample falls into.  Window membership compares atan2 results against fixed thresholds and the descriptor's
// sampling grid is rotated by sin/cos of the winner, so the LAST BIT of these libm calls can move a sample by a
// pixel.  imgfd_surf therefore evaluates exactly these calls where the reference does -- on the host's glibc, built
// the way R CMD INSTALL builds the reference (g++ -O2, no contraction) -- from the 218 doubles per point the device
// hands back, and returns five doubles per point.  (imgfd_surf_dev runs the same steps with the device libm.)
#include "surf_describe.h"

#include <math.h>

// gaussian(x, y, sig) of surf.h:41-47 with sig = 2.5
double surf_gauss_weight(double x, double y)
{
    const double sig = 2.5;
    const double sqrt_2_pi = 2.5066282746310002416123552393401041626930;
    return 1.0 / (sig * sqrt_2_pi) * exp(-(x * x + y * y) / (2 * sig * sig));
}

// The 45 sliding windows of :111-131: window k takes the samples whose angle a satisfies
//     (a1 <= a && a <= a2) || (a2 > pi && (a >= a1 || a <= -2 pi + a2)),      a1 = ang_step k - pi,  a2 = a1 + pi / 3.
// a1, a2 and -2 pi + a2 grow with k, so each of the three comparisons is true for a prefix or a suffix of the windows: the boundary
// windows are found by stepping from an estimate with THE SAME comparisons on the same doubles (the reference's expressions,
// evaluated once), and a sample is added to exactly the windows the reference's 45 x 109 tests would add it to, in sample order
// per window -- 109 x ~8 additions instead of 4 905 tests per point.
namespace {
struct Windows {
    double a1[45], a2[45], lim[45];
    bool wraps[45];
    Windows()
    {
        const double pi = 3.1415926535897932384626433832795;
        const double ang_step = (2 * pi) / 45;
        for (long k = 0; k < 45; k++) {
            a1[k] = ang_step * k - pi;
            a2[k] = a1[k] + pi / 3;
            lim[k] = -2 * pi + a2[k];
            wraps[k] = a2[k] > pi;
        }
    }
};
const Windows W;
}  // namespace

void surf_orient_host(const double *sx, const double *sy, double *out5)
{
    double wx[45], wy[45];
    for (int k = 0; k < 45; k++) wx[k] = wy[k] = 0;
    const double pi = 3.1415926535897932384626433832795, ang_step = (2 * pi) / 45;
    for (int i = 0; i < SURF_NSAMP; i++) {
        const double a = atan2(sy[i], sx[i]);
        // hi = the last window with a1 <= a (-1: none), lo = the first with a <= a2 (45: none), kw = the first with a <= lim
        int hi = a == a ? (int)((a + pi) / ang_step) : 0;  // an estimate; the loops below decide (a NaN angle: no window, as in the reference)
        if (hi > 44) hi = 44;
        if (hi < 0) hi = 0;
        while (hi < 44 && W.a1[hi + 1] <= a) hi++;
        while (hi >= 0 && !(W.a1[hi] <= a)) hi--;
        int lo = hi - 8 < 0 ? 0 : hi - 8;
        while (lo > 0 && a <= W.a2[lo - 1]) lo--;
        while (lo < 45 && !(a <= W.a2[lo])) lo++;
        int kw = 38;  // only the last windows wrap (a2 > pi): lim is compared there alone
        while (kw > 0 && a <= W.lim[kw - 1]) kw--;
        while (kw < 45 && !(a <= W.lim[kw])) kw++;
        for (int k = lo; k <= hi; k++) { wx[k] += sx[i]; wy[k] += sy[i]; }   // a1 <= a <= a2
        for (int k = 37; k < 45; k++) {                                       // the windows that wrap and did not take it above
            if (!W.wraps[k] || (k >= lo && k <= hi)) continue;
            if (k <= hi || k >= kw) { wx[k] += sx[i]; wy[k] += sy[i]; }       // a >= a1 || a <= -2 pi + a2
        }
    }
    double best_len = 0, best_ang = 0;
    for (long k = 0; k < 45; k++) {  // :132-137
        const double len = wx[k] * wx[k] + wy[k] * wy[k];
        if (len > best_len) { best_len = len; best_ang = atan2(wy[k], wx[k]); }
    }
    out5[0] = best_ang;
    out5[1] = sin(best_ang);   // point_rotator(angle), point_transforms.h:31-35
    out5[2] = cos(best_ang);
    out5[3] = sin(-best_ang);  // point_rotator(-angle) for rotating the responses back, surf.h:160
    out5[4] = cos(-best_ang);
}
