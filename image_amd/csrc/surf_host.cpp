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

void surf_orient_host(const double *sx, const double *sy, double *out5)
{
    const double pi = 3.1415926535897932384626433832795;
    double sa[SURF_NSAMP];
    for (int i = 0; i < SURF_NSAMP; i++) sa[i] = atan2(sy[i], sx[i]);
    double best_len = 0, best_ang = 0;
    const double ang_step = (2 * pi) / 45;
    for (long k = 0; k < 45; k++) {  // :111-137
        const double a1 = ang_step * k - pi, a2 = a1 + pi / 3;
        double vx = 0, vy = 0;
        for (int i = 0; i < SURF_NSAMP; i++) {
            const bool in = (a1 <= sa[i] && sa[i] <= a2) || (a2 > pi && (sa[i] >= a1 || sa[i] <= (-2 * pi + a2)));
            if (in) { vx += sx[i]; vy += sy[i]; }
        }
        const double len = vx * vx + vy * vy;
        if (len > best_len) { best_len = len; best_ang = atan2(vy, vx); }
    }
    out5[0] = best_ang;
    out5[1] = sin(best_ang);   // point_rotator(angle), point_transforms.h:31-35
    out5[2] = cos(best_ang);
    out5[3] = sin(-best_ang);  // point_rotator(-angle) for rotating the responses back, surf.h:160
    out5[4] = cos(-best_ang);
}
