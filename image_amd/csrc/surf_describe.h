// image_amd/csrc/surf_describe.h -- K19 plumbing shared by surf.hip, surf_describe.hip and surf_host.cpp.
#pragma once
#include <stddef.h>

#define SURF_NSAMP 109  // lattice points of the radius-6 disc, surf.h:88-93

struct SurfOrientTable {
    double w[SURF_NSAMP];        // gaussian(c, r, 2.5), surf.h:94
    signed char r[SURF_NSAMP], c[SURF_NSAMP];
};

// surf_host.cpp (g++ -O2, glibc): the libm half of K19
double surf_gauss_weight(double x, double y);
// sx, sy: the 109 weighted Haar responses of one point; out: angle, sin, cos, sin(-angle), cos(-angle)
void surf_orient_host(const double *sx, const double *sy, double *out5);

#ifdef IMGFD_BUILD
struct imgfd_ctx;
void surf_orient_table(SurfOrientTable *T);
#endif
