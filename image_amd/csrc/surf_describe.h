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

// The int32 integral image (integral_image.h:33-62) as the device kernels hold it, in ONE of two layouts:
//   per == 0         plain:      word of (row, x) = row * cols + x
//   per == cols / 4  by residue: word of (row, x) = row * cols + (x & 3) * per + (x >> 2)
// The residue layout is what the gather kernels of octaves 1-3 want (their level pixels sit on columns that are multiples of
// 4, 8, 16: consecutive lanes read consecutive words of one residue plane); since round 6 it is the ONLY copy of the table
// whenever the image allows it (cols a multiple of 16, band / strip scans), and every reader addresses it through this.
// a group of tiles handled by ONE launch of a back-stage kernel (blockIdx.y = tile): distances between consecutive tiles'
// buffers, in elements (all tiles of a call share one geometry, so their buffer sets are carved alike)
struct SurfGroup {
    int tiles;
    size_t table, pts, trig, des;
};

struct SurfTable {
    const unsigned *p;
    int rows, cols, per;
#ifdef __device__  /* (surf_host.cpp includes this header without the HIP runtime) */
    __device__ __forceinline__ size_t word(int r, int x) const { return (size_t)r * cols + (per ? (x & 3) * per + (x >> 2) : x); }
    __device__ __forceinline__ unsigned at(int r, int x) const { return p[word(r, x)]; }
#endif
};
#endif
