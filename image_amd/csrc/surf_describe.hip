// image_amd/csrc/surf_describe.hip -- K19 on the device: dominant orientation and 64-d SURF descriptor per interest point.
//
// Replaces compute_dominant_angle / compute_surf_descriptor, image.dlib/inst/dlib-19.20/dlib/image_keypoint/surf.h:
// 75-232, with haar_x / haar_y of dlib/image_transforms/integral_image.h:124-183, the float->integer point rounding
// floor(v+0.5) of dlib/geometry/vector.h:138-149 and point_rotator (dlib/geometry/point_transforms.h:22-49).
//
// One workgroup per point, two kernels:
//   surf_orient  109 Gaussian-weighted Haar samples on the radius-6 disc (:88-105), one lane each; then either
//                  (assisted) the weighted responses (sx, sy) are written out and the HOST forms atan2 / the 45 sliding
//                             windows / sin / cos with glibc -- the bits the reference gets -- and sends 5 doubles back;
//                  (device)   lanes 0..44 own one pi/3 window each and add the samples in the reference's order
//                             (:111-137), lane 0 picks the longest; atan2/sin/cos come from the device libm.
//   surf_desc    the 20x20 grid of rotated Haar samples is evaluated once (the reference re-samples the padding ring
//                of neighbouring buckets), then 16 buckets x 4 sums are accumulated by one lane each in the
//                reference's sample order (:163-212), then the length normalisation (:216-219, sequential sum).
// Everything but atan2/sin/cos is integer arithmetic or single IEEE double operations in the reference's order
// (library built -ffp-contract=off), so the assisted mode reproduces the reference bit for bit, and the device mode
// differs only through the last bits of the device libm's trigonometry (tests: <= 1e-9 on descriptors).
#include "common.h"
#include "surf_describe.h"

#include <math.h>

namespace {

struct Haar {
    SurfTable T;  // the integral image, plain or by column residue
    int rows, cols;
    __device__ __forceinline__ unsigned at(long r, long c) const
    {
        // valid points never leave the image (surf.h:271-285 drops those whose 32*scale box does); the clamp only keeps
        // a corrupt record from reading outside the table
        r = r < 0 ? 0 : (r >= rows ? rows - 1 : r);
        c = c < 0 ? 0 : (c >= cols ? cols - 1 : c);
        return T.at((int)r, (int)c);
    }
    // get_sum_of_area(rectangle(l,t,r,b)), integral_image.h:64-96; unsigned arithmetic = the reference's wrapping int32
    __device__ __forceinline__ unsigned box(long l, long t, long r, long b) const
    {
        unsigned tl = 0, tr = 0, bl = 0;
        const unsigned br = at(b, r);
        if (l >= 1 && t >= 1) { tl = at(t - 1, l - 1); bl = at(b, l - 1); tr = at(t - 1, r); }
        else if (l >= 1) bl = at(b, l - 1);
        else if (t >= 1) tr = at(t - 1, r);
        return br - bl - tr + tl;
    }
    __device__ __forceinline__ int haar_x(long x, long y, long width) const  // :124-152
    {
        const long left = x - width / 2, top = y - width / 2, bottom = top + width - 1;
        return (int)(box(x, top, left + width - 1, bottom) - box(left, top, x - 1, bottom));
    }
    __device__ __forceinline__ int haar_y(long x, long y, long width) const  // :154-183
    {
        const long left = x - width / 2, top = y - width / 2, right = left + width - 1;
        return (int)(box(left, y, right, top + width - 1) - box(left, top, right, y - 1));
    }
    // haar_x and haar_y of one sample together.  Their four boxes tile one w x w square cut in two along x and along y: the
    // sixteen corners of :124-183 are eight distinct table entries -- rows top-1, y-1, bottom x columns left-1, x-1, right,
    // without the centre.  With the square inside the image (every sample of a valid point: surf.h:271-285 keeps a point only
    // if its 32*scale box is) no corner is clamped or zeroed, so eight independent loads in ONE round trip give both
    // responses -- integer sums, the reference's values exactly; the general case-by-case form above was two dependent round
    // trips per box and twice the table traffic.  A square that touches the border takes that form.
    __device__ __forceinline__ void haar_xy(long x, long y, long width, int *hx, int *hy) const
    {
        const long left = x - width / 2, top = y - width / 2, right = left + width - 1, bottom = top + width - 1;
        if (left >= 1 && top >= 1 && right < cols && bottom < rows) {
            const int rt = (int)top - 1, rm = (int)y - 1, rb = (int)bottom, cl = (int)left - 1, cm = (int)x - 1, cr = (int)right;
            const unsigned tl = T.at(rt, cl), tm = T.at(rt, cm), tr = T.at(rt, cr);
            const unsigned ml = T.at(rm, cl), mr = T.at(rm, cr);
            const unsigned bl = T.at(rb, cl), bm = T.at(rb, cm), br = T.at(rb, cr);
            *hx = (int)((br - bm - tr + tm) - (bm - bl - tm + tl));  // box(x, top, right, bottom) - box(left, top, x - 1, bottom)
            *hy = (int)((br - bl - mr + ml) - (mr - ml - tr + tl));  // box(left, y, right, bottom) - box(left, top, right, y - 1)
        } else {
            *hx = haar_x(x, y, width);
            *hy = haar_y(x, y, width);
        }
    }
};

__device__ __forceinline__ long surf_to_long(double v) { return (long)floor(v + 0.5); }

}  // namespace

// pts: m x 3 (x, y, scale).  samples != nullptr: write sx[109], sy[109] per point (assisted mode).
// trig != nullptr: write angle, sin, cos, sin(-), cos(-) per point (device mode).
__global__ void __launch_bounds__(128) surf_orient(SurfTable I,
                                                   const double *__restrict__ pts, SurfOrientTable T,
                                                   double *__restrict__ samples, double *__restrict__ trig,
                                                   const unsigned *__restrict__ m_dev, SurfGroup grp)
{
    {   // blockIdx.y = tile of the group (imgfd_surf_dev: the K19 kernels of a group of tiles are one launch)
        const size_t t = blockIdx.y;
        I.p += t * grp.table; pts += t * grp.pts;
        if (trig) trig += t * grp.trig;
        if (m_dev) m_dev += t;
    }
    if (m_dev && blockIdx.x >= *m_dev) return;  // the grid covers the upper bound; the number of points lives on the device
    __shared__ double sx[SURF_NSAMP], sy[SURF_NSAMP];
    __shared__ double wx[45], wy[45];
    __shared__ unsigned long long in_windows[SURF_NSAMP];  // bit k: the sample's angle lies in window k
    const size_t p = blockIdx.x;
    const int i = threadIdx.x;
    const double x = pts[3 * p], y = pts[3 * p + 1], scale = pts[3 * p + 2];
    const long sc = (long)(scale + 0.5);
    const Haar H{I, I.rows, I.cols};
    if (i < SURF_NSAMP) {
        const long r = T.r[i], c = T.c[i];
        const long px = surf_to_long((double)(sc * c) + x), py = surf_to_long((double)(sc * r) + y);
        int hx, hy;
        H.haar_xy(px, py, 4 * sc, &hx, &hy);
        const double vx = T.w[i] * hx;
        const double vy = T.w[i] * hy;
        if (samples) {
            samples[p * (2 * SURF_NSAMP) + i] = vx;
            samples[p * (2 * SURF_NSAMP) + SURF_NSAMP + i] = vy;
        }
        sx[i] = vx;
        sy[i] = vy;
        if (trig) {
            // which of the 45 windows hold this sample's angle (:113-127): the tests of all windows here, one sample per lane, so
            // that a window's lane -- which has to add its samples in the reference's order -- only looks a bit up per sample
            const double pi = 3.1415926535897932384626433832795;
            const double ang_step = (2 * pi) / 45;
            const double a = atan2(vy, vx);
            unsigned long long bits = 0;
            for (int k = 0; k < 45; k++) {
                const double a1 = ang_step * k - pi, a2 = a1 + pi / 3;
                const bool in = (a1 <= a && a <= a2) || (a2 > pi && (a >= a1 || a <= (-2 * pi + a2)));
                bits |= (unsigned long long)in << k;
            }
            in_windows[i] = bits;
        }
    }
    if (!trig) return;
    __syncthreads();
    if (i < 45) {  // :111-131
        // A sample outside the window adds +0.0 instead of being skipped: a sum that starts at +0.0 never becomes -0.0 (x + y is
        // -0.0 only for two negative zeros), and x + 0.0 = x for every other x, so the bits are those of the reference's skipping
        // loop -- and the 109 steps are one chain of dependent f64 adds with every LDS read issued ahead of it (a branch per sample
        // put two dependent LDS round trips into every step: ~10 of the kernel's 24-31 us, round 5)
        double vx = 0, vy = 0;
#pragma unroll 8
        for (int s = 0; s < SURF_NSAMP; s++) {
            const bool in = (in_windows[s] >> i) & 1ull;
            vx += in ? sx[s] : 0.0;
            vy += in ? sy[s] : 0.0;
        }
        wx[i] = vx;
        wy[i] = vy;
    }
    __syncthreads();
    if (i == 0) {  // :132-137: first strictly longest window (its angle is formed once, behind the search: the reference's value)
        double best_len = 0;
        int best = -1;
        for (int k = 0; k < 45; k++) {
            const double len = wx[k] * wx[k] + wy[k] * wy[k];
            if (len > best_len) { best_len = len; best = k; }
        }
        const double best_ang = best >= 0 ? atan2(wy[best], wx[best]) : 0.0;
        double *t = trig + 5 * p;
        t[0] = best_ang;
        t[1] = sin(best_ang);
        t[2] = cos(best_ang);
        t[3] = sin(-best_ang);
        t[4] = cos(-best_ang);
    }
}

// trig: m x 5 (angle, sin, cos, sin(-angle), cos(-angle)); point p's descriptor goes to des[p*des_stride .. +64) and, if
// angle_out, its angle to angle_out[p*des_stride]
#ifndef SURF_DESC_NT
#define SURF_DESC_NT 64  // threads per point of surf_desc
#endif
__global__ void __launch_bounds__(SURF_DESC_NT) surf_desc(SurfTable I,
                                                const double *__restrict__ pts, const double *__restrict__ trig,
                                                double *__restrict__ des, int des_stride, double *__restrict__ angle_out,
                                                const unsigned *__restrict__ m_dev, SurfGroup grp)
{
    {
        const size_t t = blockIdx.y;
        I.p += t * grp.table; pts += t * grp.pts; trig += t * grp.trig; des += t * grp.des;
        if (angle_out) angle_out += t * grp.des;
        if (m_dev) m_dev += t;
    }
    if (m_dev && blockIdx.x >= *m_dev) return;
    __shared__ int hx[400], hy[400];
    __shared__ double rx[16 * 49], ry[16 * 49];
    __shared__ double d[64];
    __shared__ double inv_len_s;
    const size_t p = blockIdx.x;
    const int lane = threadIdx.x;
    const double x = pts[3 * p], y = pts[3 * p + 1], scale = pts[3 * p + 2];
    const double sn = trig[5 * p + 1], cs = trig[5 * p + 2], isn = trig[5 * p + 3], ics = trig[5 * p + 4];
    const long sc = (long)(scale + 0.5);
    const Haar H{I, I.rows, I.cols};
    // the 20 x 20 sample grid (:176-186)
    for (int s = lane; s < 400; s += SURF_DESC_NT) {
        const long yy = s / 20 - 10, xx = s % 20 - 10;
        const double qx = xx * scale, qy = yy * scale;
        const long px = surf_to_long((cs * qx - sn * qy) + x), py = surf_to_long((sn * qx + cs * qy) + y);
        H.haar_xy(px, py, 2 * sc, &hx[s], &hy[s]);
    }
    __syncthreads();
    // weighted, rotated back (:188-199), per bucket slot j = (yy - (r-1))*7 + (xx - (c-1))
    for (int slot = lane; slot < 16 * 49; slot += SURF_DESC_NT) {
        const int bucket = slot / 49, j = slot % 49;
        const long r = -10 + 5 * (bucket >> 2), c = -10 + 5 * (bucket & 3);
        const long yy = r - 1 + j / 7, xx = c - 1 + j % 7;
        double vx = 0, vy = 0;
        if (yy >= -10 && yy < 10 && xx >= -10 && xx < 10) {
            const int s = (int)((yy + 10) * 20 + (xx + 10));
            const double weight = 1.0 / (double)(4 + labs(r + 2 - yy) + labs(c + 2 - xx));
            const double wxh = weight * hx[s], wyh = weight * hy[s];
            vx = ics * wxh - isn * wyh;
            vy = isn * wxh + ics * wyh;
        }
        rx[slot] = vx;
        ry[slot] = vy;
    }
    __syncthreads();
    if (lane < 64) {   // lane = bucket*4 + {vx, vy, |vx|, |vy|} (:201-212), samples in the reference's order
        const int bucket = lane >> 2, comp = lane & 3;
        const long r = -10 + 5 * (bucket >> 2), c = -10 + 5 * (bucket & 3);
        const double *src = (comp & 1) ? ry + bucket * 49 : rx + bucket * 49;
        // slots outside the 20 x 20 grid hold +0.0 (written above): adding them leaves the sum's bits alone (a sum that starts at
        // +0.0 never becomes -0.0), so the 49 steps need no test and the LDS reads run ahead of the chain of adds
        (void)r; (void)c;
        double acc = 0;
#pragma unroll 7
        for (int j = 0; j < 49; j++) {
            const double v = src[j];
            acc += (comp & 2) ? fabs(v) : v;
        }
        d[lane] = acc;
    }
    __syncthreads();
    if (lane == 0) {  // :216-219
        double ss = 0;
        for (int k = 0; k < 64; k++) ss += d[k] * d[k];
        inv_len_s = 1.0 / (sqrt(ss) + 1e-7);
    }
    __syncthreads();
    if (lane < 64) des[p * des_stride + lane] = d[lane] * inv_len_s;
    if (angle_out && lane == 0) angle_out[p * des_stride] = trig[5 * p];
}

// host side ---------------------------------------------------------------------------------------------------------
void surf_orient_table(SurfOrientTable *T)
{
    int n = 0;
    for (long r = -6; r <= 6; r++)
        for (long c = -6; c <= 6; c++) {
            if (r * r + c * c >= 36) continue;
            T->w[n] = surf_gauss_weight((double)c, (double)r);  // glibc exp, as the reference evaluates it
            T->r[n] = (signed char)r;
            T->c[n] = (signed char)c;
            n++;
        }
}

imgfd_status launch_surf_orient(imgfd_ctx *ctx, const SurfTable &I, const double *d_pts, int m,
                                double *d_samples, double *d_trig, const unsigned *m_dev, const SurfGroup *grp)
{
    if (m < 1) return IMGFD_OK;
    SurfOrientTable T;
    surf_orient_table(&T);
    const SurfGroup one{1, 0, 0, 0, 0};
    const SurfGroup &G = grp ? *grp : one;
    hipLaunchKernelGGL(surf_orient, dim3(m, G.tiles), dim3(128), 0, ctx->stream, I, d_pts, T, d_samples, d_trig, m_dev, G);
    IMGFD_HIP(ctx, hipGetLastError());
    return IMGFD_OK;
}

imgfd_status launch_surf_desc(imgfd_ctx *ctx, const SurfTable &I, const double *d_pts,
                              const double *d_trig, int m, double *d_des, int des_stride, double *d_angle, const unsigned *m_dev,
                              const SurfGroup *grp)
{
    if (m < 1) return IMGFD_OK;
    const SurfGroup one{1, 0, 0, 0, 0};
    const SurfGroup &G = grp ? *grp : one;
    hipLaunchKernelGGL(surf_desc, dim3(m, G.tiles), dim3(SURF_DESC_NT), 0, ctx->stream, I, d_pts, d_trig, d_des, des_stride, d_angle, m_dev, G);
    IMGFD_HIP(ctx, hipGetLastError());
    return IMGFD_OK;
}
