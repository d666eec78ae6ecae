// image_amd/csrc/harris_subpixel.hip -- sub-pixel refinement of the Harris corners, on the device.
//
// Replaces compute_subpixel_precision(), image.CornerDetectionHarris/src/harris.cpp:340-381, and the two fits it calls
// (interpolation.cpp:27-54 quadratic approximation, :62-212 Newton iteration on the bi-quadratic interpolant): one thread
// per corner of the compacted list reads the 3x3 neighbourhood of R straight from the response plane in HBM (nothing
// about the neighbourhoods crosses PCIe), fits, and writes the refined record beside the integer one.  A fit the
// reference rejects (singular system, a Newton step that leaves the pixel) leaves the record as it was.
//
// The reference mixes float variables with double-promoted literals; every rounding below sits where its x86-64 build
// rounds (the library is built -ffp-contract=off, float division and comparisons are IEEE): the goldens
// tests/golden/harris_rcpp_default*.npz and harris_quartic*.npz are reproduced bit for bit.
#include "common.h"

namespace {

// the response around a corner, by compass point (y grows downwards: n = row y-1)
struct Patch {
    float nw, n, ne, w, c, e, sw, s, se;
};

__device__ __forceinline__ Patch patch_at(const float *__restrict__ R, int nx, int x, int y)
{
    const float *p = R + (size_t)y * nx + x;
    return Patch{p[-nx - 1], p[-nx], p[-nx + 1], p[-1], p[0], p[1], p[nx - 1], p[nx], p[nx + 1]};
}

// halves and quarters are taken in double and rounded once, as `0.5 * (float expression)` does in the reference
__device__ __forceinline__ float half_of(float v) { return (float)(0.5 * (double)v); }
__device__ __forceinline__ float quarter_of(float v) { return (float)(0.25 * (double)v); }

// Second-order Taylor model around the pixel: gradient g, Hessian h from central differences; the stationary point of
// the model is the refined position, the model's value there the refined strength (interpolation.cpp:27-54).
__device__ __forceinline__ bool taylor_fit(const Patch &m, float &x, float &y, float &strength)
{
    const float gx = half_of(m.e - m.w);
    const float gy = half_of(m.s - m.n);
    const float hxx = m.e - 2 * m.c + m.w;
    const float hyy = m.s - 2 * m.c + m.n;
    const float hxy = quarter_of(m.nw - m.ne - m.sw + m.se);
    const float det = hxx * hyy - hxy * hxy;
    if ((double)(det * det) < 1E-6) return false;  // the reference's invertibility test
    const float ux = (hyy * gx - hxy * gy) / det;
    const float uy = (hxx * gy - hxy * gx) / det;
    x -= ux;
    y -= uy;
    const float linear = m.c + gx * ux + gy * uy;
    const float quad = hxx * ux * ux + 2 * ux * uy * hxy + hyy * uy * uy;
    strength = (float)((double)linear + 0.5 * (double)quad);
    return true;
}

// The bi-quadratic surface through the nine samples, f(u, v) = sum q[i][j] u^i v^j with i, j in 0..2 (u to the right, v
// downwards, the pixel at the origin), its gradient and Hessian; terms are summed in the reference's order
// (interpolation.cpp:62-136), which is what makes the Newton iterates equal bit for bit.
struct BiQuadratic {
    float q22, q21, q12, q20, q02, q11, q10, q01, q00;

    __device__ __forceinline__ explicit BiQuadratic(const Patch &m)
    {
        const float cross = m.n + m.w + m.e + m.s, diag = m.nw + m.ne + m.sw + m.se;
        q22 = (float)((double)m.c - 0.5 * (double)cross + 0.25 * (double)diag);
        q21 = (float)(0.5 * (double)(m.n - m.s) + 0.25 * (double)(-m.nw - m.ne + m.sw + m.se));
        q12 = (float)(0.5 * (double)(m.w - m.e) + 0.25 * (double)(-m.nw + m.ne - m.sw + m.se));
        q20 = (float)(0.5 * (double)(m.w + m.e) - (double)m.c);
        q02 = (float)(0.5 * (double)(m.n + m.s) - (double)m.c);
        q11 = quarter_of(m.nw - m.ne - m.sw + m.se);
        q10 = half_of(m.e - m.w);
        q01 = half_of(m.s - m.n);
        q00 = m.c;
    }
    __device__ __forceinline__ float du(float u, float v) const { return 2 * q22 * u * v * v + 2 * q21 * u * v + 2 * q12 * v * v + 2 * q20 * u + q11 * v + q10; }
    __device__ __forceinline__ float dv(float u, float v) const { return 2 * q22 * u * u * v + 2 * q21 * u * u + 2 * q12 * u * v + 2 * q02 * v + q11 * u + q01; }
    __device__ __forceinline__ float duu(float, float v) const { return 2 * q22 * v * v + 2 * q21 * v + 2 * q20; }
    __device__ __forceinline__ float duv(float u, float v) const { return 4 * q22 * u * v + 2 * q21 * u + 2 * q12 * v + q11; }
    __device__ __forceinline__ float dvv(float u, float) const { return 2 * q22 * u * u + 2 * q12 * u + 2 * q02; }
    __device__ __forceinline__ float at(float u, float v) const
    {
        return q22 * u * u * v * v + q21 * u * u * v + q12 * u * v * v + q20 * u * u + q02 * v * v + q11 * u * v + q10 * u + q01 * v + q00;
    }
};

// Newton's method on the gradient of the surface, from the pixel centre: at most 20 steps, stop when the squared gradient
// norm of the step just taken is within 1e-10; a singular Hessian or a stationary point outside [-1, 1]^2 rejects the
// fit (interpolation.cpp:171-212)
__device__ __forceinline__ bool newton_fit(const Patch &m, float &x, float &y, float &strength)
{
    const BiQuadratic f(m);
    const float tol = 1E-10;
    float u = 0, v = 0, gu, gv;
    int step = 0;
    do {
        gu = f.du(u, v);
        gv = f.dv(u, v);
        const float a = f.duu(u, v), b = f.duv(u, v), c = f.dvv(u, v);
        const float det = a * c - b * b;
        if ((double)(det * det) < 1E-10) return false;
        u -= (gu * c - gv * b) / det;
        v -= (gv * a - gu * b) / det;
        step++;
    } while (gu * gu + gv * gv > tol && step < 20);
    if (u > 1 || u < -1 || v > 1 || v < -1 || u != u || v != v) return false;
    x += u;
    y += v;
    strength = f.at(u, v);
    return true;
}

template <int PRECISION>
__global__ void __launch_bounds__(256) harris_refine_kernel(const float *__restrict__ R, int nx, const imgfd_corner *__restrict__ in,
                                                            long long n, imgfd_corner *__restrict__ out)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    imgfd_corner c = in[i];
    const Patch m = patch_at(R, nx, (int)c.x, (int)c.y);  // NMS never emits a border pixel (harris.cpp:157-158)
    if (PRECISION == IMGFD_QUADRATIC_APPROXIMATION) (void)taylor_fit(m, c.x, c.y, c.R);
    else (void)newton_fit(m, c.x, c.y, c.R);
    out[i] = c;
}

}  // namespace

// d_out[i] = d_in[i] refined on the response plane d_R (one frame); precision: IMGFD_QUADRATIC_APPROXIMATION or
// IMGFD_QUARTIC_INTERPOLATION.  d_out may not alias d_in (the caller still ranks by the integer records' strengths).
imgfd_status launch_harris_refine(imgfd_ctx *ctx, const float *d_R, int nx, const imgfd_corner *d_in, int64_t n, int precision,
                                  imgfd_corner *d_out)
{
    if (n <= 0) return IMGFD_OK;
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    if (precision == IMGFD_QUADRATIC_APPROXIMATION)
        hipLaunchKernelGGL(harris_refine_kernel<IMGFD_QUADRATIC_APPROXIMATION>, grid, block, 0, ctx->stream, d_R, nx, d_in, (long long)n, d_out);
    else if (precision == IMGFD_QUARTIC_INTERPOLATION)
        hipLaunchKernelGGL(harris_refine_kernel<IMGFD_QUARTIC_INTERPOLATION>, grid, block, 0, ctx->stream, d_R, nx, d_in, (long long)n, d_out);
    else
        return imgfd_fail(ctx, IMGFD_ERR_INVALID, "launch_harris_refine: no such precision");
    IMGFD_HIP(ctx, hipGetLastError());
    return IMGFD_OK;
}
