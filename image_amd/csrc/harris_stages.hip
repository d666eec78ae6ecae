// image_amd/csrc/harris_stages.hip -- the element-wise stages of the Harris path (K2, K4).
//   gradient   <- central_differences / sobel_operator, image.CornerDetectionHarris/src/gradient.cpp:17-106
//   response   <- compute_corner_response, harris.cpp:78-133
// The library is compiled with -ffp-contract=off: every float expression below rounds exactly where
// the reference's x86-64 build (no FMA) rounds.
#include "common.h"
#include "harris_device.h"

// The reference fills the interior, then copies row 1 / ny-2 into rows 0 / ny-1 for columns
// 1..nx-2 (gradient.cpp:40-46), then copies column 1 / nx-2 into columns 0 / nx-1 for ALL rows
// (:49-55).  Net effect: border pixel (i,j) takes the interior value at (clamp(i,1,ny-2), clamp(j,1,nx-2)).
template <int TYPE>
__global__ void __launch_bounds__(256) gradient_kernel(const float *__restrict__ I, float *__restrict__ dx,
                                                       float *__restrict__ dy, int nx, int ny)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= nx) return;
    const size_t fo = (size_t)blockIdx.z * nx * ny;
    I += fo; dx += fo; dy += fo;
    const int j = min(max(x, 1), nx - 2);
    const int i = min(max(y, 1), ny - 2);
    const int p = i * nx + j;
    float gx, gy;
    if (TYPE == 1) {
        // gradient.cpp:80-87: float sums, double constants, double adds, float store
        gx = (float)(1. / 4. * (I[p + 1] - I[p - 1]) +
                     1. / 8. * (I[p - nx + 1] + I[p + nx + 1] - I[p - nx - 1] - I[p + nx - 1]));
        gy = (float)(1. / 4. * (I[p + nx] - I[p - nx]) +
                     1. / 8. * (I[p + nx + 1] + I[p + nx - 1] - I[p - nx + 1] - I[p - nx - 1]));
    } else {
        // gradient.cpp:34-35: 0.5*(float difference): the halving is exact
        gx = (float)(0.5 * (I[p + 1] - I[p - 1]));
        gy = (float)(0.5 * (I[p + nx] - I[p - nx]));
    }
    dx[(size_t)y * nx + x] = gx;
    dy[(size_t)y * nx + x] = gy;
}

imgfd_status launch_gradient(imgfd_ctx *ctx, const float *d_I, float *d_Ix, float *d_Iy, int nx, int ny,
                             int n_frames, int type)
{
    dim3 grid(ceil_div(nx, 256), ny, n_frames);
    if (type == IMGFD_SOBEL_OPERATOR)
        hipLaunchKernelGGL(gradient_kernel<1>, grid, dim3(256), 0, ctx->stream, d_I, d_Ix, d_Iy, nx, ny);
    else
        hipLaunchKernelGGL(gradient_kernel<0>, grid, dim3(256), 0, ctx->stream, d_I, d_Ix, d_Iy, nx, ny);
    IMGFD_HIP(ctx, hipGetLastError());
    return IMGFD_OK;
}

template <int MEASURE>
__global__ void __launch_bounds__(256) response_kernel(const float *__restrict__ A, const float *__restrict__ B,
                                                       const float *__restrict__ C, float *__restrict__ R,
                                                       size_t n, float k)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    R[i] = harris_response_value<MEASURE>(A[i], B[i], C[i], k);
}

imgfd_status launch_response(imgfd_ctx *ctx, const float *d_A, const float *d_B, const float *d_C,
                             float *d_R, int nx, int ny, int n_frames, int measure, float k)
{
    const size_t n = (size_t)nx * ny * n_frames;
    dim3 grid((unsigned)((n + 255) / 256));
    if (measure == IMGFD_SHI_TOMASI_MEASURE)
        hipLaunchKernelGGL(response_kernel<1>, grid, dim3(256), 0, ctx->stream, d_A, d_B, d_C, d_R, n, k);
    else if (measure == IMGFD_HARMONIC_MEAN_MEASURE)
        hipLaunchKernelGGL(response_kernel<2>, grid, dim3(256), 0, ctx->stream, d_A, d_B, d_C, d_R, n, k);
    else
        hipLaunchKernelGGL(response_kernel<0>, grid, dim3(256), 0, ctx->stream, d_A, d_B, d_C, d_R, n, k);
    IMGFD_HIP(ctx, hipGetLastError());
    return IMGFD_OK;
}
