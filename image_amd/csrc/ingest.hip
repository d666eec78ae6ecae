// image_amd/csrc/ingest.hip -- upload of an R-native vector with the glue's narrowing done on the device.
//
// The reference glue narrows the R vectors on the host, one element at a time:
//   (float) x[i]          NumericVector -> float         rcpp_harris.cpp:34-35
//   (unsigned char) x[i]  IntegerVector -> unsigned char  f9_rcpp.cpp:10-11, rcpp_canny.cpp:135-136
//   rgb_pixel(x[i], ..)   std::vector<int> -> bytes       rcpp_fhog.cpp:17-24, rcpp_surf.cpp:14-21
// The *_f64 / *_i32 entry points of the C ABI take those vectors as they are (what `REAL(x)` / `INTEGER(x)` point to),
// copy them over PCIe once and narrow in HBM: a 4K frame of ints is 33 MB (0.5 ms at 63 GB/s) while the host loop over
// 8.3 M elements costs several milliseconds of the single R thread.  Same casts, same values.
#include "common.h"

__global__ void __launch_bounds__(256) narrow_i32_u8(const int *__restrict__ in, unsigned char *__restrict__ out, size_t n)
{
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i + 3 < n) {
        const int4 v = *reinterpret_cast<const int4 *>(in + i);
        *reinterpret_cast<uchar4 *>(out + i) = make_uchar4((unsigned char)v.x, (unsigned char)v.y, (unsigned char)v.z, (unsigned char)v.w);
    } else {
        for (size_t k = i; k < n; k++) out[k] = (unsigned char)in[k];
    }
}

__global__ void __launch_bounds__(256) narrow_f64_f32(const double *__restrict__ in, float *__restrict__ out, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = (float)in[i];
}

// ---- the way back.  The reference glue widens its results into R's doubles one element at a time on the R thread
// (rcpp_canny.cpp:226-233: 8.3 M stores for a 4K edge map; rcpp_fhog.cpp:29-38); the *_f64out entry points widen in HBM and
// copy straight into the vector R allocated.
__global__ void __launch_bounds__(256) widen_u8_f64(const unsigned char *__restrict__ in, double *__restrict__ out, size_t n)
{
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i + 3 < n) {
        const uchar4 v = *reinterpret_cast<const uchar4 *>(in + i);
        double2 *o = reinterpret_cast<double2 *>(out + i);
        o[0] = double2{(double)v.x, (double)v.y};
        o[1] = double2{(double)v.z, (double)v.w};
    } else {
        for (size_t k = i; k < n; k++) out[k] = (double)in[k];
    }
}
__global__ void __launch_bounds__(256) widen_f32_f64(const float *__restrict__ in, double *__restrict__ out, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = (double)in[i];
}
// d_src (u8 or f32, n elements, 4-byte aligned) -> d_stage (doubles, 16-byte aligned) -> host_out, asynchronously on the context's stream
imgfd_status download_widened(imgfd_ctx *ctx, const void *d_src, bool src_is_u8, size_t n, double *d_stage, double *host_out)
{
    if (!n) return IMGFD_OK;
    if (src_is_u8) hipLaunchKernelGGL(widen_u8_f64, dim3((unsigned)((n / 4 + 256) / 256)), dim3(256), 0, ctx->stream, (const unsigned char *)d_src, d_stage, n);
    else hipLaunchKernelGGL(widen_f32_f64, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (const float *)d_src, d_stage, n);
    IMGFD_HIP(ctx, hipGetLastError());
    IMGFD_HIP(ctx, hipMemcpyAsync(host_out, d_stage, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    return IMGFD_OK;
}

size_t upload_stage_bytes(int kind, size_t n)
{
    if (kind == IMGFD_SRC_I32) return align_up(n * sizeof(int), 256);
    if (kind == IMGFD_SRC_F64) return align_up(n * sizeof(double), 256);
    return 0;
}

// copies n elements to d_dst (u8 for IMGFD_SRC_U8 / _I32, f32 for _F32 / _F64); the staging copy comes out of the
// workspace (the caller's reservation includes upload_stage_bytes)
imgfd_status upload_image(imgfd_ctx *ctx, const void *host, int kind, size_t n, void *d_dst)
{
    if (kind == IMGFD_SRC_U8 || kind == IMGFD_SRC_F32) {
        IMGFD_HIP(ctx, hipMemcpyAsync(d_dst, host, n * (kind == IMGFD_SRC_U8 ? 1 : 4), hipMemcpyHostToDevice, ctx->stream));
        return IMGFD_OK;
    }
    void *stage = ws_alloc(ctx, upload_stage_bytes(kind, n));
    if (!stage) return imgfd_fail(ctx, IMGFD_ERR_OOM, "workspace reservation too small");
    if (kind == IMGFD_SRC_I32) {
        IMGFD_HIP(ctx, hipMemcpyAsync(stage, host, n * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(narrow_i32_u8, dim3((unsigned)((n / 4 + 256) / 256)), dim3(256), 0, ctx->stream, (const int *)stage,
                           (unsigned char *)d_dst, n);
    } else {
        IMGFD_HIP(ctx, hipMemcpyAsync(stage, host, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(narrow_f64_f32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (const double *)stage,
                           (float *)d_dst, n);
    }
    IMGFD_HIP(ctx, hipGetLastError());
    return IMGFD_OK;
}
