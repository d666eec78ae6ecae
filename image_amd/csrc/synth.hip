// image_amd/csrc/synth.hip -- device twin of image_amd/synth.py: seeded synthetic u8 frames written
// straight into HBM (bench.py, config-5 style streams), so no host->device copy sits in front of the
// detectors.  Same recipe bit for bit: triangle-wave background, painter's-order rectangles, lowbias32
// hash noise 0..15.
#include "common.h"

__device__ __forceinline__ unsigned synth_hash(unsigned x)
{
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return x;
}

// rects: n_frames * n_rect rows of {x0, y0, x1, y1, value}
__global__ void __launch_bounds__(256) synth_kernel(unsigned char *__restrict__ out, int nx, int ny, size_t frame_stride,
                                                    unsigned seed0, const int *__restrict__ rects, int n_rect)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    const int f = blockIdx.z;
    if (x >= nx) return;
    const int t = (x + 2 * y) & 255;
    int v = 60 + ((t < 128 ? t : 255 - t) >> 1);
    const int *r = rects + (size_t)f * n_rect * 5;
    for (int i = 0; i < n_rect; i++)
        if (x >= r[5 * i] && x < r[5 * i + 2] && y >= r[5 * i + 1] && y < r[5 * i + 3]) v = r[5 * i + 4];
    const unsigned seed = seed0 + (unsigned)f;
    v += (int)(synth_hash(seed * 0x9E3779B9u + (unsigned)(y * nx + x)) >> 28);
    out[(size_t)f * frame_stride + (size_t)y * nx + x] = (unsigned char)min(v, 255);
}

extern "C" imgfd_status imgfd_synth_frames(imgfd_ctx *ctx, uint8_t *d_frames, int n_frames, int nx, int ny,
                                           size_t frame_stride_bytes, uint32_t seed0, const int32_t *d_rects,
                                           int n_rect)
{
    if (!ctx || !d_frames || n_frames < 0 || nx < 1 || ny < 1 || (n_rect > 0 && !d_rects))
        return imgfd_fail(ctx, IMGFD_ERR_INVALID, "imgfd_synth_frames: bad argument");
    if (!n_frames) return IMGFD_OK;
    dim3 grid(ceil_div(nx, 256), ny, n_frames);
    hipLaunchKernelGGL(synth_kernel, grid, dim3(256), 0, ctx->stream, d_frames, nx, ny, frame_stride_bytes, seed0,
                       d_rects, n_rect);
    IMGFD_HIP(ctx, hipGetLastError());
    return IMGFD_OK;
}
