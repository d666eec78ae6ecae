// image_amd/csrc/nms.hip -- Harris non-maximum suppression (K5).
//
// Replaces non_maximum_suppression(), image.CornerDetectionHarris/src/harris.cpp:141-255.  The
// reference walks each row with a Neubeck-style scan line and a shared skip[] array; what it decides
// is a pure window rule (checked against the compiled reference in the tests):
//   pixel (i,j), radius r, r <= i < ny-r, r <= j < nx-r, R(i,j) >= Th, is a corner iff every other
//   pixel q of its (2r+1)^2 window satisfies
//       R(q) <  R(i,j)   when q is in a row above, or to the right in the same row   (:188, :223-233)
//       R(q) <= R(i,j)   when q is in a row below, or to the left in the same row    (:201, :209-219)
// One thread per pixel, one wave per 64 consecutive pixels of a row, so that a single __ballot is the
// 64-bit word of the corner bit mask consumed by compact.hip; per-row corner counts are accumulated
// with one integer atomic per non-empty word.  A cheap 3x3 pre-test rejects almost every pixel above
// the threshold before the full window is read (reads hit L1/L2: R is streamed once from HBM).
#include "common.h"

__global__ void __launch_bounds__(256) harris_nms_kernel(const float *__restrict__ R, int nx, int ny, float Th,
                                                         int radius, unsigned long long *__restrict__ mask,
                                                         unsigned *__restrict__ rowcount, int words_per_row)
{
    const int lane = threadIdx.x & 63;
    const int x = blockIdx.x * 64 + lane;
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int frame = blockIdx.z;
    const float *Rf = R + (size_t)frame * nx * ny;
    bool corner = false;
    if (y < ny && x >= radius && x < nx - radius && y >= radius && y < ny - radius) {
        const float *c = Rf + (size_t)y * nx + x;
        const float v = *c;
        if (!(v < Th)) {  // skip[] = R < Th, harris.cpp:160-162
            // 3x3 pre-test with the window rule's own comparisons
            bool ok = !(c[-nx - 1] >= v) && !(c[-nx] >= v) && !(c[-nx + 1] >= v) && !(c[1] >= v) &&
                      !(c[-1] > v) && !(c[nx - 1] > v) && !(c[nx] > v) && !(c[nx + 1] > v);
            if (ok) {
                for (int dy = -radius; dy <= radius && ok; dy++) {
                    const float *row = c + (ptrdiff_t)dy * nx;
                    if (dy < 0) {
                        for (int dx = -radius; dx <= radius; dx++) ok = ok && !(row[dx] >= v);
                    } else if (dy > 0) {
                        for (int dx = -radius; dx <= radius; dx++) ok = ok && !(row[dx] > v);
                    } else {
                        for (int dx = -radius; dx < 0; dx++) ok = ok && !(row[dx] > v);
                        for (int dx = 1; dx <= radius; dx++) ok = ok && !(row[dx] >= v);
                    }
                }
            }
            corner = ok;
        }
    }
    const unsigned long long word = __ballot(corner);
    if (lane == 0 && y < ny && (int)blockIdx.x < words_per_row) {
        mask[((size_t)frame * ny + y) * words_per_row + blockIdx.x] = word;
        if (word) atomicAdd(&rowcount[(size_t)frame * ny + y], (unsigned)__popcll(word));
    }
}

imgfd_status launch_harris_nms(imgfd_ctx *ctx, const float *d_R, int nx, int ny, int n_frames, float Th,
                               int radius, const CompactBuffers &cb)
{
    // harris.cpp:151-152: nothing is detected on images not larger than the window; radius >= 1
    if (ny <= 2 * radius + 1 || nx <= 2 * radius + 1) radius = nx + ny;  // empties the search domain
    else if (radius < 1) radius = 1;
    dim3 grid(cb.words_per_row, ceil_div(ny, 4), n_frames);
    hipLaunchKernelGGL(harris_nms_kernel, grid, dim3(256), 0, ctx->stream, d_R, nx, ny, Th, radius, cb.mask,
                       cb.rowcount, cb.words_per_row);
    IMGFD_HIP(ctx, hipGetLastError());
    return IMGFD_OK;
}
