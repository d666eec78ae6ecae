// scripts/ubench/ubench5.hip -- does reading three planes at the same tile coordinates (A, B, C of the Harris response
// pass) lose bandwidth to DRAM channel aliasing?  Reads 64x64-float tiles of three planes whose base addresses differ by
// `stride` bytes, for the stride the workspace uses (32 frames x 3840 x 2160 x 4 B) and for padded strides.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
#define NX 3840
#define NY 2160

template <int NP>
__global__ void __launch_bounds__(256) tile_read(const char *__restrict__ base, size_t stride, float *sink)
{
    const size_t frame = (size_t)blockIdx.z * (NX / 4) * NY;
    const int x4 = blockIdx.x * 16, y0 = blockIdx.y * 64;
    float acc = 0.f;
    for (int i = threadIdx.x; i < 16 * 64; i += 256) {
        const int r = i / 16, q = i - r * 16;
        const size_t o = frame + (size_t)(y0 + r) * (NX / 4) + x4 + q;
#pragma unroll
        for (int p = 0; p < NP; p++) {
            const float4 v = reinterpret_cast<const float4 *>(base + p * stride)[o];
            acc += v.x + v.y + v.z + v.w;
        }
    }
    if (acc == 12345.678f) *sink = acc;
}

template <int NP>
static void run(const char *base, size_t stride, float *sink, int frames, const char *what)
{
    dim3 grid(NX / 64, NY / 64 + 0, frames);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int w = 0; w < 2; w++) hipLaunchKernelGGL(tile_read<NP>, grid, dim3(256), 0, 0, base, stride, sink);
    CHECK(hipEventRecord(e0));
    const int reps = 5;
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL(tile_read<NP>, grid, dim3(256), 0, 0, base, stride, sink);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (double)NX * (NY / 64 * 64) * 4 * frames * NP * reps;
    printf("%d plane(s), stride %-28s : %6.2f TB/s\n", NP, what, bytes / (ms * 1e-3) / 1e12);
}

int main()
{
    const int frames = 32;
    const size_t plane = (size_t)NX * NY * 4 * frames;  // 1 061 683 200 B
    char *buf; float *sink;
    CHECK(hipMalloc(&buf, 3 * plane + (64 << 20))); CHECK(hipMalloc(&sink, 4));
    CHECK(hipMemset(buf, 0, 3 * plane + (64 << 20)));
    run<1>(buf, plane, sink, frames, "-");
    run<3>(buf, plane, sink, frames, "plane (as in the workspace)");
    run<3>(buf, plane + 256, sink, frames, "plane + 256 B");
    run<3>(buf, plane + 4096, sink, frames, "plane + 4 KiB");
    run<3>(buf, plane + 65536 + 4096, sink, frames, "plane + 68 KiB");
    run<3>(buf, plane + (1 << 20) + 8192, sink, frames, "plane + 1 MiB + 8 KiB");
    run<3>(buf, plane + (16 << 20) + 256 * 37, sink, frames, "plane + 16 MiB + 9472 B");
    return 0;
}
