// scripts/ubench/ubench4.hip -- HBM throughput of tile-shaped access: every workgroup reads (and/or writes) a tile of
// W floats x H rows out of 3840-float rows (the 4K planes of the Harris path), float4 per lane, for W = 64 .. 1024.
// Answers: does a 64- or 128-column tile cost bandwidth against wider row segments?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
#define NX 3840
#define NY 2160

template <int MODE>  // 0: read only (sum), 1: write only, 2: copy
__global__ void __launch_bounds__(256) tile_rw(const float4 *__restrict__ src, float4 *__restrict__ dst, int W4, int H, float *sink)
{
    const size_t frame = (size_t)blockIdx.z * (NX / 4) * NY;
    const int x4 = blockIdx.x * W4, y0 = blockIdx.y * H;
    float acc = 0.f;
    for (int i = threadIdx.x; i < W4 * H; i += 256) {
        const int r = i / W4, q = i - r * W4;
        const size_t o = frame + (size_t)(y0 + r) * (NX / 4) + x4 + q;
        if (MODE == 0) { const float4 v = src[o]; acc += v.x + v.y + v.z + v.w; }
        else if (MODE == 1) dst[o] = make_float4(1.f, 2.f, 3.f, (float)i);
        else dst[o] = src[o];
    }
    if (MODE == 0 && acc == 12345.678f) *sink = acc;
}

template <int MODE>
static void run(const float4 *s, float4 *d, float *sink, int W, int H, int frames)
{
    dim3 grid(NX / W, NY / H, frames);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int w = 0; w < 2; w++) hipLaunchKernelGGL(tile_rw<MODE>, grid, dim3(256), 0, 0, s, d, W / 4, H, sink);
    CHECK(hipEventRecord(e0));
    const int reps = 5;
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL(tile_rw<MODE>, grid, dim3(256), 0, 0, s, d, W / 4, H, sink);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (double)NX * NY * 4 * frames * (MODE == 2 ? 2 : 1) * reps;
    printf("%-5s tile %4d x %3d : %6.2f TB/s\n", MODE == 0 ? "read" : MODE == 1 ? "write" : "copy", W, H, bytes / (ms * 1e-3) / 1e12);
}

int main()
{
    const int frames = 48;  // 1.6 GB per plane set: far beyond the 256 MB Infinity Cache
    const size_t n = (size_t)NX * NY * frames;
    float4 *s, *d; float *sink;
    CHECK(hipMalloc(&s, n * 4)); CHECK(hipMalloc(&d, n * 4)); CHECK(hipMalloc(&sink, 4));
    CHECK(hipMemset(s, 0, n * 4));
    const int shapes[][2] = {{64, 64}, {64, 24}, {128, 16}, {128, 32}, {256, 16}, {480, 8}, {960, 8}, {3840, 2}};
    for (auto &sh : shapes) { run<0>(s, d, sink, sh[0], sh[1], frames); run<1>(s, d, sink, sh[0], sh[1], frames); run<2>(s, d, sink, sh[0], sh[1], frames); }
    return 0;
}
