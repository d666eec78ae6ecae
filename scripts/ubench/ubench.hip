// scripts/ubench/ubench.hip -- instruction-rate microbenchmarks that decide the FIR kernel's design:
// how fast are v_fma_f64 / v_add_f64 / v_mul_f64 / v_cvt_f64_f32 / v_cvt_f32_f64 on gfx950?
// Build: hipcc -O3 --offload-arch=gfx950 ubench.hip -o ubench.bin ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

#define ITER 2048
#define ILP 8

template <int OP>
__global__ void __launch_bounds__(256) rate(double *out, double seed, float fseed)
{
    double a[ILP]; float f[ILP];
    for (int i = 0; i < ILP; i++) { a[i] = seed + threadIdx.x * 1e-3 + i; f[i] = fseed + threadIdx.x * 1e-3f + i; }
    const double b = seed * 0.999, c = seed * 1e-3;
    const float fb = (float)b, fc = (float)c;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int i = 0; i < ILP; i++) {
            if (OP == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            else if (OP == 1) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            else if (OP == 2) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            else if (OP == 3) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a[i]) : "v"(f[i]));
            else if (OP == 4) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[i]) : "v"(a[i]));
            else if (OP == 5) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(fb), "v"(fc));
            else if (OP == 6) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[i]) : "v"(fb));
        }
    }
    double s = 0;
    for (int i = 0; i < ILP; i++) s += a[i] + f[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP>
static void run(const char *name, double extra_per_op)
{
    double *d;
    const int blocks = 256 * 8;
    CHECK(hipMalloc(&d, sizeof(double) * blocks * 256));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int w = 0; w < 2; w++) hipLaunchKernelGGL(rate<OP>, dim3(blocks), dim3(256), 0, 0, d, 1.0001, 1.0001f);
    CHECK(hipEventRecord(e0));
    const int reps = 10;
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL(rate<OP>, dim3(blocks), dim3(256), 0, 0, d, 1.0001, 1.0001f);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double ops = (double)reps * blocks * 256 * ITER * ILP;
    printf("%-28s %8.2f Tlane-op/s  (%.3f ms/launch)%s\n", name, ops / (ms * 1e-3) / 1e12, ms / reps,
           extra_per_op > 0 ? "  [includes 1 int xor per op]" : "");
    CHECK(hipFree(d));
}

// streaming copy bandwidth, float4 per lane
__global__ void __launch_bounds__(256) copy4(const float4 *in, float4 *out, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}

int main()
{
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    printf("device: %s %s CUs=%d clock=%d kHz\n", p.name, p.gcnArchName, p.multiProcessorCount, p.clockRate);
    run<0>("v_fma_f64", 0);
    run<1>("v_add_f64", 0);
    run<2>("v_mul_f64", 0);
    run<3>("v_cvt_f64_f32", 0);
    run<4>("v_cvt_f32_f64", 0);
    run<5>("v_fma_f32", 0);
    run<6>("v_mul_f32", 0);
    for (size_t mb : {64, 166, 1024}) {
        size_t n = mb * 1024 * 1024 / 16 / 2;
        float4 *a, *b;
        CHECK(hipMalloc(&a, n * 16)); CHECK(hipMalloc(&b, n * 16));
        CHECK(hipMemset(a, 1, n * 16));
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        for (int w = 0; w < 3; w++) hipLaunchKernelGGL(copy4, dim3(2048), dim3(256), 0, 0, a, b, n);
        CHECK(hipEventRecord(e0));
        for (int r = 0; r < 20; r++) hipLaunchKernelGGL(copy4, dim3(2048), dim3(256), 0, 0, a, b, n);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("copy %5zu MB total traffic: %.2f TB/s\n", mb, 20.0 * n * 32 / (ms * 1e-3) / 1e12);
        CHECK(hipFree(a)); CHECK(hipFree(b));
    }
    return 0;
}
