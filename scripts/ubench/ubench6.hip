// scripts/ubench/ubench6.hip -- do v_fma_f64 (vector pipe) and v_mfma_f64_16x16x4_f64 (matrix pipe) overlap on gfx950?
// A 512-thread workgroup per CU (waves 0-3 and 4-7: one of each group per SIMD).  Mode 0: both groups run the FIR inner
// step on the vector pipe; mode 1: both run MFMA chains; mode 2: group A vector, group B MFMA; modes 3/4: one group idle.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
#define ITER 2048
typedef double d4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double valu_work(double seed)
{
    double a[4], x[4];
    for (int i = 0; i < 4; i++) { a[i] = seed + threadIdx.x * 1e-3 + i; x[i] = seed * 0.5 + i; }
    const double b = seed * 0.999, c = seed * 1e-3;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            double t;
            asm volatile("v_add_f64 %0, %1, %2" : "=v"(t) : "v"(x[i]), "v"(c));
            asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(t));
        }
    }
    return a[0] + a[1] + a[2] + a[3];
}
__device__ __forceinline__ double mfma_work(double seed, int n)
{
    d4 acc[4];
    for (int i = 0; i < 4; i++) acc[i] = d4{seed, seed, seed, seed};
    const double a = seed * 1e-3 + threadIdx.x * 1e-6, b = seed * 0.5;
    for (int it = 0; it < n; it++) {
#pragma unroll
        for (int i = 0; i < 4; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    return acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
}

__global__ void __launch_bounds__(512) k(double *out, double seed, int mode, int mfma_iters)
{
    const int grp = threadIdx.x >> 8;  // waves 0-3 | 4-7
    double r = 0;
    bool valu = false, mfma = false;
    if (mode == 0) valu = true;
    else if (mode == 1) mfma = true;
    else if (mode == 2) { valu = grp == 0; mfma = grp == 1; }
    else if (mode == 3) valu = grp == 0;
    else if (mode == 4) mfma = grp == 1;
    if (valu) r = valu_work(seed);
    if (mfma) r = mfma_work(seed, mfma_iters);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

int main()
{
    double *d;
    const int blocks = 256;
    CHECK(hipMalloc(&d, sizeof(double) * blocks * 512));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const char *names[] = {"vector + vector", "matrix + matrix", "vector + matrix", "vector alone  ", "matrix alone  "};
    for (int mi = 64; mi <= 256; mi *= 2)
        for (int mode = 0; mode < 5; mode++) {
            for (int w = 0; w < 2; w++) hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, d, 1.0001, mode, mi);
            CHECK(hipEventRecord(e0));
            const int reps = 20;
            for (int r = 0; r < reps; r++) hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, d, 1.0001, mode, mi);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            const double us = ms * 1e3 / reps;
            // per SIMD: one wave of each group
            const double vinstr = ITER * 8.0, minstr = mi * 4.0;
            printf("mfma iters %4d  %s : %8.1f us   (vector wave-instr per group %.0f, mfma per group %.0f; cycles@2.4GHz per SIMD %.0f)\n", mi, names[mode],
                   us, vinstr, minstr, us * 2400.0);
        }
    return 0;
}
