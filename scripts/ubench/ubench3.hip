// scripts/ubench/ubench3.hip -- do the f64 matrix pipe (v_mfma_f64_16x16x4_f64) and the f64 vector pipe (v_fma_f64) of a
// gfx950 SIMD run side by side?  MODE 0: VALU only, 1: MFMA only, 2: both interleaved in every wave, 3: waves 0,1 of a
// workgroup VALU / waves 2,3 MFMA.  If the pipes are independent, mode 2/3 take max(mode 0, mode 1), otherwise the sum.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
#define ITER 512
typedef double d4 __attribute__((ext_vector_type(4)));

template <int MODE, int NV, int NM>
__global__ void __launch_bounds__(256) k(double *out, double seed)
{
    double a[8];
    d4 acc[4];
    for (int i = 0; i < 8; i++) a[i] = seed + threadIdx.x * 1e-3 + i;
    for (int i = 0; i < 4; i++) acc[i] = (d4){seed, seed + 1, seed + 2, seed + 3};
    const double b = seed * 0.999, c = seed * 1e-3;
    const double ma = seed * 1e-3 + (threadIdx.x & 63) * 1e-6, mb = seed * 2e-3;
    const int wave = threadIdx.x >> 6;
    const bool do_v = MODE == 0 || MODE == 2 || (MODE == 3 && wave < 2);
    const bool do_m = MODE == 1 || MODE == 2 || (MODE == 3 && wave >= 2);
    for (int it = 0; it < ITER; it++) {
        if (do_m) {
#pragma unroll
            for (int i = 0; i < NM; i++) acc[i & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(ma, mb, acc[i & 3], 0, 0, 0);
        }
        if (do_v) {
#pragma unroll
            for (int i = 0; i < NV; i++) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i & 7]) : "v"(b), "v"(c));
        }
    }
    double s = 0;
    for (int i = 0; i < 8; i++) s += a[i];
    for (int i = 0; i < 4; i++) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, int NV, int NM>
static double run(const char *name)
{
    double *d;
    const int blocks = 256 * 2;  // 2 workgroups (8 waves) per CU = 2 waves per SIMD
    CHECK(hipMalloc(&d, sizeof(double) * blocks * 256));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int w = 0; w < 2; w++) hipLaunchKernelGGL((k<MODE, NV, NM>), dim3(blocks), dim3(256), 0, 0, d, 1.0001);
    CHECK(hipEventRecord(e0));
    const int reps = 10;
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL((k<MODE, NV, NM>), dim3(blocks), dim3(256), 0, 0, d, 1.0001);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps;
    printf("%-34s NV %2d NM %2d : %8.1f us per launch\n", name, NV, NM, us);
    CHECK(hipFree(d));
    return us;
}

int main()
{
    // per iteration and wave: NV v_fma_f64 (64 FMA each) and NM mfma 16x16x4 (1024 FMA each)
    const double v = run<0, 32, 4>("VALU only");
    const double m = run<1, 32, 4>("MFMA only");
    const double b = run<2, 32, 4>("both, interleaved in each wave");
    const double s = run<3, 32, 4>("both, split by wave (2 + 2)");
    const double fv = 512.0 * 256 * ITER * 32 * 2 / (v * 1e-6) / 1e12 * 2;   // FLOP/s of the VALU stream
    const double fm = 512.0 * 4 * ITER * 4 * 1024 * 2 / (m * 1e-6) / 1e12;   // waves * ITER * NM * 1024 FMA * 2
    printf("VALU stream %.1f TFLOP/s, MFMA stream %.1f TFLOP/s\n", fv / 2, fm);
    printf("interleaved: %.2f x (VALU + MFMA);  split by wave: time %.1f us vs max(VALU,MFMA)/2 = %.1f\n", b / (v + m), s, (v > m ? v : m) / 2);
    return 0;
}
