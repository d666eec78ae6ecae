// scripts/ubench/ubench7.hip -- issue rate of the instructions of the structure-tensor kernel's inner loops on gfx950, one kind at
// a time (8 independent destinations per wave, 3 waves per SIMD = the kernel's occupancy): are the f32<->f64 conversions
// full-rate like v_fma_f64 / v_add_f64, or do they take more issue slots?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
#define ITER 512
#define ILP 8

template <int PAT>
__global__ void __launch_bounds__(256) k(double *out, double seed)
{
    extern __shared__ char lds[];
    double a[ILP];
    float f[ILP];
    for (int i = 0; i < ILP; i++) { a[i] = seed + threadIdx.x * 1e-3 + i; f[i] = (float)(seed * 0.5 + i); }
    const double b = seed * 0.999, c = seed * 1e-3;
    const float bf = (float)b, cf = (float)c;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int i = 0; i < ILP; i++) {
            if (PAT == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            else if (PAT == 1) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            else if (PAT == 2) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            else if (PAT == 3) asm volatile("v_cvt_f64_f32_e32 %0, %1" : "=v"(a[i]) : "v"(f[i]));
            else if (PAT == 4) asm volatile("v_cvt_f32_f64_e32 %0, %1" : "=v"(f[i]) : "v"(a[i]));
            else if (PAT == 5) asm volatile("v_mul_f32_e32 %0, %0, %1" : "+v"(f[i]) : "v"(bf));
            else if (PAT == 6) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(bf), "v"(cf));
            else if (PAT == 7) asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(a[i]) : "s"(b), "v"(c));
            else if (PAT == 8) {  // the FIR step as the kernel issues it: pair add + fmac with a scalar tap
                double t;
                asm volatile("v_add_f64 %0, %1, %2" : "=v"(t) : "v"(a[(i + 1) % ILP]), "v"(c));
                asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(a[i]) : "s"(b), "v"(t));
            }
        }
    }
    double s = 0;
    for (int i = 0; i < ILP; i++) s += a[i] + f[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (seed == 12345.0) lds[threadIdx.x] = 1;
}

template <int PAT>
static void run(const char *name, int wgs_per_cu)
{
    double *d;
    const int blocks = 256 * wgs_per_cu;
    CHECK(hipMalloc(&d, sizeof(double) * blocks * 256));
    const size_t lds = (160 * 1024) / wgs_per_cu - 512;  // caps residency at wgs_per_cu workgroups (of 4 waves) per CU
    CHECK(hipFuncSetAttribute((const void *)k<PAT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int w = 0; w < 2; w++) hipLaunchKernelGGL((k<PAT>), dim3(blocks), dim3(256), lds, 0, d, 1.0001);
    CHECK(hipEventRecord(e0));
    const int reps = 10;
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL((k<PAT>), dim3(blocks), dim3(256), lds, 0, d, 1.0001);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const int per = PAT == 8 ? 2 : 1;
    const double ops = (double)reps * blocks * 256 * ITER * ILP * per;
    const double per_simd_ns = (ms * 1e6 / reps) / ((double)ITER * ILP * per * wgs_per_cu);
    printf("%-22s waves/SIMD %d : %7.2f Tlane-op/s   %.3f ns per wave-instruction per SIMD\n", name, wgs_per_cu, ops / (ms * 1e-3) / 1e12, per_simd_ns);
    CHECK(hipFree(d));
}

int main()
{
    for (int o : {3, 2}) {
        run<0>("v_fma_f64", o); run<7>("v_fmac_f64 (s tap)", o); run<1>("v_add_f64", o); run<2>("v_mul_f64", o);
        run<8>("add_f64 + fmac_f64", o);
        run<3>("v_cvt_f64_f32", o); run<4>("v_cvt_f32_f64", o); run<5>("v_mul_f32", o); run<6>("v_fma_f32", o);
    }
    return 0;
}
