// scripts/ubench/ubench2.hip -- f64 VALU rate vs waves per SIMD and instruction-level parallelism on gfx950.
// Pattern A: independent v_fma_f64 chains (ILP accumulators).  Pattern B: the FIR inner step
// (v_add_f64 tmp = x+y ; v_fmac_f64 acc += c*tmp) with ILP independent accumulators.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
#define ITER 1024

template <int ILP, int PAT>
__global__ void __launch_bounds__(256) k(double *out, double seed)
{
    extern __shared__ char lds[];
    double a[ILP], x[ILP];
    for (int i = 0; i < ILP; i++) { a[i] = seed + threadIdx.x * 1e-3 + i; x[i] = seed * 0.5 + i; }
    const double b = seed * 0.999, c = seed * 1e-3;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int i = 0; i < ILP; i++) {
            if (PAT == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            else {
                double t;
                asm volatile("v_add_f64 %0, %1, %2" : "=v"(t) : "v"(x[i]), "v"(c));
                asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(a[i]) : "s"(b), "v"(t));
            }
        }
    }
    double s = 0;
    for (int i = 0; i < ILP; i++) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (seed == 12345.0) lds[threadIdx.x] = 1;
}

template <int ILP, int PAT>
static void run(int wgs_per_cu)
{
    double *d;
    const int blocks = 256 * wgs_per_cu;
    CHECK(hipMalloc(&d, sizeof(double) * blocks * 256));
    const size_t lds = (160 * 1024) / wgs_per_cu - 512;  // caps residency at wgs_per_cu workgroups per CU
    CHECK(hipFuncSetAttribute((const void *)k<ILP, PAT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int w = 0; w < 2; w++) hipLaunchKernelGGL((k<ILP, PAT>), dim3(blocks), dim3(256), lds, 0, d, 1.0001);
    CHECK(hipEventRecord(e0));
    const int reps = 10;
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL((k<ILP, PAT>), dim3(blocks), dim3(256), lds, 0, d, 1.0001);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double ops = (double)reps * blocks * 256 * ITER * ILP * (PAT == 0 ? 1 : 2);
    const double per_simd_cyc = (ms * 1e-3 / reps) * 2.4e9 / ((double)ITER * ILP * (PAT == 0 ? 1 : 2) * wgs_per_cu);
    printf("pat %s  ILP %d  waves/SIMD %d : %7.2f Tlane-op/s   %.2f cycles(2.4GHz)/wave-instr/SIMD\n", PAT == 0 ? "fma    " : "add+fmac",
           ILP, wgs_per_cu, ops / (ms * 1e-3) / 1e12, per_simd_cyc);
    CHECK(hipFree(d));
}

int main()
{
    int occ[] = {1, 2, 4, 8};
    for (int o : occ) { run<1, 0>(o); run<2, 0>(o); run<4, 0>(o); run<8, 0>(o); }
    for (int o : occ) { run<1, 1>(o); run<2, 1>(o); run<4, 1>(o); run<8, 1>(o); }
    return 0;
}
