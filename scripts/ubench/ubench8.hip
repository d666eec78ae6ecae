// scripts/ubench/ubench8.hip -- issue cost of the integer / cross-lane instructions of the Canny hysteresis sweep on gfx950
// (canny_hyst_bits: 64-bit shifts and adds, DPP wave shifts, bit reversal, 64-bit compares), 8 independent destinations per
// wave, 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
#define ITER 512
#define ILP 8

template <int PAT>
__global__ void __launch_bounds__(256) k(unsigned long long *out, unsigned long long seed)
{
    extern __shared__ char lds[];
    unsigned long long a[ILP];
    unsigned b[ILP], c[ILP];
    for (int i = 0; i < ILP; i++) { a[i] = seed * (threadIdx.x + 1) + i; b[i] = (unsigned)(a[i] >> 7); c[i] = (unsigned)(a[i] >> 13); }
    const unsigned long long k64 = seed | 3;
    const unsigned k32 = (unsigned)seed | 5;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int i = 0; i < ILP; i++) {
            if (PAT == 0) asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(a[i]));
            else if (PAT == 1) asm volatile("v_lshrrev_b64 %0, 1, %0" : "+v"(a[i]));
            else if (PAT == 2) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(a[i]) : "v"(k64));
            else if (PAT == 3) { asm volatile("v_add_co_u32_e32 %0, vcc, %0, %1\n\tv_addc_co_u32_e32 %2, vcc, %2, %3, vcc" : "+v"(b[i]), "+v"(c[i]) : "v"(k32), "v"(k32) : "vcc"); }
            else if (PAT == 4) asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(b[i]) : "v"(c[i]));
            else if (PAT == 5) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(b[i]) : "v"(c[i]), "v"(k32));
            else if (PAT == 6) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(b[i]) : "v"(c[i]));
            else if (PAT == 7) asm volatile("v_bfrev_b32_e32 %0, %0" : "+v"(b[i]));
            else if (PAT == 8) asm volatile("v_cmp_ne_u64_e32 vcc, %0, %1" : : "v"(a[i]), "v"(k64) : "vcc");
            else if (PAT == 9) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(b[i]) : "v"(c[i]));
            else if (PAT == 10) asm volatile("v_lshlrev_b32_e32 %0, 1, %0" : "+v"(b[i]));
            else if (PAT == 11) asm volatile("v_bfi_b32 %0, %0, %1, %2" : "+v"(b[i]) : "v"(c[i]), "v"(k32));
        }
    }
    unsigned long long s = 0;
    for (int i = 0; i < ILP; i++) s += a[i] + b[i] + c[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (seed == 12345) lds[threadIdx.x] = 1;
}

template <int PAT>
static void run(const char *name, int wgs_per_cu, int per)
{
    unsigned long long *d;
    const int blocks = 256 * wgs_per_cu;
    CHECK(hipMalloc(&d, sizeof(unsigned long long) * blocks * 256));
    const size_t lds = (160 * 1024) / wgs_per_cu - 512;
    CHECK(hipFuncSetAttribute((const void *)k<PAT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int w = 0; w < 2; w++) hipLaunchKernelGGL((k<PAT>), dim3(blocks), dim3(256), lds, 0, d, 77ull);
    CHECK(hipEventRecord(e0));
    const int reps = 10;
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL((k<PAT>), dim3(blocks), dim3(256), lds, 0, d, 77ull);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double per_simd_ns = (ms * 1e6 / reps) / ((double)ITER * ILP * per * wgs_per_cu);
    printf("%-30s waves/SIMD %d : %.3f ns per wave-instruction per SIMD\n", name, wgs_per_cu, per_simd_ns);
    CHECK(hipFree(d));
}

int main()
{
    const int o = 4;
    run<10>("v_lshlrev_b32 (reference)", o, 1); run<5>("v_or3_b32", o, 1); run<11>("v_bfi_b32", o, 1); run<4>("v_alignbit_b32", o, 1); run<7>("v_bfrev_b32", o, 1);
    run<0>("v_lshlrev_b64", o, 1); run<1>("v_lshrrev_b64", o, 1); run<2>("v_lshl_add_u64", o, 1); run<3>("v_add_co_u32 + v_addc_co_u32", o, 2);
    run<8>("v_cmp_ne_u64", o, 1); run<6>("v_mov_b32_dpp wave_shr:1", o, 1); run<9>("v_mov_b32_dpp row_shr:1", o, 1);
    return 0;
}
