// scripts/ubench/sqrt_check.hip -- is the bare v_sqrt_f32 instruction correctly rounded on every input the fHOG histogram
// pass can feed it?  (float)(n) for the squared gradient lengths n = 0 .. 2*255^2 = 130050 (fhog.h:835-845 computes
// sqrt of such an integer per pixel).  Compares with the IEEE sqrtf sequence the compiler emits by default.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
__global__ void k(unsigned *bad, unsigned *first)
{
    const unsigned n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n > 130050u) return;
    const float x = (float)n;
    const float fast = __builtin_amdgcn_sqrtf(x);
    const float ieee = sqrtf(x);
    if (__float_as_uint(fast) != __float_as_uint(ieee)) { atomicAdd(bad, 1u); atomicMin(first, n); }
}
int main()
{
    unsigned *d, h[2] = {0u, 0xffffffffu};
    hipMalloc(&d, 8);
    hipMemcpy(d, h, 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3((130051 + 255) / 256), dim3(256), 0, 0, d, d + 1);
    hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    printf("v_sqrt_f32 vs IEEE sqrtf on n = 0..130050: %u mismatches (first at n = %u)\n", h[0], h[1]);
    return 0;
}
