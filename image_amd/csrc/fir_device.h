// image_amd/csrc/fir_device.h -- device helpers shared by the double-accumulated Gaussian FIR kernels (fir.hip,
// gauss_grad.hip): the reference's border rule and its 1-D pass, image.CornerDetectionHarris/src/gaussian.cpp:289-395.
#pragma once
#include "common.h"

// left/top: whole-sample reflection (-k -> k); right/bottom: half-sample (n-1+k -> n-k)
__device__ __forceinline__ int fir_reflect(int i, int n)
{
    const int lo = -i, hi = 2 * n - 1 - i;
    i = i < 0 ? lo : (i >= n ? hi : i);
    return min(max(i, 0), n - 1);
}

// One 1-D pass over a register window: out[o] = B[0]*d[o+R] + sum_j B[j]*(d[o+R-j] + d[o+R+j]), pair added first, j
// ascending, in double, one rounding to float (gaussian.cpp:351-359).  FMA = false issues exactly that sequence;
// FMA = true fuses only the accumulate (sum = fma(B[j], pair, sum)).
template <int R, bool FMA, int FIR_PX>
__device__ __forceinline__ void fir_window8(const double (&d)[FIR_PX + 2 * R], const double *B,
                                            float (&out)[FIR_PX])
{
#ifndef FIR_ILP
#define FIR_ILP 1
#endif
    // FIR_ILP independent output chains advance together (each chain keeps the reference's own operation order)
#pragma unroll
    for (int o0 = 0; o0 < FIR_PX; o0 += FIR_ILP) {
        double sum[FIR_ILP];
#pragma unroll
        for (int g = 0; g < FIR_ILP; g++) sum[g] = B[0] * d[o0 + g + R];
#pragma unroll
        for (int j = 1; j <= R; j++) {
            double pair[FIR_ILP];
#pragma unroll
            for (int g = 0; g < FIR_ILP; g++) pair[g] = d[o0 + g + R - j] + d[o0 + g + R + j];
#pragma unroll
            for (int g = 0; g < FIR_ILP; g++) {
                if (FMA) sum[g] = __builtin_fma(B[j], pair[g], sum[g]);
                else sum[g] += B[j] * pair[g];
            }
        }
#pragma unroll
        for (int g = 0; g < FIR_ILP; g++) out[o0 + g] = (float)sum[g];
    }
}

