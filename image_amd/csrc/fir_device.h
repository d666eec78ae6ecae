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
#pragma unroll
    for (int o = 0; o < FIR_PX; o++) {
        double sum = B[0] * d[o + R];
#pragma unroll
        for (int j = 1; j <= R; j++) {
            double pair = d[o + R - j] + d[o + R + j];
            if (FMA) sum = __builtin_fma(B[j], pair, sum);
            else sum += B[j] * pair;
        }
        out[o] = (float)sum;
    }
}

