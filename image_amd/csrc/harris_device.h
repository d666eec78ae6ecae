// image_amd/csrc/harris_device.h -- the corner strength function shared by the kernels that need a value of R:
// compute_corner_response(), image.CornerDetectionHarris/src/harris.cpp:78-133.  One definition, so the plane
// written by response_kernel, the values the fused response+NMS kernel compares and the strengths emitted with the
// corner list are the same bits (library built -ffp-contract=off: rounds where the reference's x86-64 build rounds).
#pragma once
#include "common.h"

template <int MEASURE>
__device__ __forceinline__ float harris_response_value(float a, float b, float c, float k)
{
    if (MEASURE == IMGFD_SHI_TOMASI_MEASURE) {
        // harris.cpp:112-115: float expression, float sqrt, then double arithmetic, float store
        const float D = sqrtf(a * a - 2 * a * c + 4 * b * b + c * c);
        return (float)(0.5 * (a + c) - 0.5 * D);
    } else if (MEASURE == IMGFD_HARMONIC_MEAN_MEASURE) {
        // harris.cpp:125-128: float det/trace, double divide
        const float detA = a * c - b * b;
        const float traceA = a + c;
        return (float)(2 * detA / (traceA + 0.0001));
    } else {
        // harris.cpp:100-103
        const float detA = a * c - b * b;
        const float traceA = a + c;
        return detA - k * traceA * traceA;
    }
}

// The threshold quad of four horizontally adjacent responses: bit e = "v[e] is not below the threshold (skip = R < Th,
// harris.cpp:160-162) and not beaten by its neighbour inside the quad" -- the two comparisons of the window rule's 3x3
// pre-test (nms.hip) that need no pixel outside the quad.  Written by fir_tensor's response epilogue, read by
// harris_nms_sparse.
__device__ __forceinline__ unsigned harris_quad_bits(float v0, float v1, float v2, float v3, float Th)
{
    const bool t0 = !(v0 < Th) && !(v1 >= v0);
    const bool t1 = !(v1 < Th) && !(v2 >= v1) && !(v0 > v1);
    const bool t2 = !(v2 < Th) && !(v3 >= v2) && !(v1 > v2);
    const bool t3 = !(v3 < Th) && !(v2 > v3);
    return (t0 ? 1u : 0u) | (t1 ? 2u : 0u) | (t2 ? 4u : 0u) | (t3 ? 8u : 0u);
}
