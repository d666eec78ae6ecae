// image_amd/csrc/fir_tensor_device.h -- device helpers of the structure-tensor kernel (fir_tensor.hip): launch parameters,
// buffer-addressed stores, the interleaved tap chains.  (Also included by the measured-and-rejected wave-autonomous variant,
// scripts/experiments/fir_tensor_wave.hip, which is not part of the library.)
// compute_autocorrelation_matrix(), image.CornerDetectionHarris/src/harris.cpp:44-70; gaussian.cpp:289-395.
#pragma once
#include "common.h"
#include "fir_device.h"
#include "harris_device.h"

typedef float ft_v4f __attribute__((vector_size(16)));  // native vector: always promoted to registers
// a volatile 16-byte read that is known to address LDS (a volatile access through a generic pointer becomes a flat load)
typedef volatile IMGFD_LDS_SPACE ft_v4f ft_lds_v4f;

struct TensorParams {
    const float *ix;
    const float *iy;
    float *out0, *out1, *out2;  // A, B, C -- or R in out0 (OUT = 2)
    int nx, ny;
    long frame_stride;  // elements between frames (planes are packed: pitch nx)
    int nstrips, n_frames;
    long total_units, units_per_worker;  // the batch as one line of chunk units (columns strip fastest), a worker's share of it
    int units_per_column;  // chunks of one (frame, strip) column marched in one piece
    int xcd_remap;
    float k;            // Harris constant (OUT = 2)
    // OUT = 2, optional: one byte per quad of pixels, bit e = "the response of pixel x + e is not below the threshold
    // (harris.cpp:160-162: skip = R < Th) nor beaten by a neighbour inside the quad", quad (frame, y, x / 4) at tq[(frame * ny + y) * (nx / 4) + x / 4]: what the
    // sparse NMS kernel starts from.  (A byte per lane: no cross-lane traffic in the output phase, which every wave of the
    // workgroup waits for.  Assembling 64-bit mask words here -- four ballots and a bit spread per row -- cost 14 %.)
    unsigned char *tq;
    float Th;
    double B[8];        // taps B[0..R], R <= 7
};

// a float plane addressed as a hardware buffer: store(value) at byte offset lane_off (per lane) + row_off (wave-uniform)
struct FtBuffer { __amdgpu_buffer_rsrc_t r; };
__device__ __forceinline__ FtBuffer ft_make_buffer(float *base, unsigned bytes)
{
    return FtBuffer{__builtin_amdgcn_make_buffer_rsrc(base, 0, (int)bytes, 0x00027000)};  // raw buffer, 32-bit data format
}
__device__ __forceinline__ void ft_buffer_store(const FtBuffer &b, unsigned lane_off, unsigned row_off, float v)
{
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), b.r, (int)lane_off, (int)row_off, 0);
}

// N outputs of one 1-D pass over a register window: out[o] = B[0]*d[o+R] + sum_j B[j]*(d[o+R-j] + d[o+R+j]), pair added
// first, j ascending, in double, one rounding to float (gaussian.cpp:351-359).  One output is a chain of 2R+1 DEPENDENT
// f64 instructions (15.4 cycles each when issued back to back, profiles/r01/ubench2.txt): ILP outputs advance together,
// tap by tap, so that the chains of a wave cover each other's latency.  Each chain keeps the reference's own order.
#ifndef FT_ILP
#define FT_ILP 4
#endif
template <int R, bool FMA, int N, int ILP>
__device__ __forceinline__ void ft_taps(const double (&d)[N + 2 * R], const double *B, float (&out)[N])
{
    static_assert(N % ILP == 0, "groups of ILP outputs");
#pragma unroll
    for (int o0 = 0; o0 < N; o0 += ILP) {
        double sum[ILP];
#pragma unroll
        for (int g = 0; g < ILP; g++) sum[g] = B[0] * d[o0 + g + R];
#pragma unroll
        for (int j = 1; j <= R; j++) {
            double pair[ILP];
#pragma unroll
            for (int g = 0; g < ILP; g++) pair[g] = d[o0 + g + R - j] + d[o0 + g + R + j];
#pragma unroll
            for (int g = 0; g < ILP; g++) {
                if (FMA) sum[g] = __builtin_fma(B[j], pair[g], sum[g]);
                else sum[g] += B[j] * pair[g];
            }
        }
#pragma unroll
        for (int g = 0; g < ILP; g++) out[o0 + g] = (float)sum[g];
    }
}

// ILP outputs o0 .. o0+ILP-1 of one 1-D pass from the float window w (converted to double on first use: dw[] is the
// same window in double, filled up to index `have`): the reference's sum, chains interleaved tap by tap.
// PRE: the first PRE entries of dw[] are valid on entry (a column history kept in double).
template <int R, bool FMA, int ILP, int NWIN, int PRE = 0>
__device__ __forceinline__ void ft_group(const float (&w)[NWIN], double (&dw)[NWIN], int o0, const double *B, float (&out)[ILP])
{
    // the group needs dw[o0 .. o0 + ILP + 2R); everything below o0 + 2R was converted by the previous groups
#pragma unroll
    for (int k = (o0 == 0 ? PRE : o0 + 2 * R); k < o0 + ILP + 2 * R; k++) dw[k] = (double)w[k];
    double sum[ILP];
#pragma unroll
    for (int g = 0; g < ILP; g++) sum[g] = B[0] * dw[o0 + g + R];
#pragma unroll
    for (int j = 1; j <= R; j++) {
        double pair[ILP];
#pragma unroll
        for (int g = 0; g < ILP; g++) pair[g] = dw[o0 + g + R - j] + dw[o0 + g + R + j];
#pragma unroll
        for (int g = 0; g < ILP; g++) {
            if (FMA) sum[g] = __builtin_fma(B[j], pair[g], sum[g]);
            else sum[g] += B[j] * pair[g];
        }
    }
#pragma unroll
    for (int g = 0; g < ILP; g++) out[g] = (float)sum[g];
}
