// image_amd/csrc/fhog_device.h -- geometry and the orientation rule shared by fhog.hip (stage kernels) and
// fhog_fused.hip (the cell-size-8 batch path).
#pragma once
#include "common.h"

struct FhogGeom {
    int rows, cols, cs;
    int cells_nr, cells_nc;      // fhog.h:780-781
    int visible_nr, visible_nc;  // :817-818
    int body_end;                // columns 1 .. body_end-1 take the 8-wide path, the rest the scalar tail
    int hog_nr, hog_nc;          // interior cells (:806-807)
    int out_nr, out_nc;          // with filter padding (init_hog :455)
    int off_r, off_c;            // :813-814
};

// dlib's nine unit directions are 4-digit literals (fhog.h:766-775), not cos / sin
#define FHOG_DIR_TABLE                                                                                            \
    {{1.0000f, 0.0000f}, {0.9397f, 0.3420f}, {0.7660f, 0.6428f}, {0.500f, 0.8660f}, {0.1736f, 0.9848f},        \
     {-0.1736f, 0.9848f}, {-0.5000f, 0.8660f}, {-0.7660f, 0.6428f}, {-0.9397f, 0.3420f}}

// orientation bin 0..17 of the gradient (tx, ty): fhog.h:846-859 (8-wide body) and :929-943 (scalar tail) decide
// identically.  Float arithmetic in the reference's order, no contraction (library flag).
__device__ __forceinline__ int fhog_best_orientation(int tx, int ty)
{
    const float dirs[9][2] = FHOG_DIR_TABLE;
    const float fx = (float)tx, fy = (float)ty;
    float best_dot = 0;
    int best_o = 0;
#pragma unroll
    for (int o = 0; o < 9; o++) {
        const float dot = fx * dirs[o][0] + fy * dirs[o][1];
        if (dot > best_dot) { best_dot = dot; best_o = o; }
        else if (-dot > best_dot) { best_dot = -dot; best_o = o + 9; }
    }
    return best_o;
}

// fhog_fused.hip.  Magnitude and orientation of every possible gradient (tx, ty in -255..255) as a table of 511 x 512 packed
// words, filled on the device: lut[(ty + 255) * 512 + tx + 255].
#define FHOG_LUT_BYTES (511 * 512 * 4)
bool fhog_fused_supported(const FhogGeom &g, const uint8_t *d_rgb, size_t frame_stride);
imgfd_status fhog_fused_hist(imgfd_ctx *ctx, const uint8_t *d_rgb, size_t frame_stride, const FhogGeom &g, int nf, float *hist,
                             float *norm);
