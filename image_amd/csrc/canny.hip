// image_amd/csrc/canny.hip -- Canny edge detector (K9-K12).  Placeholder until the kernels land.
#include "common.h"

extern "C" {
imgfd_status imgfd_canny(imgfd_ctx *ctx, const uint8_t *, int, int, double, double, double, int, uint8_t *, int64_t *)
{
    return imgfd_fail(ctx, IMGFD_ERR_UNSUPPORTED, "imgfd_canny: not implemented yet");
}
imgfd_status imgfd_canny_dev(imgfd_ctx *ctx, const imgfd_frames *, double, double, double, int, uint8_t *, int64_t *)
{
    return imgfd_fail(ctx, IMGFD_ERR_UNSUPPORTED, "imgfd_canny_dev: not implemented yet");
}
}
