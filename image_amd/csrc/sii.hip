// image_amd/csrc/sii.hip -- stacked-integral-images "fast Gaussian" (K6), gaussian code 1.
#include "common.h"

imgfd_status launch_sii_gaussian(imgfd_ctx *ctx, const float *d_in, float *d_out, int nx, int ny,
                                 int n_frames, float sigma)
{
    (void)d_in; (void)d_out; (void)nx; (void)ny; (void)n_frames; (void)sigma;
    return imgfd_fail(ctx, IMGFD_ERR_UNSUPPORTED, "SII fast Gaussian (gaussian code 1) is not implemented yet");
}
