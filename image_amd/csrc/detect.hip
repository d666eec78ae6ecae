// image_amd/csrc/detect.hip -- Harris + FAST-9 + Canny on one device-resident batch, overlapped on two HIP streams.
//
// Host-side scheduling only; the kernels are those of imgfd_harris_dev / imgfd_fast9_dev / imgfd_canny_dev and every
// result is what those calls return.  Why overlap: every large kernel of the three chains is bound by VALU issue and
// sustains 0.66-0.88 of it on its own (profiles/r02/*pmc_all_kernels.txt); Canny's hysteresis (rcpp_canny.cpp:184-215) is
// a fixpoint iteration whose later sweeps touch a handful of tiles -- a few waves on a 256-CU device.  Two kernels side by
// side fill each other's issue gaps.  Canny runs on the context's companion stream, FAST-9 and the Harris chain on the
// context's own stream, both from the start of the batch (measured on 32 x 4K frames, scripts/gpu_gate.sh: 61.0 Gpixel/s;
// releasing the second stream only after Canny's blur + gradient/NMS: 59.1; Harris before FAST-9: 57.3).
//
//      companion stream :  blur | grad+NMS | hysteresis sweeps ...... | expand, count |
//      context stream   :  FAST-9 | gauss+grad | structure tensor + response | NMS | compaction |
#include "common.h"

extern "C" {

imgfd_status imgfd_detect_dev(imgfd_ctx *ctx, const imgfd_frames *fr, const imgfd_stream_params *p, imgfd_corner *d_corners,
                              imgfd_point *d_points, uint8_t *d_edges, int64_t *d_counts)
try {
    if (!ctx) return IMGFD_ERR_INVALID;
    if (!fr || !p || !d_counts || fr->n_frames < 0 || (!p->harris && !p->fast9 && !p->canny) || (p->harris && p->corner_cap < 0) ||
        (p->fast9 && p->point_cap < 0) || (p->canny && !d_edges) || p->fast9_threshold < 0 || p->fast9_threshold > 255)
        return imgfd_fail(ctx, IMGFD_ERR_INVALID, "imgfd_detect_dev: bad argument");
    IMGFD_HIP(ctx, hipSetDevice(ctx->device));
    const int B = fr->n_frames;
    auto fast9 = [&]() -> imgfd_status {
        if (!p->fast9) return IMGFD_OK;
        return imgfd_fast9_dev(ctx, fr, (uint8_t)p->fast9_threshold, p->suppress_non_max, d_points, p->point_cap, d_counts + B);
    };
    auto harris = [&]() -> imgfd_status {
        if (!p->harris) return IMGFD_OK;
        return imgfd_harris_dev(ctx, fr, p->k, p->sigma_d, p->sigma_i, p->threshold, p->gaussian, p->gradient, p->measure, d_corners,
                                p->corner_cap, d_counts);
    };
    if (!p->canny) {
        IMGFD_TRY(fast9());
        return harris();
    }
    if (!p->harris && !p->fast9) return imgfd_canny_dev(ctx, fr, p->s, p->low_thr, p->high_thr, p->accGrad, d_edges, d_counts + 2 * B);
    imgfd_ctx *side = nullptr;
    IMGFD_TRY(ctx_side(ctx, &side));
    // the frames (and anything else queued on the context's stream) come first
    IMGFD_HIP(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
    IMGFD_HIP(ctx, hipStreamWaitEvent(side->stream, ctx->ev_fork, 0));
    const std::function<imgfd_status()> gate = [&]() -> imgfd_status {
        IMGFD_HIP(ctx, hipEventRecord(ctx->ev_gate, side->stream));
        IMGFD_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_gate, 0));
        IMGFD_TRY(fast9());
        return harris();
    };
    const imgfd_status st = canny_dev_hooked(side, fr, p->s, p->low_thr, p->high_thr, p->accGrad, d_edges, d_counts + 2 * B, &gate);
    if (st != IMGFD_OK) {
        if (ctx->err.empty() || !side->err.empty()) ctx->err = side->err.empty() ? ctx->err : side->err;
        return st;
    }
    // whoever waits for the context's stream waits for the edges too
    IMGFD_HIP(ctx, hipEventRecord(ctx->ev_join, side->stream));
    IMGFD_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
    return IMGFD_OK;
} catch (const std::bad_alloc &) {
    return imgfd_fail(ctx, IMGFD_ERR_OOM, "imgfd_detect_dev: out of host memory");
} catch (...) {
    return imgfd_fail(ctx, IMGFD_ERR_HIP, "imgfd_detect_dev: unexpected C++ exception");
}

}  // extern "C"
