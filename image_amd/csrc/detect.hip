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

namespace {

imgfd_status detect_body(imgfd_ctx *ctx, const imgfd_frames *fr, const imgfd_stream_params *p, imgfd_corner *d_corners,
                         imgfd_point *d_points, uint8_t *d_edges, int64_t *d_counts)
{
    const int B = fr->n_frames;
    if (!p->canny) {
        if (p->fast9) IMGFD_TRY(imgfd_fast9_dev(ctx, fr, (uint8_t)p->fast9_threshold, p->suppress_non_max, d_points, p->point_cap, d_counts + B));
        if (p->harris) IMGFD_TRY(imgfd_harris_dev(ctx, fr, p->k, p->sigma_d, p->sigma_i, p->threshold, p->gaussian, p->gradient, p->measure, d_corners, p->corner_cap, d_counts));
        return IMGFD_OK;
    }
    if (!p->harris && !p->fast9) return imgfd_canny_dev(ctx, fr, p->s, p->low_thr, p->high_thr, p->accGrad, d_edges, d_counts + 2 * B);
    imgfd_ctx *side = nullptr;
    IMGFD_TRY(ctx_side(ctx, &side));
    // Two streams: one runs Canny's chain, the other FAST-9 and the Harris chain.  For batches Canny takes the companion's stream; below
    // eight frames the context's own: then nothing stands between the caller's previous work on that stream and the blur -- no
    // fork event to cross -- and the join at the end is a wait that has usually been satisfied already (round 5, LOG.md).
    const bool swap = B < 8;
    imgfd_ctx *cc = swap ? ctx : side, *oc = swap ? side : ctx;   // Canny's context / the other detectors'
    auto fail_from = [&](imgfd_ctx *c, imgfd_status st) -> imgfd_status {
        if (c != ctx && !c->err.empty()) ctx->err = c->err;
        return st;
    };
    auto fast9 = [&]() -> imgfd_status {
        if (!p->fast9) return IMGFD_OK;
        const imgfd_status st = imgfd_fast9_dev(oc, fr, (uint8_t)p->fast9_threshold, p->suppress_non_max, d_points, p->point_cap, d_counts + B);
        return st == IMGFD_OK ? st : fail_from(oc, st);
    };
    auto harris = [&]() -> imgfd_status {
        if (!p->harris) return IMGFD_OK;
        const imgfd_status st = imgfd_harris_dev(oc, fr, p->k, p->sigma_d, p->sigma_i, p->threshold, p->gaussian, p->gradient, p->measure, d_corners,
                                                 p->corner_cap, d_counts);
        return st == IMGFD_OK ? st : fail_from(oc, st);
    };
    // the frames (and anything else queued on the context's stream) come first
    IMGFD_HIP(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
    IMGFD_HIP(ctx, hipStreamWaitEvent(side->stream, ctx->ev_fork, 0));
    // FAST-9 starts with Canny's blur.  The Harris chain is held back until Canny's gradient/NMS kernel has finished: its
    // Gaussian/gradient kernel then runs beside the hysteresis sweeps (latency-bound, a fraction of the chip), and the
    // structure-tensor kernel -- whose workgroups fill every CU they sit on -- does not stretch Canny's VALU-bound front.
    // (Round 2 got this order by accident: the 16-wave rows_scan workgroup of FAST-9's compaction found no CU with 16 free
    // wave slots until the gradient/NMS kernel had drained.  Released together with FAST-9: 43.5 instead of 40.3 ms per
    // 10 passes of 32 4K frames, profiles/r03/experiments_log.txt.)
    const int fast_at = 0;
    // (small batches: behind the blur already -- the chain then ends before Canny's does, and the join below is a wait that has been
    // satisfied by the time the stream reaches it: 182 against 190 us for a single 4K frame, profiles/r05/single_frame_variants.txt;
    // the other release points measured: profiles/r06/single_frame_gates.txt)
    const int harris_at = B < 8 ? 1 : 2;
    // Small batches are bound by Canny's chain of dependent kernels, and the HOST queues launches at 3-8 us each: with the other
    // detectors queued from inside the hook, a single 4K frame's first hysteresis sweep reached its queue 39 us after
    // gradient/NMS had finished (profiles/r04/f_single_frame_timeline.txt: the ten launches of FAST-9 and the Harris chain sat in
    // between).  Below eight frames the hook therefore only RECORDS the release events where they belong in Canny's stream; the waits
    // and the other detectors' launches are queued after the last Canny launch.  Same dependencies on the device, the critical
    // chain first on the host.
    const bool defer = B < 8;
    bool fast_due = false, harris_due = false;
    const std::function<imgfd_status(int)> hook = [&](int pos) -> imgfd_status {  // pos: 0 before Canny's blur, 1 behind it, 2 behind gradient/NMS
        if (pos == fast_at) {
            IMGFD_HIP(ctx, hipEventRecord(ctx->ev_gate, cc->stream));
            if (defer) fast_due = true;
            else {
                IMGFD_HIP(ctx, hipStreamWaitEvent(oc->stream, ctx->ev_gate, 0));
                IMGFD_TRY(fast9());
            }
        }
        if (pos == harris_at) {
            IMGFD_HIP(ctx, hipEventRecord(ctx->ev_gate2, cc->stream));
            if (defer) harris_due = true;
            else {
                IMGFD_HIP(ctx, hipStreamWaitEvent(oc->stream, ctx->ev_gate2, 0));
                IMGFD_TRY(harris());
            }
        }
        return IMGFD_OK;
    };
    const imgfd_status st = canny_dev_hooked(cc, fr, p->s, p->low_thr, p->high_thr, p->accGrad, d_edges, d_counts + 2 * B, &hook);
    if (st != IMGFD_OK) return fail_from(cc, st);
    if (fast_due) {
        IMGFD_HIP(ctx, hipStreamWaitEvent(oc->stream, ctx->ev_gate, 0));
        IMGFD_TRY(fast9());
    }
    if (harris_due) {
        IMGFD_HIP(ctx, hipStreamWaitEvent(oc->stream, ctx->ev_gate2, 0));
        IMGFD_TRY(harris());
    }
    // whoever waits for the context's stream waits for the companion's work too
    IMGFD_HIP(ctx, hipEventRecord(ctx->ev_join, side->stream));
    IMGFD_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
    return IMGFD_OK;
}

}  // namespace

extern "C" {

imgfd_status imgfd_detect_dev(imgfd_ctx *ctx, const imgfd_frames *fr, const imgfd_stream_params *p, imgfd_corner *d_corners,
                              imgfd_point *d_points, uint8_t *d_edges, int64_t *d_counts)
try {
    if (!ctx) return IMGFD_ERR_INVALID;
    if (!fr || !p || !d_counts || fr->n_frames < 0 || (!p->harris && !p->fast9 && !p->canny) || (p->harris && p->corner_cap < 0) ||
        (p->fast9 && p->point_cap < 0) || (p->canny && !d_edges) || p->fast9_threshold < 0 || p->fast9_threshold > 255)
        return imgfd_fail(ctx, IMGFD_ERR_INVALID, "imgfd_detect_dev: bad argument");
    IMGFD_HIP(ctx, hipSetDevice(ctx->device));
    // (A recorded hipGraph of the launch sequence of a repeating small-batch call replayed in the same time as the eager launches,
    // rounds 3 and 5: a single frame is bound by the device's chain of dependent kernels, not by the host.  The capture path left the
    // library in round 6: scripts/experiments/r06_pruned_switches.patch.)
    return detect_body(ctx, fr, p, d_corners, d_points, d_edges, d_counts);
} catch (const std::bad_alloc &) {
    return imgfd_fail(ctx, IMGFD_ERR_OOM, "imgfd_detect_dev: out of host memory");
} catch (...) {
    return imgfd_fail(ctx, IMGFD_ERR_HIP, "imgfd_detect_dev: unexpected C++ exception");
}

}  // extern "C"
