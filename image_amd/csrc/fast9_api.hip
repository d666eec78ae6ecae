// image_amd/csrc/fast9_api.hip -- FAST-9 behind the C ABI (host-pointer and device-resident entry points).
// Mirrors detect_corners(), image.CornerDetectionF9/src/f9_rcpp.cpp:8-35, i.e. F9::detectCorners
// (f9.cpp:5730 -> Impl::detectCorners :66-82).
#include "common.h"

#include <stdlib.h>

#include <algorithm>

static imgfd_status fast9_host(imgfd_ctx *ctx, const void *img, int kind, int width, int height, int bytes_per_row,
                               uint8_t threshold, int suppress_non_max, imgfd_points *out)
{
    if (!ctx || !out) return IMGFD_ERR_INVALID;
    out->points = nullptr;
    out->n = 0;
    if (!img || width < 0 || height < 0 || bytes_per_row < width || !frame_fits(bytes_per_row, height))
        return imgfd_fail(ctx, IMGFD_ERR_INVALID, "imgfd_fast9: bad image geometry");
    if (width < 7 || height < 7) return IMGFD_OK;  // empty search domain, f9.cpp:2959-2960
    IMGFD_HIP(ctx, hipSetDevice(ctx->device));
    const size_t img_bytes = (size_t)bytes_per_row * height;
    const int64_t cap = (int64_t)(width - 6) * (height - 6);
    const size_t need = align_up(img_bytes, 256) + compact_bytes(width, height, 1) +
                        align_up(sizeof(imgfd_point) * (size_t)cap, 256) + upload_stage_bytes(kind, img_bytes) + 4096;
    IMGFD_TRY(ws_reserve(ctx, need));
    uint8_t *d_img = (uint8_t *)ws_alloc(ctx, img_bytes);
    CompactBuffers cb;
    IMGFD_TRY(compact_carve(ctx, width, height, 1, &cb));
    imgfd_point *d_points = (imgfd_point *)ws_alloc(ctx, sizeof(imgfd_point) * (size_t)cap);
    int64_t *d_count = (int64_t *)ws_alloc(ctx, sizeof(int64_t));
    if (!d_img || !d_points || !d_count) return imgfd_fail(ctx, IMGFD_ERR_OOM, "workspace reservation too small");
    IMGFD_TRY(upload_image(ctx, img, kind, img_bytes, d_img));
    IMGFD_TRY(compact_clear(ctx, cb, height, 1));
    IMGFD_TRY(launch_fast9(ctx, d_img, width, height, bytes_per_row, img_bytes, 1, threshold, suppress_non_max, cb));
    IMGFD_TRY(compact_emit(ctx, cb, width, height, 1, 1, nullptr, d_points, cap, d_count));
    int64_t n = 0;
    IMGFD_HIP(ctx, hipMemcpyAsync(&n, d_count, sizeof n, hipMemcpyDeviceToHost, ctx->stream));
    IMGFD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (n > cap) n = cap;
    out->n = n;
    if (n) {
        out->points = (imgfd_point *)malloc(sizeof(imgfd_point) * (size_t)n);
        if (!out->points) return imgfd_fail(ctx, IMGFD_ERR_OOM, "malloc of the point list failed");
        IMGFD_HIP(ctx, hipMemcpyAsync(out->points, d_points, sizeof(imgfd_point) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
        IMGFD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return IMGFD_OK;
}

extern "C" {

imgfd_status imgfd_fast9(imgfd_ctx *ctx, const uint8_t *img, int width, int height, int bytes_per_row,
                         uint8_t threshold, int suppress_non_max, imgfd_points *out)
{
    return imgfd_guard(ctx, [&] { return fast9_host(ctx, img, IMGFD_SRC_U8, width, height, bytes_per_row, threshold, suppress_non_max, out); });
}

imgfd_status imgfd_fast9_i32(imgfd_ctx *ctx, const int32_t *x, int width, int height, int bytes_per_row,
                             uint8_t threshold, int suppress_non_max, imgfd_points *out)
{
    return imgfd_guard(ctx, [&] { return fast9_host(ctx, x, IMGFD_SRC_I32, width, height, bytes_per_row, threshold, suppress_non_max, out); });
}

imgfd_status imgfd_fast9_dev(imgfd_ctx *ctx, const imgfd_frames *fr, uint8_t threshold,
                             int suppress_non_max, imgfd_point *d_points, int64_t cap, int64_t *d_counts)
{
    if (!ctx || !fr || !fr->d_frames || (!d_points && cap > 0) || !d_counts || cap < 0 || fr->n_frames < 0 || fr->dtype != 0)
        return imgfd_fail(ctx, IMGFD_ERR_INVALID, "imgfd_fast9_dev: bad argument (frames must be u8)");
    if (fr->nx < 1 || fr->ny < 1 || fr->row_stride_bytes < fr->nx || !frame_fits(fr->nx, fr->ny))
        return imgfd_fail(ctx, IMGFD_ERR_INVALID, "imgfd_fast9_dev: bad frame geometry");
    if (fr->n_frames == 0) return IMGFD_OK;
    IMGFD_HIP(ctx, hipSetDevice(ctx->device));
    const int w = fr->nx, h = fr->ny;
    if (w < 7 || h < 7) {  // empty search domain (f9.cpp:2959-2960), as the host entry point answers
        IMGFD_HIP(ctx, hipMemsetAsync(d_counts, 0, sizeof(int64_t) * fr->n_frames, ctx->stream));
        return IMGFD_OK;
    }
    const size_t per_frame = compact_bytes(w, h, 1);
    const int chunk = sub_batch_frames(ctx, fr->n_frames, per_frame, (size_t)1 << 30);
    IMGFD_TRY(ws_reserve(ctx, compact_bytes(w, h, chunk) + 4096));
    CompactBuffers cb;
    IMGFD_TRY(compact_carve(ctx, w, h, chunk, &cb));
    for (int f0 = 0; f0 < fr->n_frames; f0 += chunk) {
        const int nf = std::min(chunk, fr->n_frames - f0);
        const uint8_t *base = (const uint8_t *)fr->d_frames + (size_t)f0 * fr->frame_stride_bytes;
        IMGFD_TRY(compact_clear(ctx, cb, h, nf));
        IMGFD_TRY(launch_fast9(ctx, base, w, h, fr->row_stride_bytes, fr->frame_stride_bytes, nf, threshold,
                               suppress_non_max, cb));
        IMGFD_TRY(compact_emit(ctx, cb, w, h, nf, 1, nullptr, d_points + (size_t)f0 * cap, cap, d_counts + f0));
    }
    return IMGFD_OK;
}

}  // extern "C"
