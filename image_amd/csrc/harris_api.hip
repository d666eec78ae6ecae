// image_amd/csrc/harris_api.hip -- host orchestration of the Harris path behind the C ABI.
//
// Mirrors detect_corners() (image.CornerDetectionHarris/src/rcpp_harris.cpp:19-60), harris_scale()
// (harris.cpp:554-608) and harris() (:473-546): the per-pixel stages run on the device (fir.hip,
// harris_stages.hip, nms.hip, compact.hip), and so does the sub-pixel fit of the compacted corner list
// (compute_subpixel_precision :340-381 with interpolation.cpp -> harris_subpixel.hip); what stays on the host is the
// ranking / selection of a few thousand records (select_output_corners :263-332, select_corners :443-465), because the
// reference's order among equal strengths is std::sort's.
#include "common.h"

#include <math.h>
#include <stdlib.h>

#include <algorithm>
#include <chrono>
#include <vector>

// zoom_out(): zoom.cpp:121-139.  bicubic_interpolation_at() is evaluated at integer coordinates only
// (uu-x == 0), where cubic_interpolation() returns v[1] exactly: a 2x decimation.
__global__ void __launch_bounds__(256) zoom_out_kernel(const float *__restrict__ I, float *__restrict__ Iz, int nx,
                                                       int nxx, int nyy)
{
    const int j1 = blockIdx.x * blockDim.x + threadIdx.x;
    const int i1 = blockIdx.y;
    if (j1 < nxx && i1 < nyy) Iz[(size_t)i1 * nxx + j1] = I[(size_t)(2 * i1) * nx + 2 * j1];
}

namespace {

// ---- the per-corner host stages: ranking and selection (select_output_corners(), harris.cpp:263-332; the scale check
// select_corners(), :443-465).  They run over the few thousand records the device stages leave, on the host, because the
// reference's own order among corners of EQUAL strength is whatever std::sort leaves, and only the same std::sort over the
// same sequence with the same "stronger first" comparison (harris.cpp:29-36) reproduces it.  Everything here works on
// indices into the raster-ordered list; the records themselves (and their sub-pixel refinements, made on the device:
// harris_subpixel.hip) are only looked up at the end.
using CornerIdx = uint32_t;

struct StrongerFirst {
    const imgfd_corner *c;
    bool operator()(CornerIdx a, CornerIdx b) const { return c[a].R > c[b].R; }
};

// ranks [first, last) strongest first and keeps at most `limit` of them (limit < 0: all); returns the new end
CornerIdx *rank_and_cut(CornerIdx *first, CornerIdx *last, const imgfd_corner *c, long limit)
{
    std::sort(first, last, StrongerFirst{c});
    if (limit >= 0 && last - first > limit) last = first + limit;
    return last;
}

// The corners to report, as indices into c[0..n), in output order.
//   ALL_CORNERS: raster order.  ALL_CORNERS_SORTED: by strength.  N_CORNERS: the N strongest.
//   DISTRIBUTED_N_CORNERS: the image is cut into cells x cells boxes, every box contributes its N / cells^2 strongest
//   (at least one), and the union is ranked and cut to N.
std::vector<CornerIdx> select_output(const std::vector<imgfd_corner> &c, int strategy, int cells, int N, int nx, int ny)
{
    const size_t n = c.size();
    std::vector<CornerIdx> keep(n);
    for (size_t i = 0; i < n; i++) keep[i] = (CornerIdx)i;
    const long limit = N < 0 ? 0 : N;
    if (strategy == IMGFD_ALL_CORNERS_SORTED) {
        rank_and_cut(keep.data(), keep.data() + n, c.data(), -1);
    } else if (strategy == IMGFD_N_CORNERS) {
        keep.resize((size_t)(rank_and_cut(keep.data(), keep.data() + n, c.data(), limit) - keep.data()));
    } else if (strategy == IMGFD_DISTRIBUTED_N_CORNERS) {
        const int bx = std::max(1, std::min(cells, nx)), by = std::max(1, std::min(cells, ny));
        const int boxes = bx * by;
        const long quota = std::max(1, N / boxes);
        const float box_w = (float)nx / bx, box_h = (float)ny / by;  // float division and truncation, as the reference bins
        // a stable counting sort by box: inside a box the corners stay in raster order, the sequence the reference sorts
        std::vector<int> box_of(n);
        std::vector<size_t> begin((size_t)boxes + 1, 0);
        for (size_t i = 0; i < n; i++) {
            const int col = (int)(c[i].x / box_w), row = (int)(c[i].y / box_h);
            box_of[i] = std::max(0, std::min(boxes - 1, row * bx + col));
            begin[(size_t)box_of[i] + 1]++;
        }
        for (int b = 0; b < boxes; b++) begin[(size_t)b + 1] += begin[b];
        std::vector<CornerIdx> by_box(n);
        {
            std::vector<size_t> fill(begin.begin(), begin.end() - 1);
            for (size_t i = 0; i < n; i++) by_box[fill[(size_t)box_of[i]]++] = (CornerIdx)i;
        }
        keep.clear();
        for (int b = 0; b < boxes; b++) {
            CornerIdx *first = by_box.data() + begin[b];
            keep.insert(keep.end(), first, rank_and_cut(first, by_box.data() + begin[(size_t)b + 1], c.data(), quota));
        }
        keep.resize((size_t)(rank_and_cut(keep.data(), keep.data() + keep.size(), c.data(), limit) - keep.data()));
    }
    return keep;
}

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct HarrisArgs {
    float k, sigma_d, sigma_i, Th;
    int gauss, grad, measure, strategy, cells, N, precision, verbose;
};

// floats per frame of the scratch plane `tmp`: one image plane, or the padded cumulative-sum plane of the SII Gaussian
size_t harris_tmp_floats(int nx, int ny, float sigma_d, float sigma_i, int gauss)
{
    size_t n = (size_t)nx * ny;
    if (gauss != IMGFD_STD_GAUSSIAN) {  // code 1: SII everywhere; code 2: SII for the structure tensor (harris.cpp:64-65)
        n = std::max(n, sii_scratch_floats(nx, ny, sigma_i));
        if (gauss == IMGFD_FAST_GAUSSIAN) n = std::max(n, sii_scratch_floats(nx, ny, sigma_d));
    }
    return n;
}

size_t harris_ws_bytes(int nx, int ny, int n_frames, int64_t cap, size_t tmp_floats)
{
    const size_t plane = align_up(sizeof(float) * (size_t)nx * ny * n_frames, 256);
    return 7 * plane + align_up(sizeof(float) * tmp_floats * n_frames, 256) + compact_bytes(nx, ny, n_frames) + align_up(sizeof(imgfd_corner) * (size_t)cap * n_frames, 256) +
           align_up(sizeof(imgfd_corner) * (size_t)cap * n_frames, 256) + align_up(sizeof(int64_t) * n_frames, 256) + 4096;
}

struct HarrisPlanes {
    float *Is, *Ix, *Iy, *A, *B, *C, *R, *tmp;
    CompactBuffers cb;
};

imgfd_status carve_planes(imgfd_ctx *ctx, int nx, int ny, int n_frames, size_t tmp_floats, HarrisPlanes *hp)
{
    const size_t bytes = sizeof(float) * (size_t)nx * ny * n_frames;
    float **pl[7] = {&hp->Is, &hp->Ix, &hp->Iy, &hp->A, &hp->B, &hp->C, &hp->R};
    for (auto p : pl) {
        *p = (float *)ws_alloc(ctx, bytes);
        if (!*p) return imgfd_fail(ctx, IMGFD_ERR_OOM, "workspace reservation too small");
    }
    hp->tmp = (float *)ws_alloc(ctx, sizeof(float) * tmp_floats * n_frames);
    if (!hp->tmp) return imgfd_fail(ctx, IMGFD_ERR_OOM, "workspace reservation too small");
    return compact_carve(ctx, nx, ny, n_frames, &hp->cb);
}

// the device part of harris(): harris.cpp:510-523.  Input may be u8 or f32 with its own pitch.
imgfd_status harris_device_stages(imgfd_ctx *ctx, const void *d_in, int in_is_u8, int in_pitch,
                                  size_t in_frame_stride, int nx, int ny, int n_frames, const HarrisArgs &a,
                                  const HarrisPlanes &hp, imgfd_corner *d_corners, int64_t cap, int64_t *d_counts,
                                  double *stage_seconds, bool need_R_plane)
{
    double t0 = 0;
    auto tick = [&](int stage) -> imgfd_status {
        if (!stage_seconds) return IMGFD_OK;
        IMGFD_HIP(ctx, hipStreamSynchronize(ctx->stream));
        const double t = now_s();
        if (stage >= 0) stage_seconds[stage] = t - t0;
        t0 = t;
        return IMGFD_OK;
    };
    IMGFD_TRY(tick(-1));
    bool rc_cleared = false;  // the marching Gaussian/gradient kernel clears the row counters on its way
    // (the row counts are cleared right before the NMS kernel that fills them: queued first, the 0.3 MB fill sat 89 us in
    // front of the Gaussian/gradient kernel in imgfd_detect_dev, waiting for a slot beside the first hysteresis sweep)
    if (!stage_seconds && gauss_grad_fused_supported(nx, ny, a.sigma_d, a.gauss)) {
        // the default path: Gaussian (radius 3) and gradient in one kernel, the smoothed plane stays on chip
        IMGFD_TRY(launch_gauss_grad_fused(ctx, d_in, in_is_u8, in_pitch, in_frame_stride, hp.Ix, hp.Iy, nx, ny, n_frames,
                                          a.sigma_d, a.grad, hp.cb.rowcount, &rc_cleared));
    } else {
        IMGFD_TRY(launch_gaussian(ctx, d_in, in_is_u8, in_pitch, in_frame_stride, hp.Is, nx, ny, n_frames, a.sigma_d,
                                  a.gauss, hp.tmp));
        IMGFD_TRY(tick(0));
        IMGFD_TRY(launch_gradient(ctx, hp.Is, hp.Ix, hp.Iy, nx, ny, n_frames, a.grad));
        IMGFD_TRY(tick(1));
    }
    const int radius = (int)(2 * a.sigma_i + 0.5);  // harris.cpp:523, double -> int truncation
    if (!stage_seconds && tensor_response_supported(ctx, nx, ny, a.sigma_i, a.gauss, a.measure, hp.Ix, hp.Iy, hp.R)) {
        // the default path: the structure tensor never leaves the CU -- its kernel's epilogue evaluates the corner
        // response (harris.cpp:78-133) and only R is stored; NMS then reads 4 B/px instead of 12
        IMGFD_TRY(prof_mark(ctx));
        unsigned char *tq = reinterpret_cast<unsigned char *>(hp.A);  // the A plane is idle on this path: it holds the threshold quads
        IMGFD_TRY(launch_tensor_response(ctx, hp.Ix, hp.Iy, hp.R, nx, ny, n_frames, a.sigma_i, a.k, tq, a.Th));
        IMGFD_TRY(prof_mark(ctx));
        if (!rc_cleared) IMGFD_TRY(compact_clear(ctx, hp.cb, ny, n_frames));
        IMGFD_TRY(launch_harris_nms_sparse(ctx, hp.R, tq, nx, ny, n_frames, a.Th, radius, hp.cb));
        IMGFD_TRY(compact_emit(ctx, hp.cb, nx, ny, n_frames, 0, hp.R, d_corners, cap, d_counts));
        return IMGFD_OK;
    }
    IMGFD_TRY(prof_mark(ctx));
    IMGFD_TRY(launch_structure_tensor(ctx, hp.Ix, hp.Iy, hp.A, hp.B, hp.C, nx, ny, n_frames, a.sigma_i, a.gauss,
                                      hp.tmp));
    IMGFD_TRY(prof_mark(ctx));
    IMGFD_TRY(tick(2));
    if (!rc_cleared) IMGFD_TRY(compact_clear(ctx, hp.cb, ny, n_frames));
    if (!need_R_plane && harris_resp_nms_supports(nx, ny, radius)) {
        // batch path: response + NMS in one kernel, strengths recomputed for the corner records (no R plane)
        IMGFD_TRY(launch_harris_resp_nms(ctx, hp.A, hp.B, hp.C, nx, ny, n_frames, a.measure, a.k, a.Th, radius, hp.cb));
        IMGFD_TRY(compact_emit_abc(ctx, hp.cb, nx, ny, n_frames, hp.A, hp.B, hp.C, a.measure, a.k, d_corners, cap, d_counts));
        IMGFD_TRY(tick(4));
        return IMGFD_OK;
    }
    IMGFD_TRY(launch_response(ctx, hp.A, hp.B, hp.C, hp.R, nx, ny, n_frames, a.measure, a.k));
    IMGFD_TRY(tick(3));
    IMGFD_TRY(launch_harris_nms(ctx, hp.R, nx, ny, n_frames, a.Th, radius, hp.cb));
    IMGFD_TRY(compact_emit(ctx, hp.cb, nx, ny, n_frames, 0, hp.R, d_corners, cap, d_counts));
    IMGFD_TRY(tick(4));
    return IMGFD_OK;
}

// harris(): one scale.  Device: the per-pixel stages, the raster-ordered corner list and -- when asked for -- the sub-pixel
// fit of EVERY corner beside it (a few thousand threads; the selection below needs the integer records' strengths, so both
// lists come back, 12 bytes per corner each).  Host: the selection.  d_I is an f32 device plane.
imgfd_status harris_one(imgfd_ctx *ctx, const float *d_I, int nx, int ny, const HarrisArgs &a,
                        std::vector<imgfd_corner> &corners, double *stage_seconds)
{
    corners.clear();
    if (nx < 3 || ny < 3) return IMGFD_OK;  // harris.cpp:493
    // the window rule admits at most one corner per 2x2 block (radius >= 1)
    const int64_t cap = (int64_t)nx * ny / 4 + 16;
    // d_I lives in the caller's arena slice; planes are carved after the current watermark
    const size_t mark = ctx->ws_used;
    HarrisPlanes hp;
    IMGFD_TRY(carve_planes(ctx, nx, ny, 1, harris_tmp_floats(nx, ny, a.sigma_d, a.sigma_i, a.gauss), &hp));
    imgfd_corner *d_corners = (imgfd_corner *)ws_alloc(ctx, sizeof(imgfd_corner) * (size_t)cap);
    imgfd_corner *d_refined = (imgfd_corner *)ws_alloc(ctx, sizeof(imgfd_corner) * (size_t)cap);
    int64_t *d_count = (int64_t *)ws_alloc(ctx, sizeof(int64_t));
    if (!d_corners || !d_refined || !d_count) return imgfd_fail(ctx, IMGFD_ERR_OOM, "workspace reservation too small");
    const bool sub_px = a.precision == IMGFD_QUADRATIC_APPROXIMATION || a.precision == IMGFD_QUARTIC_INTERPOLATION;
    IMGFD_TRY(harris_device_stages(ctx, d_I, 0, nx, (size_t)nx * ny, nx, ny, 1, a, hp, d_corners, cap, d_count,
                                   stage_seconds, /*need_R_plane=*/sub_px || stage_seconds != nullptr));
    int64_t n = 0;
    IMGFD_HIP(ctx, hipMemcpyAsync(&n, d_count, sizeof n, hipMemcpyDeviceToHost, ctx->stream));
    IMGFD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (n > cap) n = cap;
    std::vector<imgfd_corner> found((size_t)n), refined;
    double t = now_s();
    if (n) {
        if (sub_px) {
            IMGFD_TRY(launch_harris_refine(ctx, hp.R, nx, d_corners, n, a.precision, d_refined));
            refined.resize((size_t)n);
            IMGFD_HIP(ctx, hipMemcpyAsync(refined.data(), d_refined, sizeof(imgfd_corner) * n, hipMemcpyDeviceToHost, ctx->stream));
        }
        IMGFD_HIP(ctx, hipMemcpyAsync(found.data(), d_corners, sizeof(imgfd_corner) * n, hipMemcpyDeviceToHost, ctx->stream));
        IMGFD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    if (stage_seconds && sub_px) stage_seconds[6] = now_s() - t;  // the fit and its way back
    t = now_s();
    const std::vector<CornerIdx> keep = select_output(found, a.strategy, a.cells, a.N, nx, ny);
    const std::vector<imgfd_corner> &src = sub_px ? refined : found;
    corners.reserve(keep.size());
    for (CornerIdx i : keep) corners.push_back(src[i]);
    if (stage_seconds) stage_seconds[5] = now_s() - t;
    ctx->ws_used = mark;
    return IMGFD_OK;
}

// harris_scale(): harris.cpp:554-608.  A corner of this scale stays when some corner of the half-size image lies within
// sigma_i of its halved position (select_corners, :443-465; the halving runs in double, the distance in float)
imgfd_status harris_scale(imgfd_ctx *ctx, const float *d_I, int nx, int ny, int Nscales, HarrisArgs a,
                          std::vector<imgfd_corner> &corners, double *stage_seconds)
{
    if (Nscales <= 1 || nx <= 64 || ny <= 64) return harris_one(ctx, d_I, nx, ny, a, corners, stage_seconds);
    const int nxx = nx / 2, nyy = ny / 2;
    const size_t mark = ctx->ws_used;
    float *d_Iz = (float *)ws_alloc(ctx, sizeof(float) * (size_t)nxx * nyy);
    if (!d_Iz) return imgfd_fail(ctx, IMGFD_ERR_OOM, "workspace reservation too small");
    hipLaunchKernelGGL(zoom_out_kernel, dim3(ceil_div(nxx, 256), nyy), dim3(256), 0, ctx->stream, d_I, d_Iz, nx, nxx,
                       nyy);
    std::vector<imgfd_corner> coarse;
    HarrisArgs az = a;
    az.sigma_i = a.sigma_i / 2;
    IMGFD_TRY(harris_scale(ctx, d_Iz, nxx, nyy, Nscales - 1, az, coarse, nullptr));
    ctx->ws_used = mark;
    IMGFD_TRY(harris_one(ctx, d_I, nx, ny, a, corners, stage_seconds));
    const float reach2 = a.sigma_i * a.sigma_i;
    auto confirmed = [&](const imgfd_corner &fine) {
        for (const imgfd_corner &z : coarse) {
            const float ex = (float)((double)z.x - (double)fine.x / 2.), ey = (float)((double)z.y - (double)fine.y / 2.);
            if (!(ex * ex + ey * ey > reach2)) return true;
        }
        return false;
    };
    corners.erase(std::remove_if(corners.begin(), corners.end(), [&](const imgfd_corner &c) { return !confirmed(c); }), corners.end());
    return IMGFD_OK;
}

}  // namespace

extern "C" {

static imgfd_status harris_host(imgfd_ctx *ctx, const void *img, int kind, int nx, int ny, float k, float sigma_d,
                                float sigma_i, float threshold, int gaussian, int gradient, int strategy,
                                int Nselect, int measure, int Nscales, int precision, int cells, int verbose,
                                imgfd_corners *out)
{
    if (!ctx || !out) return IMGFD_ERR_INVALID;
    out->corners = nullptr;
    out->n = 0;
    memset(out->stage_seconds, 0, sizeof out->stage_seconds);
    if (!img || nx < 0 || ny < 0 || !frame_fits(nx, ny)) return imgfd_fail(ctx, IMGFD_ERR_INVALID, "imgfd_harris: bad image");
    if (nx < 3 || ny < 3) return IMGFD_OK;  // harris.cpp:493 (harris_scale falls through to harris for small images)
    IMGFD_HIP(ctx, hipSetDevice(ctx->device));
    // arena: the input plane, a half-size pyramid for the scale check, and one set of stage planes
    const size_t plane = align_up(sizeof(float) * (size_t)nx * ny, 256);
    // (the coarser scales use sigma_i/2, /4, ... on smaller images: their scratch is never larger than this one)
    size_t need = plane + harris_ws_bytes(nx, ny, 1, (int64_t)nx * ny / 4 + 16,
                                          harris_tmp_floats(nx, ny, sigma_d, sigma_i, gaussian));
    if (Nscales > 1) need += plane;  // sum of the decimated copies is < plane/3; keep it simple
    need += upload_stage_bytes(kind, (size_t)nx * ny);
    IMGFD_TRY(ws_reserve(ctx, need));
    float *d_I = (float *)ws_alloc(ctx, sizeof(float) * (size_t)nx * ny);
    if (!d_I) return imgfd_fail(ctx, IMGFD_ERR_OOM, "workspace reservation too small");
    IMGFD_TRY(upload_image(ctx, img, kind, (size_t)nx * ny, d_I));
    HarrisArgs a{k, sigma_d, sigma_i, threshold, gaussian, gradient, measure, strategy, cells, Nselect, precision, verbose};
    std::vector<imgfd_corner> corners;
    IMGFD_TRY(harris_scale(ctx, d_I, nx, ny, Nscales, a, corners, verbose ? out->stage_seconds : nullptr));
    out->n = (int64_t)corners.size();
    if (out->n) {
        out->corners = (imgfd_corner *)malloc(sizeof(imgfd_corner) * corners.size());
        if (!out->corners) return imgfd_fail(ctx, IMGFD_ERR_OOM, "malloc of the corner list failed");
        memcpy(out->corners, corners.data(), sizeof(imgfd_corner) * corners.size());
    }
    return IMGFD_OK;
}

imgfd_status imgfd_harris(imgfd_ctx *ctx, const float *img, int nx, int ny, float k, float sigma_d,
                          float sigma_i, float threshold, int gaussian, int gradient, int strategy,
                          int Nselect, int measure, int Nscales, int precision, int cells, int verbose,
                          imgfd_corners *out)
{
    return imgfd_guard(ctx, [&] { return harris_host(ctx, img, IMGFD_SRC_F32, nx, ny, k, sigma_d, sigma_i, threshold, gaussian, gradient, strategy, Nselect,
                       measure, Nscales, precision, cells, verbose, out); });
}

imgfd_status imgfd_harris_f64(imgfd_ctx *ctx, const double *x, int nx, int ny, float k, float sigma_d,
                              float sigma_i, float threshold, int gaussian, int gradient, int strategy,
                              int Nselect, int measure, int Nscales, int precision, int cells, int verbose,
                              imgfd_corners *out)
{
    return imgfd_guard(ctx, [&] { return harris_host(ctx, x, IMGFD_SRC_F64, nx, ny, k, sigma_d, sigma_i, threshold, gaussian, gradient, strategy, Nselect,
                       measure, Nscales, precision, cells, verbose, out); });
}

imgfd_status imgfd_harris_dev(imgfd_ctx *ctx, const imgfd_frames *fr, float k, float sigma_d,
                              float sigma_i, float threshold, int gaussian, int gradient, int measure,
                              imgfd_corner *d_corners, int64_t cap, int64_t *d_counts)
{
    if (!ctx || !fr || !fr->d_frames || (!d_corners && cap > 0) || !d_counts || cap < 0 || fr->n_frames < 0 ||
        fr->nx < 0 || fr->ny < 0 || !frame_fits(fr->nx, fr->ny))
        return imgfd_fail(ctx, IMGFD_ERR_INVALID, "imgfd_harris_dev: bad argument");
    const int nx = fr->nx, ny = fr->ny;
    if (fr->n_frames == 0) return IMGFD_OK;
    IMGFD_HIP(ctx, hipSetDevice(ctx->device));
    if (nx < 3 || ny < 3) {
        IMGFD_HIP(ctx, hipMemsetAsync(d_counts, 0, sizeof(int64_t) * fr->n_frames, ctx->stream));
        return IMGFD_OK;
    }
    const int esz = fr->dtype == 0 ? 1 : 4;
    if (fr->row_stride_bytes % esz || fr->frame_stride_bytes % esz)
        return imgfd_fail(ctx, IMGFD_ERR_INVALID, "imgfd_harris_dev: strides must be multiples of the element size");
    // sub-batches bounded by 12 GiB of stage planes
    const size_t per_frame = 8 * sizeof(float) * (size_t)nx * ny;
    const int chunk = sub_batch_frames(ctx, fr->n_frames, per_frame, (size_t)12 << 30);
    const size_t tmp_floats = harris_tmp_floats(nx, ny, sigma_d, sigma_i, gaussian);
    IMGFD_TRY(ws_reserve(ctx, harris_ws_bytes(nx, ny, chunk, 0, tmp_floats)));
    HarrisPlanes hp;
    IMGFD_TRY(carve_planes(ctx, nx, ny, chunk, tmp_floats, &hp));
    HarrisArgs a{k, sigma_d, sigma_i, threshold, gaussian, gradient, measure, 0, 0, 0, 0, 0};
    for (int f0 = 0; f0 < fr->n_frames; f0 += chunk) {
        const int nf = std::min(chunk, fr->n_frames - f0);
        const char *base = (const char *)fr->d_frames + (size_t)f0 * fr->frame_stride_bytes;
        IMGFD_TRY(harris_device_stages(ctx, base, fr->dtype == 0, fr->row_stride_bytes / esz,
                                       fr->frame_stride_bytes / esz, nx, ny, nf, a, hp, d_corners + (size_t)f0 * cap,
                                       cap, d_counts + f0, nullptr, /*need_R_plane=*/false));
    }
    return IMGFD_OK;
}

// ---- stage doorways -------------------------------------------------------------------------
imgfd_status imgfd_k_gaussian(imgfd_ctx *ctx, const float *d_in, float *d_out, int nx, int ny, float sigma, int type)
{
    if (!ctx || !d_in || !d_out || nx < 1 || ny < 1) return imgfd_fail(ctx, IMGFD_ERR_INVALID, "imgfd_k_gaussian: bad argument");
    const size_t tb = std::max(gaussian_tmp_bytes(nx, ny, 1, sigma, type, 1), sizeof(float) * (size_t)nx * ny);
    IMGFD_TRY(ws_reserve(ctx, tb + 4096));
    float *tmp = (float *)ws_alloc(ctx, tb);
    return launch_gaussian(ctx, d_in, 0, nx, (size_t)nx * ny, d_out, nx, ny, 1, sigma, type, tmp);
}

imgfd_status imgfd_k_gradient(imgfd_ctx *ctx, const float *d_I, float *d_Ix, float *d_Iy, int nx, int ny, int type)
{
    if (!ctx || !d_I || !d_Ix || !d_Iy || nx < 3 || ny < 3) return imgfd_fail(ctx, IMGFD_ERR_INVALID, "imgfd_k_gradient: bad argument");
    return launch_gradient(ctx, d_I, d_Ix, d_Iy, nx, ny, 1, type);
}

imgfd_status imgfd_k_gauss_grad_u8(imgfd_ctx *ctx, const uint8_t *d_u8, float *d_Ix, float *d_Iy, int nx, int ny, float sigma_d,
                                   int grad_type)
{
    if (!ctx || !d_u8 || !d_Ix || !d_Iy || nx < 3 || ny < 3) return imgfd_fail(ctx, IMGFD_ERR_INVALID, "imgfd_k_gauss_grad_u8: bad argument");
    if (!gauss_grad_fused_supported(nx, ny, sigma_d, IMGFD_STD_GAUSSIAN))
        return imgfd_fail(ctx, IMGFD_ERR_UNSUPPORTED, "imgfd_k_gauss_grad_u8: the fused kernel serves radius 3 (sigma_d in [1, 4/3)) only");
    return launch_gauss_grad_fused(ctx, d_u8, 1, nx, (size_t)nx * ny, d_Ix, d_Iy, nx, ny, 1, sigma_d, grad_type);
}

imgfd_status imgfd_k_structure_tensor(imgfd_ctx *ctx, const float *d_Ix, const float *d_Iy, float *d_A,
                                      float *d_B, float *d_C, int nx, int ny, float sigma, int gauss)
{
    if (!ctx || !d_Ix || !d_Iy || !d_A || !d_B || !d_C || nx < 1 || ny < 1)
        return imgfd_fail(ctx, IMGFD_ERR_INVALID, "imgfd_k_structure_tensor: bad argument");
    const size_t tb = std::max(gaussian_tmp_bytes(nx, ny, 1, sigma, gauss == IMGFD_NO_GAUSSIAN ? IMGFD_FAST_GAUSSIAN : gauss, 3),
                               sizeof(float) * (size_t)nx * ny);
    IMGFD_TRY(ws_reserve(ctx, tb + 4096));
    float *tmp = (float *)ws_alloc(ctx, tb);
    return launch_structure_tensor(ctx, d_Ix, d_Iy, d_A, d_B, d_C, nx, ny, 1, sigma, gauss, tmp);
}

imgfd_status imgfd_k_response(imgfd_ctx *ctx, const float *d_A, const float *d_B, const float *d_C,
                              float *d_R, int nx, int ny, int measure, float k)
{
    if (!ctx || !d_A || !d_B || !d_C || !d_R || nx < 1 || ny < 1)
        return imgfd_fail(ctx, IMGFD_ERR_INVALID, "imgfd_k_response: bad argument");
    return launch_response(ctx, d_A, d_B, d_C, d_R, nx, ny, 1, measure, k);
}

imgfd_status imgfd_k_nms(imgfd_ctx *ctx, const float *d_R, int nx, int ny, float Th, int radius,
                         imgfd_corner *d_corners, int64_t cap, int64_t *d_count)
{
    if (!ctx || !d_R || !d_corners || !d_count || nx < 1 || ny < 1 || cap < 0)
        return imgfd_fail(ctx, IMGFD_ERR_INVALID, "imgfd_k_nms: bad argument");
    IMGFD_TRY(ws_reserve(ctx, compact_bytes(nx, ny, 1) + 4096));
    CompactBuffers cb;
    IMGFD_TRY(compact_carve(ctx, nx, ny, 1, &cb));
    IMGFD_TRY(compact_clear(ctx, cb, ny, 1));
    // the tiled kernel of the batch path (radius <= 6; it falls back to the per-pixel kernel beyond)
    IMGFD_TRY(launch_harris_nms_tiled(ctx, d_R, nx, ny, 1, Th, radius, cb));
    return compact_emit(ctx, cb, nx, ny, 1, 0, d_R, d_corners, cap, d_count);
}

imgfd_status imgfd_k_nms_quads(imgfd_ctx *ctx, const float *d_R, int nx, int ny, float Th, int radius,
                               imgfd_corner *d_corners, int64_t cap, int64_t *d_count)
{
    if (!ctx || !d_R || !d_corners || !d_count || nx < 4 || nx % 4 != 0 || ny < 1 || cap < 0)
        return imgfd_fail(ctx, IMGFD_ERR_INVALID, "imgfd_k_nms_quads: bad argument (rows of whole quads)");
    const size_t quads = (size_t)(nx / 4) * ny;
    IMGFD_TRY(ws_reserve(ctx, compact_bytes(nx, ny, 1) + quads + 8192));
    CompactBuffers cb;
    IMGFD_TRY(compact_carve(ctx, nx, ny, 1, &cb));
    unsigned char *tq = (unsigned char *)ws_alloc(ctx, quads);
    if (!tq) return imgfd_fail(ctx, IMGFD_ERR_OOM, "workspace reservation too small");
    IMGFD_TRY(compact_clear(ctx, cb, ny, 1));
    // the batch path's pair: threshold quads (there: written by the structure-tensor kernel's epilogue), then the sparse kernel
    IMGFD_TRY(launch_harris_threshold_quads(ctx, d_R, tq, nx, ny, 1, Th));
    IMGFD_TRY(launch_harris_nms_sparse(ctx, d_R, tq, nx, ny, 1, Th, radius, cb));
    return compact_emit(ctx, cb, nx, ny, 1, 0, d_R, d_corners, cap, d_count);
}

imgfd_status imgfd_k_tensor_response(imgfd_ctx *ctx, const float *d_Ix, const float *d_Iy, float *d_R, int nx, int ny,
                                     float sigma, float k)
{
    if (!ctx || !d_Ix || !d_Iy || !d_R || nx < 1 || ny < 1)
        return imgfd_fail(ctx, IMGFD_ERR_INVALID, "imgfd_k_tensor_response: bad argument");
    if (!tensor_response_supported(ctx, nx, ny, sigma, IMGFD_STD_GAUSSIAN, IMGFD_HARRIS_MEASURE, d_Ix, d_Iy, d_R))
        return imgfd_fail(ctx, IMGFD_ERR_UNSUPPORTED, "imgfd_k_tensor_response: needs the discrete Gaussian with radius 7, 3 or 1, "
                                                      "16-byte aligned planes and rows of whole quads");
    return launch_tensor_response(ctx, d_Ix, d_Iy, d_R, nx, ny, 1, sigma, k);
}

const char *imgfd_tensor_kernel_name(imgfd_ctx *ctx)
{
    (void)ctx;
    return "fir_tensor<7, fma, vec, response> (structure tensor + Harris response, 12 B/px)";
}

imgfd_status imgfd_time_structure_tensor(imgfd_ctx *ctx, const float *d_Ix, const float *d_Iy,
                                         float *d_A, float *d_B, float *d_C, int nx, int ny,
                                         float sigma, int gauss, int warmup, int iters, double *avg_us)
{
    return imgfd_time_structure_tensor_batch(ctx, d_Ix, d_Iy, d_A, d_B, d_C, nx, ny, 1, sigma, gauss, warmup, iters, avg_us);
}

imgfd_status imgfd_time_structure_tensor_batch(imgfd_ctx *ctx, const float *d_Ix, const float *d_Iy,
                                               float *d_A, float *d_B, float *d_C, int nx, int ny, int n_frames,
                                               float sigma, int gauss, int warmup, int iters, double *avg_us)
{
    if (!ctx || !avg_us || iters < 1 || n_frames < 1) return imgfd_fail(ctx, IMGFD_ERR_INVALID, "imgfd_time_structure_tensor: bad argument");
    IMGFD_TRY(ws_reserve(ctx, sizeof(float) * (size_t)nx * ny * n_frames + 4096));
    float *tmp = (float *)ws_alloc(ctx, sizeof(float) * (size_t)nx * ny * n_frames);
    for (int i = 0; i < warmup; i++)
        IMGFD_TRY(launch_structure_tensor(ctx, d_Ix, d_Iy, d_A, d_B, d_C, nx, ny, n_frames, sigma, gauss, tmp));
    IMGFD_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    for (int i = 0; i < iters; i++)
        IMGFD_TRY(launch_structure_tensor(ctx, d_Ix, d_Iy, d_A, d_B, d_C, nx, ny, n_frames, sigma, gauss, tmp));
    IMGFD_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    IMGFD_HIP(ctx, hipEventSynchronize(ctx->ev1));
    float ms = 0;
    IMGFD_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    *avg_us = 1e3 * (double)ms / iters;
    return IMGFD_OK;
}

}  // extern "C"
