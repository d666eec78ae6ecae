// image_amd/csrc/frame_stream.hip -- host frame streams: pinned, double-buffered upload overlapped with the kernels
// (SURVEY.md 8f row 2).  Host-side code only; the kernels are those behind imgfd_harris_dev / imgfd_fast9_dev /
// imgfd_canny_dev.
//
// The reference processes one image per R call (H/R/pkg.R:76-90 -> rcpp_harris.cpp:34-35, F9/R/image_detect_corners.R:
// 10-27 -> f9_rcpp.cpp:10-11, CE/R/canny_edges_detector.R:63 -> rcpp_canny.cpp:135-136); a caller with a directory of
// frames loops over them.  Here that loop sits below the C ABI:
//
//      helper thread :  [stage batch i+1 into pinned memory]  hipMemcpyAsync on the COPY stream, event `uploaded`
//      caller thread :  hipStreamWaitEvent(ctx stream, uploaded[i]); kernels of batch i; D2H of counts/lists; event `done`
//
// Three slots (device frames + device results + pinned results) rotate: one uploading, one in the kernels, one whose
// results the caller is still reading.
#include "common.h"

#include <stdlib.h>

#include <algorithm>
#include <new>
#include <thread>
#include <vector>

namespace {

enum { FS_SLOTS = 3 };
enum { SLOT_FREE = 0, SLOT_UPLOADING = 1, SLOT_COMPUTED = 2, SLOT_COLLECTED = 3 };

struct Slot {
    uint8_t *d_frames = nullptr;
    uint8_t *h_stage = nullptr;  // pinned staging, allocated on first use with pageable input
    imgfd_corner *d_corners = nullptr;
    imgfd_point *d_points = nullptr;
    uint8_t *d_edges = nullptr;
    int64_t *d_counts = nullptr;  // [3][batch]
    char *h_res = nullptr;        // pinned: counts, corners, points, edges
    hipEvent_t uploaded = nullptr, done = nullptr;
    std::thread up;
    hipError_t up_err = hipSuccess;
    int n = 0;
    int state = SLOT_FREE;
    int64_t first = 0;
};

}  // namespace

struct imgfd_stream {
    imgfd_ctx *ctx = nullptr;
    int nx = 0, ny = 0, batch = 0;
    imgfd_stream_params p;
    hipStream_t copy = nullptr;
    Slot slot[FS_SLOTS];
    int64_t submitted = 0;   // batches
    int64_t computed = 0;    // batches whose kernels have been launched
    int64_t collected = 0;   // batches handed back
    int64_t frames_seen = 0;
    size_t off_counts = 0, off_corners = 0, off_points = 0, off_edges = 0, res_bytes = 0;
};

namespace {

// copy with a few threads: one core moves ~10 GB/s, PCIe 5 x16 wants ~50
void staged_copy(uint8_t *dst, const uint8_t *src, size_t frame_bytes, size_t stride, int n)
{
    const size_t total = frame_bytes * (size_t)n;
    int nt = (int)std::min<size_t>(8, total >> 22);  // one thread per 4 MiB, at most 8
    if (nt <= 1) {
        for (int f = 0; f < n; f++) memcpy(dst + (size_t)f * frame_bytes, src + (size_t)f * stride, frame_bytes);
        return;
    }
    std::vector<std::thread> th;
    // split every frame into nt pieces so the split also helps a single large frame
    auto piece = [=](int t) {
        const size_t a = frame_bytes * (size_t)t / nt, b = frame_bytes * (size_t)(t + 1) / nt;
        for (int f = 0; f < n; f++) memcpy(dst + (size_t)f * frame_bytes + a, src + (size_t)f * stride + a, b - a);
    };
    for (int t = 0; t < nt; t++) {
        try {
            th.emplace_back(piece, t);
        } catch (...) {  // thread limit reached: copy this piece here
            piece(t);
        }
    }
    for (auto &t : th) t.join();
}

bool is_pinned(const void *p)
{
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) {
        (void)hipGetLastError();  // pageable memory: not an error for us
        return false;
    }
    return at.type == hipMemoryTypeHost;
}

void join_upload(Slot &s)
{
    if (s.up.joinable()) s.up.join();
}

// kernels + result download of the batch in slot s; asynchronous apart from the hysteresis read-backs of Canny
imgfd_status run_slot(imgfd_stream *st, Slot &s)
{
    imgfd_ctx *ctx = st->ctx;
    join_upload(s);
    if (s.up_err != hipSuccess) {
        ctx->err = std::string("frame stream upload failed: ") + hipGetErrorString(s.up_err);
        return IMGFD_ERR_HIP;
    }
    IMGFD_HIP(ctx, hipStreamWaitEvent(ctx->stream, s.uploaded, 0));
    const imgfd_stream_params &p = st->p;
    const size_t fb = (size_t)st->nx * st->ny;
    imgfd_frames fr;
    fr.d_frames = s.d_frames;
    fr.n_frames = s.n;
    fr.nx = st->nx;
    fr.ny = st->ny;
    fr.frame_stride_bytes = fb;
    fr.row_stride_bytes = st->nx;
    fr.dtype = 0;
    int64_t *cnt = s.d_counts;  // three arrays of s.n counts, back to back
    IMGFD_TRY(imgfd_detect_dev(ctx, &fr, &p, s.d_corners, s.d_points, s.d_edges, cnt));
    IMGFD_HIP(ctx, hipMemcpyAsync(s.h_res + st->off_counts, cnt, sizeof(int64_t) * 3 * s.n, hipMemcpyDeviceToHost, ctx->stream));
    if (p.harris && p.corner_cap > 0)
        IMGFD_HIP(ctx, hipMemcpyAsync(s.h_res + st->off_corners, s.d_corners, sizeof(imgfd_corner) * p.corner_cap * s.n,
                                      hipMemcpyDeviceToHost, ctx->stream));
    if (p.fast9 && p.point_cap > 0)
        IMGFD_HIP(ctx, hipMemcpyAsync(s.h_res + st->off_points, s.d_points, sizeof(imgfd_point) * p.point_cap * s.n,
                                      hipMemcpyDeviceToHost, ctx->stream));
    if (p.canny && p.keep_edges)
        IMGFD_HIP(ctx, hipMemcpyAsync(s.h_res + st->off_edges, s.d_edges, fb * s.n, hipMemcpyDeviceToHost, ctx->stream));
    IMGFD_HIP(ctx, hipEventRecord(s.done, ctx->stream));
    s.state = SLOT_COMPUTED;
    st->computed++;
    return IMGFD_OK;
}

void free_slot(Slot &s)
{
    join_upload(s);
    if (s.d_frames) (void)hipFree(s.d_frames);
    if (s.h_stage) (void)hipHostFree(s.h_stage);
    if (s.d_corners) (void)hipFree(s.d_corners);
    if (s.d_points) (void)hipFree(s.d_points);
    if (s.d_edges) (void)hipFree(s.d_edges);
    if (s.d_counts) (void)hipFree(s.d_counts);
    if (s.h_res) (void)hipHostFree(s.h_res);
    if (s.uploaded) (void)hipEventDestroy(s.uploaded);
    if (s.done) (void)hipEventDestroy(s.done);
    s = Slot();
}

}  // namespace

extern "C" {

void imgfd_stream_default_params(imgfd_stream_params *p)
{
    if (!p) return;
    memset(p, 0, sizeof *p);
    p->harris = p->fast9 = p->canny = 1;
    // image_harris() defaults, H/R/pkg.R:56-66
    p->k = 0.06f;
    p->sigma_d = 1.0f;
    p->sigma_i = 2.5f;
    p->threshold = 130.0f;
    // image_detect_corners() defaults, F9/R/image_detect_corners.R:48
    p->fast9_threshold = 50;
    p->suppress_non_max = 0;
    // image_canny_edge_detector() defaults, CE/R/canny_edges_detector.R:63
    p->s = 2.0;
    p->low_thr = 3.0;
    p->high_thr = 10.0;
    p->accGrad = 1;
}

imgfd_status imgfd_stream_open(imgfd_ctx *ctx, int nx, int ny, int batch_frames, const imgfd_stream_params *params,
                               imgfd_stream **out)
{
    if (out) *out = nullptr;
    if (!ctx || !out || !params || nx < 1 || ny < 1 || batch_frames < 1 || params->corner_cap < 0 || params->point_cap < 0 ||
        params->fast9_threshold < 0 || params->fast9_threshold > 255 || !frame_fits(nx, ny))
        return imgfd_fail(ctx, IMGFD_ERR_INVALID, "imgfd_stream_open: bad argument");
    if (!params->harris && !params->fast9 && !params->canny)
        return imgfd_fail(ctx, IMGFD_ERR_INVALID, "imgfd_stream_open: no detector selected");
    IMGFD_HIP(ctx, hipSetDevice(ctx->device));
    imgfd_stream *st = new (std::nothrow) imgfd_stream();
    if (!st) return imgfd_fail(ctx, IMGFD_ERR_OOM, "imgfd_stream_open: out of memory");
    st->ctx = ctx;
    st->nx = nx;
    st->ny = ny;
    st->batch = batch_frames;
    st->p = *params;
    const size_t fb = (size_t)nx * ny, B = (size_t)batch_frames;
    size_t o = 0;
    st->off_counts = o;
    o = align_up(o + sizeof(int64_t) * 3 * B, 256);
    st->off_corners = o;
    if (params->harris) o = align_up(o + sizeof(imgfd_corner) * (size_t)params->corner_cap * B, 256);
    st->off_points = o;
    if (params->fast9) o = align_up(o + sizeof(imgfd_point) * (size_t)params->point_cap * B, 256);
    st->off_edges = o;
    if (params->canny && params->keep_edges) o = align_up(o + fb * B, 256);
    st->res_bytes = o;
    hipError_t e = hipStreamCreateWithFlags(&st->copy, hipStreamNonBlocking);
    for (int i = 0; i < FS_SLOTS && e == hipSuccess; i++) {
        Slot &s = st->slot[i];
        e = hipMalloc((void **)&s.d_frames, fb * B);
        if (e == hipSuccess) e = hipMalloc((void **)&s.d_counts, sizeof(int64_t) * 3 * B);
        if (e == hipSuccess) e = hipMemset(s.d_counts, 0, sizeof(int64_t) * 3 * B);
        if (e == hipSuccess && params->harris)
            e = hipMalloc((void **)&s.d_corners, std::max<size_t>(1, (size_t)params->corner_cap * B) * sizeof(imgfd_corner));
        if (e == hipSuccess && params->fast9)
            e = hipMalloc((void **)&s.d_points, std::max<size_t>(1, (size_t)params->point_cap * B) * sizeof(imgfd_point));
        if (e == hipSuccess && params->canny) e = hipMalloc((void **)&s.d_edges, fb * B);
        if (e == hipSuccess) e = hipHostMalloc((void **)&s.h_res, st->res_bytes, 0);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&s.uploaded, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&s.done, hipEventDisableTiming);
    }
    if (e != hipSuccess) {
        ctx->err = std::string("imgfd_stream_open: ") + hipGetErrorString(e);
        imgfd_stream_close(st);
        return e == hipErrorOutOfMemory ? IMGFD_ERR_OOM : IMGFD_ERR_HIP;
    }
    *out = st;
    return IMGFD_OK;
}

imgfd_status imgfd_stream_submit(imgfd_stream *st, const uint8_t *frames, int n_frames, size_t frame_stride_bytes)
{
    if (!st) return IMGFD_ERR_INVALID;
    imgfd_ctx *ctx = st->ctx;
    const size_t fb = (size_t)st->nx * st->ny;
    if (!frames || n_frames < 1 || n_frames > st->batch || frame_stride_bytes < fb)
        return imgfd_fail(ctx, IMGFD_ERR_INVALID, "imgfd_stream_submit: bad argument (1..batch frames, stride >= nx*ny)");
    if (st->submitted - st->collected >= 2)
        return imgfd_fail(ctx, IMGFD_ERR_INVALID, "imgfd_stream_submit: two batches pending, collect one first");
    IMGFD_HIP(ctx, hipSetDevice(ctx->device));
    Slot &s = st->slot[st->submitted % FS_SLOTS];
    // the slot last held batch submitted-3, collected at least one collect ago: its results may be overwritten now
    join_upload(s);
    const bool pinned = is_pinned(frames);
    if (!pinned && !s.h_stage) IMGFD_HIP(ctx, hipHostMalloc((void **)&s.h_stage, fb * (size_t)st->batch, 0));
    s.n = n_frames;
    s.first = st->frames_seen;
    s.state = SLOT_UPLOADING;
    s.up_err = hipSuccess;
    const int device = ctx->device;
    hipStream_t copy = st->copy;
    Slot *sp = &s;
    auto upload = [=] {
        hipError_t e = hipSetDevice(device);
        if (e == hipSuccess) {
            if (pinned) {
                if (frame_stride_bytes == fb)
                    e = hipMemcpyAsync(sp->d_frames, frames, fb * (size_t)n_frames, hipMemcpyHostToDevice, copy);
                else
                    e = hipMemcpy2DAsync(sp->d_frames, fb, frames, frame_stride_bytes, fb, (size_t)n_frames,
                                         hipMemcpyHostToDevice, copy);
            } else {
                staged_copy(sp->h_stage, frames, fb, frame_stride_bytes, n_frames);
                e = hipMemcpyAsync(sp->d_frames, sp->h_stage, fb * (size_t)n_frames, hipMemcpyHostToDevice, copy);
            }
        }
        if (e == hipSuccess) e = hipEventRecord(sp->uploaded, copy);
        sp->up_err = e;
    };
    try {  // no C++ exception may cross the C boundary: without a helper thread the upload runs on the caller's
        s.up = std::thread(upload);
    } catch (...) {
        upload();
    }
    st->submitted++;
    st->frames_seen += n_frames;
    // the batch before this one goes into the kernels while this one is on the bus
    if (st->computed < st->submitted - 1) IMGFD_TRY(run_slot(st, st->slot[st->computed % FS_SLOTS]));
    return IMGFD_OK;
}

imgfd_status imgfd_stream_collect(imgfd_stream *st, imgfd_stream_result *res)
{
    if (!st || !res) return IMGFD_ERR_INVALID;
    imgfd_ctx *ctx = st->ctx;
    memset(res, 0, sizeof *res);
    if (st->collected == st->submitted) return IMGFD_OK;  // nothing pending
    IMGFD_HIP(ctx, hipSetDevice(ctx->device));
    while (st->computed <= st->collected) IMGFD_TRY(run_slot(st, st->slot[st->computed % FS_SLOTS]));
    Slot &s = st->slot[st->collected % FS_SLOTS];
    IMGFD_HIP(ctx, hipEventSynchronize(s.done));
    const imgfd_stream_params &p = st->p;
    const int B = s.n;  // stride of the three count arrays of this batch
    const int64_t *cnt = (const int64_t *)(s.h_res + st->off_counts);
    res->n_frames = s.n;
    res->first_frame = s.first;
    res->harris_counts = p.harris ? cnt : nullptr;
    res->fast9_counts = p.fast9 ? cnt + B : nullptr;
    res->canny_counts = p.canny ? cnt + 2 * B : nullptr;
    res->corners = (p.harris && p.corner_cap > 0) ? (const imgfd_corner *)(s.h_res + st->off_corners) : nullptr;
    res->points = (p.fast9 && p.point_cap > 0) ? (const imgfd_point *)(s.h_res + st->off_points) : nullptr;
    res->edges = (p.canny && p.keep_edges) ? (const uint8_t *)(s.h_res + st->off_edges) : nullptr;
    s.state = SLOT_COLLECTED;
    st->collected++;
    return IMGFD_OK;
}

void imgfd_stream_close(imgfd_stream *st)
{
    if (!st) return;
    (void)hipSetDevice(st->ctx->device);
    for (Slot &s : st->slot) join_upload(s);
    if (st->copy) (void)hipStreamSynchronize(st->copy);
    (void)hipStreamSynchronize(st->ctx->stream);
    for (Slot &s : st->slot) free_slot(s);
    if (st->copy) (void)hipStreamDestroy(st->copy);
    delete st;
}

void *imgfd_host_alloc(size_t bytes)
{
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, 0) != hipSuccess) return nullptr;
    return p;
}

void imgfd_host_free(void *p)
{
    if (p) (void)hipHostFree(p);
}

}  // extern "C"
