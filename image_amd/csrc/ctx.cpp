// image_amd/csrc/ctx.cpp -- context, workspace arena and small utilities of libimgfd.
#include "common.h"

#include <stdlib.h>

#include <algorithm>

namespace {
struct TuneKey { const char *name, *env; int imgfd_ctx::Tune::*field; };
const std::vector<TuneKey> &tune_keys()
{
    static const std::vector<TuneKey> keys = {
        {"fir_mode", "IMGFD_FIR_MODE", nullptr},
        {"fhog_fused", "IMGFD_FHOG_FUSED", &imgfd_ctx::Tune::fhog_fused},
        {"fhog_bands", "IMGFD_FHOG_BANDS", &imgfd_ctx::Tune::fhog_bands},
        {"hyst_sweeps", "IMGFD_HYST_SWEEPS", &imgfd_ctx::Tune::hyst_sweeps},
        {"gauss_march", "IMGFD_GAUSS_MARCH", &imgfd_ctx::Tune::gauss_march},
        {"gauss_march_seg", "IMGFD_GAUSS_MARCH_SEG", &imgfd_ctx::Tune::gauss_march_seg},
        {"tensor_workers", "IMGFD_TENSOR_WORKERS", &imgfd_ctx::Tune::tensor_workers},
        {"max_chunk_frames", "IMGFD_MAX_CHUNK_FRAMES", &imgfd_ctx::Tune::max_chunk_frames},
        {"surf_lanes", "IMGFD_SURF_LANES", &imgfd_ctx::Tune::surf_lanes},
        {"surf_group", "IMGFD_SURF_GROUP", &imgfd_ctx::Tune::surf_group},
        {"surf_sort_cap", "IMGFD_SURF_SORT_CAP", &imgfd_ctx::Tune::surf_sort_cap},
        {"surf_rec_cap", "IMGFD_SURF_REC_CAP", &imgfd_ctx::Tune::surf_rec_cap},
    };
    return keys;
}
}  // namespace

extern "C" {

int imgfd_version(void) { return IMGFD_VERSION; }

static imgfd_status ctx_init(int device, void *stream, bool own, imgfd_ctx **out)
{
    if (!out) return IMGFD_ERR_INVALID;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return IMGFD_ERR_NO_DEVICE;
    if (device < 0 || device >= count) return IMGFD_ERR_INVALID;
    if (hipSetDevice(device) != hipSuccess) return IMGFD_ERR_NO_DEVICE;
    imgfd_ctx *ctx = new (std::nothrow) imgfd_ctx();
    if (!ctx) return IMGFD_ERR_OOM;
    ctx->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) {
        ctx->num_cu = prop.multiProcessorCount;
        ctx->coop = prop.cooperativeLaunch != 0;
    }
    if (own) {
        if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
            delete ctx;
            return IMGFD_ERR_HIP;
        }
    } else {
        ctx->stream = (hipStream_t)stream;
    }
    ctx->own_stream = own;
    if (hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess) {
        delete ctx;
        return IMGFD_ERR_HIP;
    }
    for (const TuneKey &k : tune_keys())
        if (const char *e = getenv(k.env)) {
            if (k.field) ctx->tune.*(k.field) = atoi(e);
            else ctx->fir_mode = atoi(e) ? 1 : 0;  // "fir_mode": the one switch with a setter of its own (imgfd_set_fir_mode)
        }
    *out = ctx;
    return IMGFD_OK;
}

imgfd_status imgfd_device_count(int *count)
{
    if (!count) return IMGFD_ERR_INVALID;
    *count = 0;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return IMGFD_ERR_NO_DEVICE;
    *count = n;
    return IMGFD_OK;
}

imgfd_status imgfd_ctx_create(int device, imgfd_ctx **out) { return ctx_init(device, nullptr, true, out); }

imgfd_status imgfd_ctx_create_on_stream(int device, void *stream, imgfd_ctx **out)
{
    return ctx_init(device, stream, false, out);
}

void imgfd_ctx_destroy(imgfd_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);  // nothing of this context is in flight when its graph, events and buffers go
    if (ctx->side) imgfd_ctx_destroy(ctx->side);
    for (hipEvent_t e : {ctx->ev_fork, ctx->ev_gate, ctx->ev_gate2, ctx->ev_join})
        if (e) (void)hipEventDestroy(e);
    if (ctx->ws) (void)hipFree(ctx->ws);
    if (ctx->pin) (void)hipHostFree(ctx->pin);
    if (ctx->aux) (void)hipFree(ctx->aux);
    if (ctx->fhog_lut) (void)hipFree(ctx->fhog_lut);
    if (ctx->taps_dev) (void)hipFree(ctx->taps_dev);
    if (ctx->surf_pool && ctx->surf_pool_free) ctx->surf_pool_free(ctx->surf_pool);
    for (hipEvent_t e : ctx->surf_ev) (void)hipEventDestroy(e);
    if (ctx->clk_stream) { (void)hipStreamSynchronize(ctx->clk_stream); (void)hipStreamDestroy(ctx->clk_stream); }
    if (ctx->clk_ring) (void)hipFree(ctx->clk_ring);
    if (ctx->canny_taps && ctx->canny_taps_free) ctx->canny_taps_free(ctx->canny_taps);
    if (ctx->canny_report) (void)hipHostFree(ctx->canny_report);
    for (hipEvent_t e : ctx->prof_ev) (void)hipEventDestroy(e);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char *imgfd_last_error(const imgfd_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }
void *imgfd_ctx_stream(imgfd_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

imgfd_status imgfd_ctx_sync(imgfd_ctx *ctx)
{
    if (!ctx) return IMGFD_ERR_INVALID;
    if (ctx->side) {
        const imgfd_status st = imgfd_ctx_sync(ctx->side);
        if (st != IMGFD_OK) { ctx->err = ctx->side->err; return st; }
    }
    IMGFD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return IMGFD_OK;
}

void imgfd_free(void *p) { free(p); }

imgfd_status imgfd_set_fir_mode(imgfd_ctx *ctx, int mode)
{
    if (!ctx || (mode != 0 && mode != 1)) return IMGFD_ERR_INVALID;
    for (imgfd_ctx *c = ctx; c; c = c->side) c->fir_mode = mode;  // the whole chain of companions (imgfd_surf_dev runs up to four lanes)
    return IMGFD_OK;
}

imgfd_status imgfd_set_tuning(imgfd_ctx *ctx, const char *name, int value)
{
    if (!ctx || !name) return IMGFD_ERR_INVALID;
    for (const TuneKey &k : tune_keys())
        if (!strcmp(k.name, name)) {
            if (!k.field) return imgfd_set_fir_mode(ctx, value);
            for (imgfd_ctx *c = ctx; c; c = c->side) c->tune.*(k.field) = value;  // every companion, not only the first
            return IMGFD_OK;
        }
    return imgfd_fail(ctx, IMGFD_ERR_INVALID, "imgfd_set_tuning: unknown switch");
}

imgfd_status imgfd_get_counter(imgfd_ctx *ctx, const char *name, int64_t *value)
{
    if (!ctx || !name || !value) return IMGFD_ERR_INVALID;
    if (!strcmp(name, "gauss_march_launches")) { *value = ctx->gauss_march_launches; return IMGFD_OK; }
    if (!strcmp(name, "canny_sweeps_queued")) {  // sweep launches the last Canny call on this context (its companion's, for imgfd_detect_dev) queued
        *value = ctx->canny_sweeps ? ctx->canny_sweeps : (ctx->side ? ctx->side->canny_sweeps : 0);
        return IMGFD_OK;
    }
    if (!strcmp(name, "canny_frames_unconverged") || !strcmp(name, "canny_sweeps_working")) {
        // diagnostics of the last Canny call on this context (its companion's, for imgfd_detect_dev): frames the queued sweeps
        // did not finish (the union-find kernels did), and the number of the last sweep that changed anything.  Waits for the stream.
        const imgfd_ctx *c = ctx->canny_flags ? ctx : (ctx->side && ctx->side->canny_flags ? ctx->side : nullptr);
        *value = 0;
        if (!c) return IMGFD_OK;
        // the flags live in the workspace arena of the call that wrote them: an arena that has been regrown since is gone
        const char *lo = c->ws, *hi = c->ws + c->ws_size, *p = reinterpret_cast<const char *>(c->canny_flags);
        if (!c->ws || p < lo || p + 4 * (32 + (size_t)c->canny_frames) > hi) return imgfd_fail(ctx, IMGFD_ERR_INVALID, "imgfd_get_counter: no Canny call since the workspace was last resized");
        IMGFD_HIP(ctx, hipStreamSynchronize(c->stream));
        std::vector<unsigned> f(32 + (size_t)c->canny_frames);  // HY_SWEEPS_MAX sweep flags, then one word per frame
        IMGFD_HIP(ctx, hipMemcpy(f.data(), c->canny_flags, 4 * f.size(), hipMemcpyDeviceToHost));
        for (int i = 0; i < c->canny_frames; i++) {
            if (!strcmp(name, "canny_frames_unconverged")) *value += f[32 + i] == (unsigned)c->canny_sweeps;
            else *value = std::max<int64_t>(*value, f[32 + i]);
        }
        return IMGFD_OK;
    }
    for (const TuneKey &k : tune_keys())
        if (!strcmp(k.name, name)) { *value = k.field ? ctx->tune.*(k.field) : ctx->fir_mode; return IMGFD_OK; }
    return imgfd_fail(ctx, IMGFD_ERR_INVALID, "imgfd_get_counter: unknown name");
}

imgfd_status imgfd_profile_k3(imgfd_ctx *ctx, int enable)
{
    if (!ctx) return IMGFD_ERR_INVALID;
    // the whole chain of companions: imgfd_detect_dev runs the Harris chain on the companion's stream for small batches
    for (imgfd_ctx *c = ctx; c; c = c->side) {
        c->prof_on = enable != 0;
        c->prof_used = 0;
    }
    return IMGFD_OK;
}

imgfd_status imgfd_profile_k3_read(imgfd_ctx *ctx, double *total_us, int *launches)
{
    if (!ctx || !total_us || !launches) return IMGFD_ERR_INVALID;
    double tot = 0;
    size_t all = 0;
    for (imgfd_ctx *c = ctx; c; c = c->side) {  // each context's pairs were recorded on its own stream
        IMGFD_HIP(ctx, hipStreamSynchronize(c->stream));
        const size_t pairs = c->prof_used / 2;
        for (size_t i = 0; i < pairs; i++) {
            float ms = 0;
            IMGFD_HIP(ctx, hipEventElapsedTime(&ms, c->prof_ev[2 * i], c->prof_ev[2 * i + 1]));
            tot += 1e3 * (double)ms;
        }
        all += pairs;
        c->prof_used = 0;
    }
    *total_us = tot;
    *launches = (int)all;
    return IMGFD_OK;
}

}  // extern "C"

imgfd_status prof_mark(imgfd_ctx *ctx)
{
    if (!ctx->prof_on) return IMGFD_OK;
    if (ctx->prof_used == ctx->prof_ev.size()) {
        hipEvent_t e;
        IMGFD_HIP(ctx, hipEventCreate(&e));
        ctx->prof_ev.push_back(e);
    }
    IMGFD_HIP(ctx, hipEventRecord(ctx->prof_ev[ctx->prof_used++], ctx->stream));
    return IMGFD_OK;
}

int tile_run_length(const imgfd_ctx *ctx, int tiles_x, int bands, int frames)
{
    const int num_cu = ctx->num_cu;
    // Measured on 32 x 4K frames (scripts/gpu_u8.sh, profiles/r02/u8_runs.txt): runs of 4 cut the fetched bytes by a third
    // at the same or a slightly better duration; longer runs fetch less still but serialise too much of a workgroup's
    // latency (staging -> phases -> barriers) and run slower, as does a grid of resident workgroups walking the runs
    // round-robin instead of one workgroup per run.  A small batch keeps one tile per workgroup: it needs the parallelism.
    const long tiles = (long)tiles_x * bands * frames;
    return tiles >= 64L * num_cu ? 4 : 1;
}

imgfd_status ws_reserve(imgfd_ctx *ctx, size_t bytes)
{
    ctx->ws_used = 0;
    ctx->canny_flags = nullptr;  // a new call carves the arena: what the last Canny call left in it (diagnostic counters) is no longer there
    if (bytes <= ctx->ws_size) return IMGFD_OK;
    // Kernels of earlier calls may still be using the old arena: wait for the stream, then free it here.  (Growth is rare --
    // the arena only ever grows -- and parking the old arena until the next imgfd_ctx_sync kept it alive for as long as a
    // caller of the host entry points, which synchronise on their own, never called that function.)
    if (ctx->ws) {
        IMGFD_HIP(ctx, hipStreamSynchronize(ctx->stream));
        (void)hipFree(ctx->ws);
    }
    ctx->ws = nullptr;
    ctx->ws_size = 0;
    bytes = align_up(bytes + bytes / 8, (size_t)1 << 20);
    void *p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) {
        ctx->err = "hipMalloc of the device workspace failed";
        return IMGFD_ERR_OOM;
    }
    ctx->ws = (char *)p;
    ctx->ws_size = bytes;
    return IMGFD_OK;
}

void *ws_alloc(imgfd_ctx *ctx, size_t bytes)
{
    size_t off = align_up(ctx->ws_used, 256);
    if (off + bytes > ctx->ws_size) return nullptr;
    ctx->ws_used = off + bytes;
    return ctx->ws + off;
}

imgfd_status ctx_side(imgfd_ctx *ctx, imgfd_ctx **side)
{
    if (!ctx->side) {
        imgfd_ctx *s = nullptr;
        const imgfd_status st = imgfd_ctx_create(ctx->device, &s);
        if (st != IMGFD_OK) return imgfd_fail(ctx, st, "could not create the companion context");
        s->fir_mode = ctx->fir_mode;
        s->tune = ctx->tune;
        s->prof_on = ctx->prof_on;
        // the three events first; the companion is published only when everything it needs exists
        hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
        for (int i = 0; i < 4; i++) {
            if (hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) != hipSuccess) {
                for (int j = 0; j < i; j++) (void)hipEventDestroy(ev[j]);
                imgfd_ctx_destroy(s);
                return imgfd_fail(ctx, IMGFD_ERR_HIP, "could not create the companion context's events");
            }
        }
        ctx->ev_fork = ev[0]; ctx->ev_gate = ev[1]; ctx->ev_join = ev[2]; ctx->ev_gate2 = ev[3];
        ctx->side = s;
    }
    *side = ctx->side;
    return IMGFD_OK;
}

imgfd_status aux_reserve(imgfd_ctx *ctx, size_t bytes)
{
    if (bytes <= ctx->aux_size) return IMGFD_OK;
    IMGFD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->aux) (void)hipFree(ctx->aux);
    ctx->aux = nullptr;
    ctx->aux_size = 0;
    void *p = nullptr;
    bytes = align_up(bytes, (size_t)1 << 20);
    if (hipMalloc(&p, bytes) != hipSuccess) {
        ctx->err = "hipMalloc of the auxiliary device buffer failed";
        return IMGFD_ERR_OOM;
    }
    ctx->aux = (char *)p;
    ctx->aux_size = bytes;
    return IMGFD_OK;
}

imgfd_status pin_reserve(imgfd_ctx *ctx, size_t bytes)
{
    if (bytes <= ctx->pin_size) return IMGFD_OK;
    IMGFD_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->pin) (void)hipHostFree(ctx->pin);
    ctx->pin = nullptr;
    ctx->pin_size = 0;
    void *p = nullptr;
    bytes = align_up(bytes, (size_t)1 << 20);
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) {
        ctx->err = "hipHostMalloc of the pinned staging buffer failed";
        return IMGFD_ERR_OOM;
    }
    ctx->pin = (char *)p;
    ctx->pin_size = bytes;
    return IMGFD_OK;
}
