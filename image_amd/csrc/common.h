// image_amd/csrc/common.h -- shared host-side plumbing of libimgfd (context, workspace, error macros).
#pragma once
#include <hip/hip_runtime.h>

#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <functional>
#include <new>
#include <string>
#include <vector>

#include "../../include/imgfd.h"

#define IMGFD_MAX_TAPS 64 /* FIR half-size (B[0..size-1]) the kernels accept */

struct imgfd_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = true;
    int fir_mode = 1;  // 1 = fused accumulate, 0 = strict
    int num_cu = 256;
    bool coop = false;  // workgroups of one launch can wait for each other (hipDeviceProp_t::cooperativeLaunch; false on the tests' emulator)
    std::string err;
    // grow-only device workspace arena (bump-allocated per call, reset at call entry; when it has to grow, the stream is
    // drained and the old arena freed on the spot)
    char *ws = nullptr;
    size_t ws_size = 0;
    size_t ws_used = 0;
    // pinned host staging
    char *pin = nullptr;
    size_t pin_size = 0;
    // second grow-only device buffer for stages that start after the workspace arena has been carved (SURF K19)
    char *aux = nullptr;
    size_t aux_size = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // second context (own stream and workspace) for work that overlaps this context's stream (imgfd_detect_dev)
    imgfd_ctx *side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_gate = nullptr, ev_gate2 = nullptr, ev_join = nullptr;
    // Gaussian taps beyond the IMGFD_MAX_TAPS a kernel argument holds (sigma > 21): grow-only device copy (fir.hip)
    double *taps_dev = nullptr;
    size_t taps_cap = 0;
    long gauss_march_launches = 0;
    // Canny hysteresis, last call on this context: where its sweep flags live (workspace), sweeps queued, frames (diagnostic
    // counters "canny_frames_unconverged" / "canny_sweeps_working": imgfd_get_counter waits for the stream and reads them back)
    // Canny: the wrapped Gaussian taps of the last (nx, ny, s) -- making them costs nx + ny calls of exp(), ~80 us of host time
    // for a 4K frame, more than a single frame's kernels leave idle (canny.hip owns the object and its destructor)
    void *canny_taps = nullptr;
    void (*canny_taps_free)(void *) = nullptr;
    const unsigned *canny_flags = nullptr;
    // small batches: how many sweeps the recent calls needed, reported by canny_finish into pinned memory (canny.hip)
    unsigned *canny_report = nullptr;
    unsigned canny_report_seq = 0, canny_report_base = 0, canny_report_queued[4] = {0, 0, 0, 0};  // (CANNY_REPORTS slots)
    int canny_report_nx = 0, canny_report_ny = 0;
    int canny_finish_fit = -1;  // blocks of canny_finish the stream's compute units hold at once (-1: not asked yet; canny.hip)
    int canny_sweeps = 0, canny_frames = 0;
    // fHOG: magnitude + orientation of every integer gradient (fhog_fused.hip), built on first use
    unsigned *fhog_lut = nullptr;
    // lab switches (imgfd_set_tuning / IMGFD_* environment variables read ONCE at context creation; include/imgfd.h
    // lists them).  Defaults are the measured best; none changes a result.
    struct Tune {
        // (the measured-and-lost alternatives these replaced -- tile widths and block shapes of the sweeps, release points of imgfd_detect_dev,
        // hipGraph replay, tiled Harris NMS, 512-thread fHOG workgroups, plain-table SURF gathers ... -- left the library in round 6:
        // LOG.md, scripts/experiments/r06_pruned_switches.patch)
        int fhog_fused = 1;         // 1: cell_size 8 through fhog_hist8; 0: the three stage kernels (which serve every other cell size / width)
        int fhog_bands = 0;         // bands a workgroup of fhog_hist8 marches through (0: chosen from the batch size; tests: 1..3)
        int hyst_sweeps = 0;        // Canny hysteresis: sweeps queued before the union-find step (0: chosen; tests lower it to reach the union-find part)
        int gauss_march = 1;        // u8 frames whose width is a multiple of 16: the marching Gaussian + gradient kernel (0: the tile kernel, which serves every other shape)
        int gauss_march_seg = 0;    // rows per segment of that kernel (0: from the batch; tests: every length class)
        int tensor_workers = 0;     // workgroups of fir_tensor (0: one per compute unit; tests: few workers, several segments each)
        int max_chunk_frames = 0;   // frames per sub-batch of the *_dev entry points (0: from the 12 GiB / 1 GiB budgets; tests: small batches cross sub-batch boundaries)
        int surf_lanes = 3;         // imgfd_surf_dev: the fronts (integral image + pyramid) of a group's tiles go round-robin over this many HIP streams (1..4)
        int surf_group = 4;         // imgfd_surf_dev: tiles per group (1..16): a buffer set per tile, the latency-bound back stages (maximum test, ranking, K19) as ONE launch each per group
        int surf_rec_cap = 1 << 18; // imgfd_surf_dev: candidate records a tile's buffer holds before the tile reports -needed (tests lower it)
        int surf_sort_cap = 2048;   // imgfd_surf_dev: selected records ranked by the LDS sort; more are ranked all-pairs (tests lower it)
    } tune;
    // imgfd_surf: helper threads for the host half of K19 (surf.hip owns the object and its destructor)
    void *surf_pool = nullptr;
    void (*surf_pool_free)(void *) = nullptr;
    // imgfd_surf_dev: events between the front streams and the back stream of a batch (created on first use)
    std::vector<hipEvent_t> surf_ev;
    // imgfd_clock_probe (clock.hip): a stream of its own, a ring of (shader cycles, wall ticks) samples
    hipStream_t clk_stream = nullptr;
    unsigned long long *clk_ring = nullptr;
    int clk_n = 0, clk_rate_khz = 0;
    // in-pipeline K3 timing (imgfd_profile_k3)
    bool prof_on = false;
    std::vector<hipEvent_t> prof_ev;  // pairs
    size_t prof_used = 0;
};

#define IMGFD_HIP(ctx, call)                                                                       \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            char b_[512];                                                                          \
            snprintf(b_, sizeof b_, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            (ctx)->err = b_;                                                                       \
            return IMGFD_ERR_HIP;                                                                  \
        }                                                                                          \
    } while (0)

#define IMGFD_TRY(expr)                       \
    do {                                      \
        imgfd_status s_ = (expr);             \
        if (s_ != IMGFD_OK) return s_;        \
    } while (0)

static inline imgfd_status imgfd_fail(imgfd_ctx *ctx, imgfd_status s, const char *msg)
{
    if (ctx) ctx->err = msg;
    return s;
}

// No C++ exception may cross the C boundary: the entry points whose host side allocates (std::vector, std::thread,
// std::string) run their body through this.
template <typename F>
static inline imgfd_status imgfd_guard(imgfd_ctx *ctx, F &&body) noexcept
{
    try {
        return body();
    } catch (const std::bad_alloc &) {
        try { if (ctx) ctx->err = "out of host memory"; } catch (...) {}
        return IMGFD_ERR_OOM;
    } catch (...) {
        try { if (ctx) ctx->err = "unexpected C++ exception in the host stage"; } catch (...) {}
        return IMGFD_ERR_HIP;
    }
}

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
// The kernels index the pixels (RGB: the bytes) of one frame with 32-bit integers, like the reference's own loops (an R
// matrix holds at most 2^31 - 1 elements); larger frames are refused at the boundary instead of wrapping inside a kernel.
constexpr int64_t IMGFD_MAX_FRAME_ELEMS = ((int64_t)1 << 31) - ((int64_t)1 << 24);
static inline bool frame_fits(int64_t nx, int64_t ny, int64_t channels = 1) { return nx * ny * channels <= IMGFD_MAX_FRAME_ELEMS; }
// frames per sub-batch of a *_dev entry point: as many as fit `budget` bytes of stage planes, at least one.
// The lab switch "max_chunk_frames" (tests) lowers it so that small batches cross sub-batch boundaries too.
static inline int sub_batch_frames(const imgfd_ctx *ctx, int n_frames, size_t per_frame_bytes, size_t budget)
{
    size_t c = budget / (per_frame_bytes ? per_frame_bytes : 1);
    if (c > (size_t)n_frames) c = (size_t)n_frames;
    if (c < 1) c = 1;
    const int m = ctx->tune.max_chunk_frames;
    if (m >= 1 && (size_t)m < c) c = (size_t)m;
    return (int)c;
}
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// streaming (non-temporal) store: data the kernel itself will not read again
#define IMGFD_STREAM_STORE(value, ptr) __builtin_nontemporal_store((value), (ptr))
// a kernel's big output plane (read next by another kernel, from HBM: a batch's planes exceed every cache): streaming when
// IMGFD_NT_OUT is set for the translation unit
#ifndef IMGFD_NT_OUT
#define IMGFD_NT_OUT 0
#endif
#if IMGFD_NT_OUT
#define IMGFD_OUT_STORE(value, ptr) __builtin_nontemporal_store((value), (ptr))
#else
#define IMGFD_OUT_STORE(value, ptr) (*(ptr) = (value))
#endif
// The one place where the device build and the host-side emulator build of the tests (HIPEMU, tests/hipemu: the same
// sources compiled by g++) part over compiler-specific syntax; gfx950 builtins are emulated in tests/hipemu/hip/hip_runtime.h.
#ifdef HIPEMU
#define IMGFD_OPAQUE(x) ((void)0)
#define IMGFD_WAVES_PER_EU(lo, hi)
#define IMGFD_LDS_SPACE
typedef hipemu_u16x2 imgfd_u16x2;
#else
typedef unsigned short imgfd_u16x2 __attribute__((ext_vector_type(2)));  // two 16-bit lanes of a dword (packed instructions)
#define IMGFD_LDS_SPACE __attribute__((address_space(3)))  // a pointer that is known to address LDS
#define IMGFD_OPAQUE(x) asm volatile("" : "+v"(x))  // keeps the optimiser from rewriting the expression that produced x
// LDS (not registers) fixes the residency of the marching kernels: tell the register allocator so, or it trades VGPRs
// for an occupancy the LDS footprint can never reach (and spills)
#define IMGFD_WAVES_PER_EU(lo, hi) __attribute__((amdgpu_waves_per_eu(lo, hi)))
#endif

// ---- XCD-aware tile order.  Workgroup ids are dealt round-robin to the 8 XCDs, each with an L2 of its own: with the natural
// order, the tiles to the left and right of a tile (and the tile rows above and below it) run on OTHER XCDs and fetch the
// 128-byte lines they share with it from HBM again.  Workgroup `id` of `total` takes tile xcd * (total / 8) + ... instead: an
// XCD owns a contiguous run of the tile order, walked in step by its CUs, so shared lines are L2 hits.
__device__ __forceinline__ unsigned imgfd_xcd_tile(unsigned id, unsigned total)
{
    const unsigned q = total >> 3, r = total & 7u, xcd = id & 7u, local = id >> 3;
    return xcd * q + (xcd < r ? xcd : r) + local;
}

// ---- tile runs: the 2-D tile kernels over u8 frames (fast9_tile, gauss_grad_tile) walk `run` consecutive tiles of one
// band of rows per workgroup instead of one tile.  A 64-pixel tile row with its halo straddles two 128-byte lines; with
// one tile per workgroup the x-neighbour (another workgroup, dealt to another XCD with its own L2) fetched both lines
// again: 4-5 times the algorithmic bytes came from HBM.  Inside a run the shared lines are re-read within microseconds
// by the same CU and mostly hit its L2.
struct TileRuns {
    int tiles_x, bands, frames, run, runs_per_band;
    unsigned total;  // runs_per_band * bands * frames
};
static inline TileRuns tile_runs(int tiles_x, int bands, int frames, int run)
{
    TileRuns t;
    t.tiles_x = tiles_x; t.bands = bands; t.frames = frames;
    t.run = run < 1 ? 1 : (run > tiles_x ? tiles_x : run);
    t.runs_per_band = (tiles_x + t.run - 1) / t.run;
    t.total = (unsigned)t.runs_per_band * (unsigned)bands * (unsigned)frames;
    return t;
}
// run length for a frame batch (lab switch "tile_run" overrides: experiments)
int tile_run_length(const imgfd_ctx *ctx, int tiles_x, int bands, int frames);

// workspace: all allocations of one API call are carved from one arena; ws_reserve() guarantees
// capacity up front so no pointer handed out earlier in the call is invalidated.
imgfd_status ws_reserve(imgfd_ctx *ctx, size_t bytes);
void *ws_alloc(imgfd_ctx *ctx, size_t bytes);  // 256-byte aligned; nullptr if the reservation is exceeded
static inline void ws_reset(imgfd_ctx *ctx) { ctx->ws_used = 0; }
imgfd_status pin_reserve(imgfd_ctx *ctx, size_t bytes);
imgfd_status aux_reserve(imgfd_ctx *ctx, size_t bytes);
// the context's companion (created on first use): same device, own non-blocking stream, own workspace
imgfd_status ctx_side(imgfd_ctx *ctx, imgfd_ctx **side);
// records a profiling event on the stream when K3 profiling is on (no-op otherwise)
imgfd_status prof_mark(imgfd_ctx *ctx);

// ingest.hip: element type of a host vector handed to an entry point
enum { IMGFD_SRC_U8 = 0, IMGFD_SRC_I32 = 1, IMGFD_SRC_F32 = 2, IMGFD_SRC_F64 = 3 };
size_t upload_stage_bytes(int kind, size_t n);
imgfd_status upload_image(imgfd_ctx *ctx, const void *host, int kind, size_t n, void *d_dst);
// results widened to R's doubles in HBM (d_stage: n doubles of workspace) and copied into host_out; asynchronous on ctx->stream
imgfd_status download_widened(imgfd_ctx *ctx, const void *d_src, bool src_is_u8, size_t n, double *d_stage, double *host_out);

// ---- kernel launchers shared between translation units (device pointers, async on ctx->stream)
struct FrameGeom {
    int nx, ny;
    int n_frames;
};

// fir.hip
imgfd_status launch_gaussian(imgfd_ctx *ctx, const void *d_in, int in_is_u8, int in_pitch,
                             size_t in_frame_stride_bytes, float *d_out, int nx, int ny, int n_frames,
                             float sigma, int type, float *d_tmp);
imgfd_status launch_structure_tensor(imgfd_ctx *ctx, const float *d_Ix, const float *d_Iy, float *d_A,
                                     float *d_B, float *d_C, int nx, int ny, int n_frames, float sigma,
                                     int gauss, float *d_tmp);
// fir_tensor.hip: the marching structure-tensor kernel.  out_mode 0: A, B, C; 1: A, B, C stored as float4 rows through
// LDS; 2: Harris corner response only (d_A receives R, d_B / d_C unused).  IMGFD_ERR_UNSUPPORTED (no error message set)
// when no specialised instance serves the radius / alignment: the caller falls back.
bool tensor_fast_path(int R);
imgfd_status launch_tensor_march(imgfd_ctx *ctx, const float *d_Ix, const float *d_Iy, float *d_A, float *d_B, float *d_C,
                                 int nx, int ny, int n_frames, int R, const double *B, float k, int out_mode,
                                 unsigned char *d_tq = nullptr, float Th = 0.f);
// structure tensor + Harris response in one kernel (R plane only); false when that path does not apply
bool tensor_response_supported(const imgfd_ctx *ctx, int nx, int ny, float sigma, int gauss, int measure, const float *d_Ix, const float *d_Iy, const float *d_R);
imgfd_status launch_tensor_response(imgfd_ctx *ctx, const float *d_Ix, const float *d_Iy, float *d_R, int nx, int ny,
                                    int n_frames, float sigma, float k, unsigned char *d_tq = nullptr, float Th = 0.f);
// scratch (bytes) launch_gaussian / launch_structure_tensor need in d_tmp for `n_frames` frames
size_t gaussian_tmp_bytes(int nx, int ny, int n_frames, float sigma, int type, int planes);
// sii.hip
size_t sii_scratch_floats(int nx, int ny, float sigma);
imgfd_status launch_sii_gaussian(imgfd_ctx *ctx, const float *d_in, float *d_out, int nx, int ny, int n_frames, float sigma,
                                 float *d_cum);
// taps B[0..size-1] exactly as gaussian.cpp:307-330; returns size (= radius + 1), or -1 if more than IMGFD_MAX_TAPS
int fir_coeffs(float sigma, int precision, double *B);
// the size alone (any sigma)
int fir_size(float sigma, int precision);
// gauss_grad.hip: discrete Gaussian (radius 3) + gradient in one kernel; returns false when the fused kernel does not
// apply (other radii, image narrower than the kernel) and the caller runs the two separate stages instead
bool gauss_grad_fused_supported(int nx, int ny, float sigma, int gauss_type);
bool gauss_grad_march_supported(const void *d_in, int in_is_u8, int in_pitch, size_t in_frame_stride, const float *d_Ix,
                                const float *d_Iy, int nx, int ny);
imgfd_status launch_gauss_grad_march(imgfd_ctx *ctx, const void *d_in, int in_pitch, size_t in_frame_stride, float *d_Ix,
                                     float *d_Iy, int nx, int ny, int n_frames, const double *B, int grad_type, unsigned *d_rowcount = nullptr);
imgfd_status launch_gauss_grad_fused(imgfd_ctx *ctx, const void *d_in, int in_is_u8, int in_pitch, size_t in_frame_stride,
                                     float *d_Ix, float *d_Iy, int nx, int ny, int n_frames, float sigma, int grad_type,
                                     unsigned *d_rowcount = nullptr, bool *rowcount_cleared = nullptr);  // optional: ny counters per frame to clear on the way
// harris_subpixel.hip: compute_subpixel_precision (harris.cpp:340-381) over a compacted corner list, d_out != d_in
imgfd_status launch_harris_refine(imgfd_ctx *ctx, const float *d_R, int nx, const imgfd_corner *d_in, int64_t n, int precision,
                                  imgfd_corner *d_out);
// harris_stages.hip
imgfd_status launch_gradient(imgfd_ctx *ctx, const float *d_I, float *d_Ix, float *d_Iy, int nx, int ny,
                             int n_frames, int type);
imgfd_status launch_response(imgfd_ctx *ctx, const float *d_A, const float *d_B, const float *d_C,
                             float *d_R, int nx, int ny, int n_frames, int measure, float k);
// compact.hip: ordered (raster) stream compaction from a per-pixel bit mask
struct CompactBuffers {
    unsigned long long *mask;  // n_frames * ny * words_per_row
    unsigned *rowcount;        // n_frames * ny   (must be zero before the producer kernel runs)
    unsigned *rowoff;          // n_frames * ny
    int words_per_row;
};
size_t compact_bytes(int nx, int ny, int n_frames);
imgfd_status compact_carve(imgfd_ctx *ctx, int nx, int ny, int n_frames, CompactBuffers *cb);
imgfd_status compact_clear(imgfd_ctx *ctx, const CompactBuffers &cb, int ny, int n_frames);
// scan row counts, then scatter: kind 0 = imgfd_corner {x,y,R[y*nx+x]}, kind 1 = imgfd_point {x,y}
imgfd_status compact_emit(imgfd_ctx *ctx, const CompactBuffers &cb, int nx, int ny, int n_frames, int kind,
                          const float *d_R, void *d_out, int64_t cap, int64_t *d_counts);
imgfd_status compact_emit_abc(imgfd_ctx *ctx, const CompactBuffers &cb, int nx, int ny, int n_frames, const float *d_A,
                              const float *d_B, const float *d_C, int measure, float k, void *d_out, int64_t cap,
                              int64_t *d_counts);
// nms.hip
bool harris_resp_nms_supports(int nx, int ny, int radius);
imgfd_status launch_harris_resp_nms(imgfd_ctx *ctx, const float *d_A, const float *d_B, const float *d_C, int nx, int ny,
                                    int n_frames, int measure, float k, float Th, int radius, const CompactBuffers &cb);
imgfd_status launch_harris_nms(imgfd_ctx *ctx, const float *d_R, int nx, int ny, int n_frames, float Th,
                               int radius, const CompactBuffers &cb);
// the threshold quads of an R plane that is already in memory (stage doorway; nx % 4 == 0, d_tq: nx / 4 * ny * n_frames bytes)
imgfd_status launch_harris_threshold_quads(imgfd_ctx *ctx, const float *d_R, unsigned char *d_tq, int nx, int ny, int n_frames, float Th);
// NMS from the R plane and the threshold quads launch_tensor_response left in d_tq (the batch path; nx % 4 == 0)
imgfd_status launch_harris_nms_sparse(imgfd_ctx *ctx, const float *d_R, const unsigned char *d_tq, int nx, int ny, int n_frames,
                                      float Th, int radius, const CompactBuffers &cb);
// the tiled NMS of launch_harris_resp_nms on a materialised R plane (stage doorway; radius up to the LDS halo)
imgfd_status launch_harris_nms_tiled(imgfd_ctx *ctx, const float *d_R, int nx, int ny, int n_frames, float Th, int radius,
                                     const CompactBuffers &cb);
// fast9.hip
imgfd_status launch_fast9(imgfd_ctx *ctx, const uint8_t *d_img, int w, int h, int stride,
                          size_t frame_stride, int n_frames, int threshold, int nonmax, const CompactBuffers &cb);
// canny.hip: imgfd_canny_dev with a hook that runs on the host while the (first chunk of the) batch is being queued --
// called with the position reached (0 before the blur, 1 behind it, 2 behind gradient/NMS; canny_device); imgfd_detect_dev
// queues the other detectors from it (detect.hip: FAST-9 at position 0, Harris at 1 for small batches and 2 otherwise)
imgfd_status canny_dev_hooked(imgfd_ctx *ctx, const imgfd_frames *fr, double s, double low_thr, double high_thr, int accGrad,
                              uint8_t *d_edges, int64_t *d_counts, const std::function<imgfd_status(int)> *hook);
