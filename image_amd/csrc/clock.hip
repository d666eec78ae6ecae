// image_amd/csrc/clock.hip -- the shader clock the device runs at WHILE the library's kernels run (measurement aid of
// bench.py; nothing of the reference corresponds to it).
//
// The chip clocks to its power budget, so the same kernel takes 3-7 % longer on one box than on another and longer under an
// f64-heavy load than alone: a throughput number without the clock it was taken at cannot tell a slow box from a slow kernel.
// A probe is ONE wavefront on a stream of its own, launched beside whatever the context's streams run: it reads s_memtime
// (the shader-clock cycle counter, MI355X_MICROARCH.md: "tick = shader cycle") and s_memrealtime (the constant-rate counter
// behind wall_clock64(), rate = hipDeviceAttributeWallClockRate) at its start and `span` microseconds later and leaves the
// two differences in a ring of samples; imgfd_clock_probe_read turns them into GHz.
#include "common.h"

#define CLK_RING 4096

__global__ void __launch_bounds__(64) clock_probe_kernel(unsigned long long *__restrict__ ring, unsigned slot, unsigned long long span_ticks)
{
    if (threadIdx.x != 0) return;
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), t0 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = r0;
    while (r1 - r0 < span_ticks) {
        __builtin_amdgcn_s_sleep(32);
        r1 = __builtin_amdgcn_s_memrealtime();
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    ring[2 * slot] = t1 - t0;
    ring[2 * slot + 1] = r1 - r0;
}

extern "C" {

imgfd_status imgfd_clock_probe(imgfd_ctx *ctx, int span_us)
{
    if (!ctx || span_us < 1 || span_us > 100000) return IMGFD_ERR_INVALID;
    IMGFD_HIP(ctx, hipSetDevice(ctx->device));
    if (!ctx->clk_ring) {
        int khz = 0;
        if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, ctx->device) != hipSuccess || khz <= 0) khz = 100000;  // 100 MHz on gfx9
        ctx->clk_rate_khz = khz;
        IMGFD_HIP(ctx, hipStreamCreateWithFlags(&ctx->clk_stream, hipStreamNonBlocking));
        IMGFD_HIP(ctx, hipMalloc((void **)&ctx->clk_ring, sizeof(unsigned long long) * 2 * CLK_RING));
    }
    if (ctx->clk_n >= CLK_RING) return IMGFD_OK;  // the ring is full: later probes are dropped until the next read
    const unsigned long long ticks = (unsigned long long)span_us * (unsigned long long)ctx->clk_rate_khz / 1000ull;
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, ctx->clk_stream, ctx->clk_ring, (unsigned)ctx->clk_n, ticks);
    IMGFD_HIP(ctx, hipGetLastError());
    ctx->clk_n++;
    return IMGFD_OK;
}

imgfd_status imgfd_clock_probe_read(imgfd_ctx *ctx, double *mean_ghz, double *min_ghz, double *max_ghz, int *samples)
try {
    if (!ctx || !mean_ghz || !min_ghz || !max_ghz || !samples) return IMGFD_ERR_INVALID;
    *mean_ghz = *min_ghz = *max_ghz = 0;
    *samples = 0;
    if (!ctx->clk_ring || ctx->clk_n == 0) return IMGFD_OK;
    IMGFD_HIP(ctx, hipStreamSynchronize(ctx->clk_stream));
    std::vector<unsigned long long> h(2 * (size_t)ctx->clk_n);
    IMGFD_HIP(ctx, hipMemcpy(h.data(), ctx->clk_ring, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost));
    double sum = 0, lo = 1e30, hi = 0;
    int n = 0;
    for (int i = 0; i < ctx->clk_n; i++) {
        if (!h[2 * i + 1]) continue;
        const double ghz = (double)h[2 * i] / (double)h[2 * i + 1] * (double)ctx->clk_rate_khz * 1e-6;  // cycles per tick x ticks per second
        if (!(ghz > 0.05 && ghz < 10.0)) continue;  // a sample whose cycle counter stepped backwards between the two reads (seen once in ~10^3 probes)
        sum += ghz; lo = std::min(lo, ghz); hi = std::max(hi, ghz); n++;
    }
    ctx->clk_n = 0;
    if (n) { *mean_ghz = sum / n; *min_ghz = lo; *max_ghz = hi; *samples = n; }
    return IMGFD_OK;
} catch (...) {
    return imgfd_fail(ctx, IMGFD_ERR_OOM, "imgfd_clock_probe_read: out of host memory");
}

}  // extern "C"
