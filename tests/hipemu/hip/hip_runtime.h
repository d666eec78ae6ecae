// tests/hipemu/hip/hip_runtime.h -- TEST INFRASTRUCTURE ONLY.
//
// A tiny single-threaded HIP *emulator* for g++: it lets the unmodified kernel sources of
// image_amd/csrc/*.hip be compiled for the host (g++ -x c++ -Itests/hipemu) so that kernel LOGIC
// (tiling, halos, border rules, ordered compaction) can be parity-checked against the oracle in
// the GPU-less build container (`pytest -m "not gpu"`).  It is never used by the product:
// image_amd loads only libimgfd.so (hipcc, gfx950) and fails loudly without it.
//
// Model: blocks run one after another; the threads of a block are ucontext fibers scheduled
// round-robin; __syncthreads() and the wave64 cross-lane intrinsics are rendezvous points that
// yield until every (live) participant has arrived.  Wave intrinsics must be executed by all 64
// lanes of a wave (convergent code), which is also what the kernels need on real hardware.
#pragma once
#include <time.h>
#include <ucontext.h>

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <set>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __constant__
#define __forceinline__ inline __attribute__((always_inline))
#define __noinline__ __attribute__((noinline))
#define __shared__ static
#define __launch_bounds__(...)
#define HIP_DYNAMIC_SHARED(type, var) type *var = reinterpret_cast<type *>(hipemu::dyn_smem());
#define HIPEMU 1

struct dim3 {
    unsigned x, y, z;
    constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint3_ { unsigned x, y, z; };

// ------------------------------------------------------------------ vector types
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct double2 { double x, y; };
struct int2 { int x, y; };
struct int4 { int x, y, z, w; };
struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
struct uchar4 { unsigned char x, y, z, w; };
struct uchar2 { unsigned char x, y; };
struct ushort4 { unsigned short x, y, z, w; };
static inline float2 make_float2(float x, float y) { return {x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return {x, y, z, w}; }
static inline double2 make_double2(double x, double y) { return {x, y}; }
static inline int2 make_int2(int x, int y) { return {x, y}; }
static inline int4 make_int4(int x, int y, int z, int w) { return {x, y, z, w}; }
static inline uint2 make_uint2(unsigned x, unsigned y) { return {x, y}; }
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return {x, y, z, w}; }
static inline uchar4 make_uchar4(unsigned char x, unsigned char y, unsigned char z, unsigned char w) { return {x, y, z, w}; }

// ------------------------------------------------------------------ runtime state
namespace hipemu {

struct Fiber {
    ucontext_t ctx;
    char *stack = nullptr;
    bool done = true;
    uint3_ tid{};
    unsigned linear = 0;
    unsigned long wave_ops = 0;   // number of wave rendezvous this lane has completed
    unsigned long block_ops = 0;  // number of block barriers this thread has completed
};

struct Wave {
    unsigned long arrived = 0;    // monotonic count of lane arrivals at wave rendezvous
    unsigned nlanes = 0;
    unsigned long long slots[2][64];
};

struct State {
    ucontext_t sched;
    std::vector<Fiber> fibers;
    std::vector<Wave> waves;
    Fiber *cur = nullptr;
    unsigned nthreads = 0;
    unsigned long block_arrived = 0;
    unsigned long progress = 0;
    std::function<void()> body;
    std::vector<char> dyn;
    dim3 grid, block;
    uint3_ bid{};
};

inline State &S() { static State s; return s; }
inline void *dyn_smem() { return S().dyn.data(); }

static const size_t kStack = 256 * 1024;

inline void yield() { State &s = S(); swapcontext(&s.cur->ctx, &s.sched); }

inline void trampoline() {
    State &s = S();
    s.body();
    s.cur->done = true;
    s.progress++;
    swapcontext(&s.cur->ctx, &s.sched);
}

inline void run_block() {
    State &s = S();
    unsigned n = s.nthreads;
    if (s.fibers.size() < n) s.fibers.resize(n);
    unsigned nw = (n + 63) / 64;
    s.waves.assign(nw, Wave());
    for (unsigned w = 0; w < nw; w++) s.waves[w].nlanes = std::min(64u, n - 64 * w);
    s.block_arrived = 0;
    for (unsigned t = 0; t < n; t++) {
        Fiber &f = s.fibers[t];
        if (!f.stack) f.stack = (char *)malloc(kStack);
        f.done = false; f.linear = t; f.wave_ops = 0; f.block_ops = 0;
        f.tid.x = t % s.block.x; f.tid.y = (t / s.block.x) % s.block.y; f.tid.z = t / (s.block.x * s.block.y);
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = f.stack; f.ctx.uc_stack.ss_size = kStack; f.ctx.uc_link = nullptr;
        makecontext(&f.ctx, (void (*)())trampoline, 0);
    }
    unsigned live = n;
    while (live) {
        unsigned long before = s.progress;
        live = 0;
        for (unsigned t = 0; t < n; t++) {
            Fiber &f = s.fibers[t];
            if (f.done) continue;
            s.cur = &f;
            swapcontext(&s.sched, &f.ctx);
            if (!f.done) live++;
        }
        if (live && s.progress == before) {
            fprintf(stderr, "hipemu: deadlock in block (%u,%u,%u): %u threads wait at a barrier or wave "
                            "intrinsic the others never reach (divergent __syncthreads/__ballot/__shfl?)\n",
                    s.bid.x, s.bid.y, s.bid.z, live);
            abort();
        }
    }
}

inline void launch(dim3 grid, dim3 block, size_t shmem, std::function<void()> body) {
    State &s = S();
    s.grid = grid; s.block = block; s.body = std::move(body);
    s.nthreads = block.x * block.y * block.z;
    if (s.dyn.size() < shmem + 16) s.dyn.resize(shmem + 16);
    for (unsigned z = 0; z < grid.z; z++)
        for (unsigned y = 0; y < grid.y; y++)
            for (unsigned x = 0; x < grid.x; x++) {
                s.bid = {x, y, z};
                run_block();
            }
}

// block barrier: wait until every thread of the block has arrived (exited threads never do: the
// kernels keep barriers outside divergent exits, as HIP requires)
inline void block_barrier() {
    State &s = S();
    Fiber *f = s.cur;
    s.block_arrived++;
    s.progress++;
    unsigned long need = (f->block_ops + 1) * (unsigned long)s.nthreads;
    while (s.block_arrived < need) yield();
    f->block_ops++;
}

// wave rendezvous with a 64-bit payload per lane; returns pointer to the 64 slots of this op
inline const unsigned long long *wave_exchange(unsigned long long v) {
    State &s = S();
    Fiber *f = s.cur;
    Wave &w = s.waves[f->linear / 64];
    unsigned par = f->wave_ops & 1;
    w.slots[par][f->linear % 64] = v;
    w.arrived++;
    s.progress++;
    unsigned long need = (f->wave_ops + 1) * (unsigned long)w.nlanes;
    while (w.arrived < need) yield();
    f->wave_ops++;
    return w.slots[par];
}
inline unsigned wave_nlanes() { State &s = S(); return s.waves[s.cur->linear / 64].nlanes; }

struct IdxProxyX { operator unsigned() const; };
}  // namespace hipemu

// threadIdx/blockIdx/blockDim/gridDim as objects whose members read the current fiber
struct hipemu_tid_t {
    struct X { operator unsigned() const { return hipemu::S().cur->tid.x; } } x;
    struct Y { operator unsigned() const { return hipemu::S().cur->tid.y; } } y;
    struct Z { operator unsigned() const { return hipemu::S().cur->tid.z; } } z;
};
struct hipemu_bid_t {
    struct X { operator unsigned() const { return hipemu::S().bid.x; } } x;
    struct Y { operator unsigned() const { return hipemu::S().bid.y; } } y;
    struct Z { operator unsigned() const { return hipemu::S().bid.z; } } z;
};
struct hipemu_bdim_t {
    struct X { operator unsigned() const { return hipemu::S().block.x; } } x;
    struct Y { operator unsigned() const { return hipemu::S().block.y; } } y;
    struct Z { operator unsigned() const { return hipemu::S().block.z; } } z;
};
struct hipemu_gdim_t {
    struct X { operator unsigned() const { return hipemu::S().grid.x; } } x;
    struct Y { operator unsigned() const { return hipemu::S().grid.y; } } y;
    struct Z { operator unsigned() const { return hipemu::S().grid.z; } } z;
};
static hipemu_tid_t threadIdx;
static hipemu_bid_t blockIdx;
static hipemu_bdim_t blockDim;
static hipemu_gdim_t gridDim;
static const int warpSize = 64;

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    hipemu::launch(dim3(grid), dim3(block), (shmem), [&]() { kernel(__VA_ARGS__); })

// ------------------------------------------------------------------ device intrinsics
static inline void __syncthreads() { hipemu::block_barrier(); }
static inline void __threadfence() {}
static inline void __threadfence_block() {}
static inline unsigned __lane_id() { return hipemu::S().cur->linear % 64; }

static inline unsigned long long __ballot(int pred) {
    const unsigned long long *s = hipemu::wave_exchange(pred ? 1ull : 0ull);
    unsigned long long m = 0;
    unsigned n = hipemu::wave_nlanes();
    for (unsigned i = 0; i < n; i++) m |= (s[i] & 1ull) << i;
    return m;
}
static inline int __any(int pred) { return __ballot(pred) != 0; }
static inline int __all(int pred) {
    unsigned n = hipemu::wave_nlanes();
    unsigned long long full = n == 64 ? ~0ull : ((1ull << n) - 1);
    return __ballot(pred) == full;
}
template <typename T> static inline T hipemu_shfl_idx(T v, int src) {
    static_assert(sizeof(T) <= 8, "shfl payload");
    unsigned long long raw = 0;
    memcpy(&raw, &v, sizeof(T));
    const unsigned long long *s = hipemu::wave_exchange(raw);
    T out;
    memcpy(&out, &s[src & 63], sizeof(T));
    return out;
}
template <typename T> static inline T __shfl(T v, int src, int width = 64) {
    int lane = (int)__lane_id();
    int base = lane & ~(width - 1);
    return hipemu_shfl_idx(v, base + (src & (width - 1)));
}
template <typename T> static inline T __shfl_up(T v, unsigned d, int width = 64) {
    int lane = (int)__lane_id();
    int base = lane & ~(width - 1);
    int src = lane - (int)d;
    return hipemu_shfl_idx(v, src < base ? lane : src);
}
template <typename T> static inline T __shfl_down(T v, unsigned d, int width = 64) {
    int lane = (int)__lane_id();
    int base = lane & ~(width - 1);
    int src = lane + (int)d;
    return hipemu_shfl_idx(v, src >= base + width ? lane : src);
}
template <typename T> static inline T __shfl_xor(T v, int m, int width = 64) {
    int lane = (int)__lane_id();
    int base = lane & ~(width - 1);
    int src = lane ^ m;
    return hipemu_shfl_idx(v, src >= base + width ? lane : src);
}
static inline unsigned __umul24(unsigned a, unsigned b) { return (unsigned)((unsigned long long)(a & 0xffffffu) * (unsigned long long)(b & 0xffffffu)); }  // v_mul_u32_u24: low 32 bits of the product of the low 24 bits
static inline int __mul24(int a, int b) { return (int)((long long)((a << 8) >> 8) * (long long)((b << 8) >> 8)); }  // v_mul_i32_i24: the low 24 bits of each operand, sign-extended
static inline int __popc(unsigned x) { return __builtin_popcount(x); }
static inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
static inline int __ffs(int x) { return __builtin_ffs(x); }
static inline int __ffsll(long long x) { return __builtin_ffsll(x); }
static inline int __clz(int x) { return x ? __builtin_clz((unsigned)x) : 32; }
static inline int __clzll(long long x) { return x ? __builtin_clzll((unsigned long long)x) : 64; }
static inline unsigned __brev(unsigned x) {
    unsigned r = 0;
    for (int i = 0; i < 32; i++) r |= ((x >> i) & 1u) << (31 - i);
    return r;
}

template <typename T> static inline T atomicAdd(T *p, T v) { T o = *p; *p = o + v; return o; }
template <typename T> static inline T atomicSub(T *p, T v) { T o = *p; *p = o - v; return o; }
template <typename T> static inline T atomicMax(T *p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <typename T> static inline T atomicMin(T *p, T v) { T o = *p; if (v < o) *p = v; return o; }
template <typename T> static inline T atomicOr(T *p, T v) { T o = *p; *p = o | v; return o; }
template <typename T> static inline T atomicAnd(T *p, T v) { T o = *p; *p = o & v; return o; }
template <typename T> static inline T atomicExch(T *p, T v) { T o = *p; *p = v; return o; }
template <typename T> static inline T atomicCAS(T *p, T c, T v) { T o = *p; if (o == c) *p = v; return o; }

// HIP device code sees min/max overloads for scalars
static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
static inline unsigned min(unsigned a, unsigned b) { return a < b ? a : b; }
static inline unsigned max(unsigned a, unsigned b) { return a > b ? a : b; }
static inline long min(long a, long b) { return a < b ? a : b; }
static inline long max(long a, long b) { return a > b ? a : b; }
static inline long long min(long long a, long long b) { return a < b ? a : b; }
static inline long long max(long long a, long long b) { return a > b ? a : b; }
static inline unsigned long min(unsigned long a, unsigned long b) { return a < b ? a : b; }
static inline unsigned long max(unsigned long a, unsigned long b) { return a > b ? a : b; }
static inline float min(float a, float b) { return fminf(a, b); }
static inline float max(float a, float b) { return fmaxf(a, b); }
static inline double min(double a, double b) { return fmin(a, b); }
static inline double max(double a, double b) { return fmax(a, b); }

// wave-uniform values are what kernels pass here: every emulated lane already holds the same value
static inline int __builtin_amdgcn_readfirstlane(int v) { return v; }
// v_readlane_b32: the value lane `src` holds (src wave-uniform) -- a shuffle every lane of the wave takes part in
static inline int __builtin_amdgcn_readlane(int v, int src) { return __shfl(v, src); }

// ---- gfx950 builtins the product sources use unconditionally (emulated here, so that the kernels carry no test branches)
static inline double __builtin_amdgcn_rsq(double x) { return 1.0 / sqrt(x); }  // v_rsq_f64: a seed, refined by the caller
static inline void __builtin_amdgcn_s_sleep(int) {}
// s_memtime / s_memrealtime (clock.hip): one wall counter that advances per read, 21 shader cycles per tick ("2.1 GHz" at 100 MHz)
inline unsigned long long &hipemu_wall_ticks() { static unsigned long long t = 0; return t; }
static inline unsigned long long __builtin_amdgcn_s_memrealtime() { return hipemu_wall_ticks() += 100; }
static inline unsigned long long __builtin_amdgcn_s_memtime() { return 21 * hipemu_wall_ticks(); }
static inline void __builtin_amdgcn_s_setprio(int) {}
static inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
static inline float __builtin_amdgcn_rsqf(float x) { return (float)(1.0 / sqrt((double)x)); }  // v_rsq_f32 (1 ulp on the device)
// v_perm_b32: byte k of the result is picked by selector byte k: 0..3 = bytes of b, 4..7 = bytes of a, 0x0c = 0x00
static inline unsigned __builtin_amdgcn_perm(unsigned a, unsigned b, unsigned sel) {
    unsigned r = 0;
    for (int k = 0; k < 4; k++) {
        const unsigned c = (sel >> (8 * k)) & 0xffu;
        unsigned byte = 0;
        if (c < 4) byte = (b >> (8 * c)) & 0xffu;
        else if (c < 8) byte = (a >> (8 * (c - 4))) & 0xffu;
        else if (c != 0x0c) { fprintf(stderr, "hipemu: v_perm_b32 selector 0x%02x not emulated\n", c); abort(); }
        r |= byte << (8 * k);
    }
    return r;
}
// v_mov_b32_dpp with bound_ctrl: wave_shr:1 (0x138: lane i reads lane i-1) and wave_shl:1 (0x130: lane i reads lane i+1);
// a lane without a source reads 0
// (returns int like the device builtin: a result OR-ed into a wider word without a cast sign-extends here as it does there)
static inline int __builtin_amdgcn_mov_dpp(unsigned v, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
    if (ctrl >= 0 && ctrl <= 0xff && row_mask == 0xf && bank_mask == 0xf) {  // quad_perm: lane i reads lane (i & ~3) + perm[i & 3] of its quad
        const int l = (int)__lane_id();
        const unsigned long long *s = hipemu::wave_exchange(v);
        return (int)(unsigned)s[(l & ~3) + ((ctrl >> (2 * (l & 3))) & 3)];
    }
    if ((ctrl != 0x138 && ctrl != 0x130) || row_mask != 0xf || bank_mask != 0xf || !bound_ctrl) {
        fprintf(stderr, "hipemu: DPP control 0x%x not emulated\n", ctrl);
        abort();
    }
    const int lane = (int)__lane_id();
    const unsigned r = ctrl == 0x138 ? __shfl_up(v, 1) : __shfl_down(v, 1);
    return (int)((ctrl == 0x138 ? lane == 0 : lane == 63) ? 0u : r);
}
// streaming store hint
#define __builtin_nontemporal_store(value, ptr) (*(ptr) = (value))
// two unsigned 16-bit lanes in a dword (the device build's `unsigned short ext_vector_type(2)`) with the element-wise
// builtins the kernels use on it (v_pk_sub_u16 clamp, v_pk_max_u16, v_pk_min_u16)
struct hipemu_u16x2 { unsigned short x, y; };
static inline hipemu_u16x2 __builtin_elementwise_sub_sat(hipemu_u16x2 a, hipemu_u16x2 b) {
    return hipemu_u16x2{(unsigned short)(a.x > b.x ? a.x - b.x : 0), (unsigned short)(a.y > b.y ? a.y - b.y : 0)};
}
static inline hipemu_u16x2 __builtin_elementwise_max(hipemu_u16x2 a, hipemu_u16x2 b) {
    return hipemu_u16x2{a.x > b.x ? a.x : b.x, a.y > b.y ? a.y : b.y};
}
static inline hipemu_u16x2 __builtin_elementwise_min(hipemu_u16x2 a, hipemu_u16x2 b) {
    return hipemu_u16x2{a.x < b.x ? a.x : b.x, a.y < b.y ? a.y : b.y};
}
// raw buffer addressing: resource = base pointer (+ size, flags: ignored), store at base + voffset + soffset
struct __amdgpu_buffer_rsrc_t { char *base; };
static inline __amdgpu_buffer_rsrc_t __builtin_amdgcn_make_buffer_rsrc(void *base, short, int, int) { return __amdgpu_buffer_rsrc_t{(char *)base}; }
static inline void __builtin_amdgcn_raw_buffer_store_b32(unsigned data, __amdgpu_buffer_rsrc_t r, int voffset, int soffset, int) {
    memcpy(r.base + (size_t)(unsigned)voffset + (size_t)(unsigned)soffset, &data, 4);
}

struct hipemu_v4u { unsigned x, y, z, w; };
static inline hipemu_v4u __builtin_amdgcn_raw_buffer_load_b128(__amdgpu_buffer_rsrc_t r, int voffset, int soffset, int) {
    hipemu_v4u v;
    memcpy(&v, r.base + (size_t)(unsigned)voffset + (size_t)(unsigned)soffset, 16);
    return v;
}
// wavefront-scope fence: no instruction on the device; wave barrier: on the device a scheduling fence (the wave's lanes run in
// lock step), here the rendezvous of the wave's 64 fibers that lock step stands for
#define __builtin_amdgcn_fence(order, scope) ((void)0)
// scoped atomic load: one thread at a time runs here
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __hip_atomic_load(p, order, scope) (*(p))
#define __hip_atomic_store(p, v, order, scope) (*(p) = (v))
#define __hip_atomic_fetch_add(p, v, order, scope) atomicAdd((p), (v))
#define __hip_atomic_store(p, v, order, scope) (*(p) = (v))
static inline void __builtin_amdgcn_wave_barrier() { (void)hipemu::wave_exchange(0); }
static inline unsigned __builtin_amdgcn_raw_buffer_load_b32(__amdgpu_buffer_rsrc_t r, int voffset, int soffset, int) {
    unsigned v;
    memcpy(&v, r.base + (size_t)(unsigned)voffset + (size_t)(unsigned)soffset, 4);
    return v;
}

// ------------------------------------------------------------------ host API
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNoDevice = 100 };
typedef struct hipemu_stream *hipStream_t;
typedef struct hipemu_event { double t; } *hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
struct hipDeviceProp_t {
    char name[256]; char gcnArchName[256]; int multiProcessorCount; size_t totalGlobalMem; int warpSize;
    size_t sharedMemPerBlock; int maxThreadsPerBlock; int clockRate;
    int cooperativeLaunch;  // 0: the emulator runs the blocks of a launch one after another -- they cannot wait for each other
};
#define hipHostMallocDefault 0
#define hipStreamNonBlocking 1
#define hipEventDefault 0
#define hipEventDisableTiming 2
enum hipMemoryType { hipMemoryTypeUnregistered = 0, hipMemoryTypeHost = 1, hipMemoryTypeDevice = 2 };
struct hipPointerAttribute_t { hipMemoryType type; };
// pinned allocations (hipHostMalloc) are remembered so that hipPointerGetAttributes can tell them from pageable memory
inline std::mutex &hipemu_pinned_lock() { static std::mutex m; return m; }
inline std::set<const void *> &hipemu_pinned() { static std::set<const void *> s; return s; }

static inline const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "hipemu error"; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipPeekAtLastError() { return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int) {
    memset(p, 0, sizeof(*p));
    strcpy(p->name, "hipemu (host)"); strcpy(p->gcnArchName, "hipemu");
    p->multiProcessorCount = 256; p->totalGlobalMem = 1ull << 34; p->warpSize = 64;
    p->sharedMemPerBlock = 160 * 1024; p->maxThreadsPerBlock = 1024; p->clockRate = 2400000;
    return hipSuccess;
}
enum hipDeviceAttribute_t { hipDeviceAttributeWallClockRate = 1 };
static inline hipError_t hipDeviceGetAttribute(int *v, hipDeviceAttribute_t, int) { *v = 100000; return hipSuccess; }  // kHz
static inline hipError_t hipMalloc(void **p, size_t n) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
template <typename T> static inline hipError_t hipMalloc(T **p, size_t n) { return hipMalloc((void **)p, n); }
static inline hipError_t hipFree(void *p) { free(p); return hipSuccess; }
static inline hipError_t hipHostMalloc(void **p, size_t n, unsigned = 0) {
    hipError_t e = hipMalloc(p, n);
    if (e == hipSuccess) { std::lock_guard<std::mutex> g(hipemu_pinned_lock()); hipemu_pinned().insert(*p); }
    return e;
}
template <typename T> static inline hipError_t hipHostMalloc(T **p, size_t n, unsigned f = 0) { return hipHostMalloc((void **)p, n, f); }
static inline hipError_t hipHostFree(void *p) {
    { std::lock_guard<std::mutex> g(hipemu_pinned_lock()); hipemu_pinned().erase(p); }
    free(p); return hipSuccess;
}
static inline hipError_t hipPointerGetAttributes(hipPointerAttribute_t *a, const void *p) {
    std::lock_guard<std::mutex> g(hipemu_pinned_lock());
    if (!hipemu_pinned().count(p)) return hipErrorInvalidValue;
    a->type = hipMemoryTypeHost; return hipSuccess;
}
static inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { if (n) memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind k, hipStream_t = 0) { return hipMemcpy(d, s, n, k); }
static inline hipError_t hipMemcpy2DAsync(void *d, size_t dp, const void *s, size_t sp, size_t w, size_t h, hipMemcpyKind, hipStream_t = 0) {
    for (size_t r = 0; r < h; r++) memmove((char *)d + r * dp, (const char *)s + r * sp, w);
    return hipSuccess;
}
static inline hipError_t hipMemset(void *d, int v, size_t n) { if (n) memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t = 0) { return hipMemset(d, v, n); }
static inline hipError_t hipStreamCreate(hipStream_t *s) { *s = nullptr; return hipSuccess; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = nullptr; return hipSuccess; }
static inline hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned, int) { *s = nullptr; return hipSuccess; }
static inline hipError_t hipDeviceGetStreamPriorityRange(int *least, int *greatest) { *least = 0; *greatest = -1; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline double hipemu_now() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }
static inline hipError_t hipEventCreate(hipEvent_t *e) { *e = new hipemu_event{0}; return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return hipEventCreate(e); }
static inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t = 0) { e->t = hipemu_now(); return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned = 0) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(b->t - a->t); return hipSuccess; }
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
static inline hipError_t hipFuncSetAttribute(const void *, hipFuncAttribute, int) { return hipSuccess; }
// stream capture / graphs: not emulated -- hipStreamBeginCapture refuses, the product falls back to eager launches
typedef struct hipemuGraph *hipGraph_t;
typedef struct hipemuGraphExec *hipGraphExec_t;
enum hipStreamCaptureMode { hipStreamCaptureModeGlobal = 0, hipStreamCaptureModeThreadLocal = 1, hipStreamCaptureModeRelaxed = 2 };
static inline hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) { return hipErrorInvalidValue; }
static inline hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t *g) { *g = nullptr; return hipErrorInvalidValue; }
static inline hipError_t hipGraphInstantiate(hipGraphExec_t *e, hipGraph_t, void *, void *, size_t) { *e = nullptr; return hipErrorInvalidValue; }
static inline hipError_t hipGraphDestroy(hipGraph_t) { return hipSuccess; }
static inline hipError_t hipGraphExecDestroy(hipGraphExec_t) { return hipSuccess; }
static inline hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return hipErrorInvalidValue; }
static inline hipError_t hipMemGetInfo(size_t *f, size_t *t) { *f = 1ull << 33; *t = 1ull << 34; return hipSuccess; }
static inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int *n, const void *, int, size_t) { *n = 2; return hipSuccess; }
static inline hipError_t hipExtStreamGetCUMask(hipStream_t, uint32_t, uint32_t *) { return hipErrorInvalidValue; }
