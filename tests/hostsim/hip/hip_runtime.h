// tests/hostsim/hip/hip_runtime.h — TEST INFRASTRUCTURE.  A host stand-in for <hip/hip_runtime.h> that lets the product's own kernel
// sources (realism-effects_amd/csrc/*.hip, unmodified except for the six textual substitutions listed in tests/hostsim/Makefile) be
// compiled for x86 and executed thread by thread on the CPU: `librfx_hostsim.so` exports the same C ABI as librfx_hip.so, so the `-m gpu`
// tests can exercise the kernels' LOGIC (indexing, tiles and aprons, launch shapes, the C ABI's state handling) in a container without a
// GPU.  It is NOT a fallback: nothing under realism-effects_amd/ knows it exists, it is built and loaded only by tests that ask for it
// (RFX_HIP_LIB), it is orders of magnitude slower than the device, and the hardware transcendentals are replaced by libm (so it says nothing about the
// device's bits — only about the program's structure).
//
// Execution model: hipLaunchKernelGGL runs the blocks of a grid on OpenMP threads; inside a block every thread is a FIBER with its own stack
// (hostsim.cpp: a cooperative scheduler, one context switch = six pushes and a stack-pointer swap).  A fiber runs until it reaches a
// synchronisation point — __syncthreads(), a wave shuffle, __ballot — and yields; a wave operation completes when every lane of the
// wavefront (64 consecutive linear thread ids) is blocked or finished (lanes that are not AT the operation do not take part: the
// device's exec mask), a barrier when every unfinished thread of the block waits at it.  That is the device's semantics for any valid
// kernel — no idempotence requirement on what a thread did before a point (the round-1..3 simulator REPLAYED threads from the top and
// needed one), so LDS may be re-used across phases and counters may be bumped with atomics.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <type_traits>

#define __host__
#define __device__
#define __global__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define RFX_KERNARGS_IN_LOOP(A) (A)  /* an optimiser fence of the device build (rfx_device.h): the argument block itself */
#define RFX_WAVE_JOIN() ((void)hostsim_wave_exchange(HOSTSIM_JOIN, 0, 0))  /* the reconvergence point after a divergent region (rfx_device.h) */
#define RFX_WAVES_PER_EU(n)  /* a register-allocation hint of the device compiler: nothing to simulate */
#define __constant__ static
#define __shared__ static thread_local
#define __restrict__

// ---------------------------------------------------------------- vector types
struct alignas(8) float2 { float x, y; };
struct float3 { float x, y, z; };
struct alignas(16) float4 { float x, y, z, w; };
struct alignas(8) uint2 { unsigned int x, y; };
struct alignas(16) uint4 { unsigned int x, y, z, w; };
struct alignas(8) int2 { int x, y; };
struct uchar4 { unsigned char x, y, z, w; };
struct dim3 {
    unsigned int x, y, z;
    dim3(unsigned int x_ = 1, unsigned int y_ = 1, unsigned int z_ = 1) : x(x_), y(y_), z(z_) {}
};
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline float3 make_float3(float x, float y, float z) { return float3{x, y, z}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline uint2 make_uint2(unsigned int x, unsigned int y) { return uint2{x, y}; }
static inline uint4 make_uint4(unsigned int x, unsigned int y, unsigned int z, unsigned int w) { return uint4{x, y, z, w}; }
static inline int2 make_int2(int x, int y) { return int2{x, y}; }
static inline uchar4 make_uchar4(unsigned char x, unsigned char y, unsigned char z, unsigned char w) { return uchar4{x, y, z, w}; }

// ---------------------------------------------------------------- execution state (hostsim.cpp)
struct hostsim_idx { unsigned int x, y, z; };
extern thread_local hostsim_idx threadIdx, blockIdx, blockDim, gridDim;
extern thread_local unsigned char *hostsim_lds;  // dynamic shared memory of the running block
static inline unsigned int hostsim_tid() { return threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z); }
enum { HOSTSIM_BALLOT = 1, HOSTSIM_SHFL_XOR = 2, HOSTSIM_READFIRST = 3, HOSTSIM_SHFL = 4, HOSTSIM_PERMUTE = 5, HOSTSIM_BPERMUTE = 6, HOSTSIM_JOIN = 7 };
void hostsim_barrier_wait();                                                         // the running fiber waits for its block
unsigned long long hostsim_wave_exchange(int kind, unsigned long long payload, int arg);  // ... for its wavefront; returns the lane's result
void hostsim_run_block(unsigned int nthreads, unsigned int bx, unsigned int by, void (*call)(void *), void *ctx);
static inline void __syncthreads() { hostsim_barrier_wait(); }
static inline float __shfl_xor(float v, int lane_mask) {  // wave64: the partner is lane ^ mask of the same wavefront (own value if it does not take part)
    unsigned int u;
    std::memcpy(&u, &v, 4);
    u = (unsigned int)hostsim_wave_exchange(HOSTSIM_SHFL_XOR, u, lane_mask);
    std::memcpy(&v, &u, 4);
    return v;
}
static inline int __shfl_xor(int v, int lane_mask) { return (int)(unsigned int)hostsim_wave_exchange(HOSTSIM_SHFL_XOR, (unsigned int)v, lane_mask); }
static inline int __shfl(int v, int src_lane) { return (int)(unsigned int)hostsim_wave_exchange(HOSTSIM_SHFL, (unsigned int)v, src_lane & 63); }
// HIP's __ballot: the TRUE wave-wide ballot (bit i = lane i's predicate; lanes that are not here contribute 0)
static inline unsigned long long __ballot(int p) { return hostsim_wave_exchange(HOSTSIM_BALLOT, p != 0, 0); }
static inline int __builtin_amdgcn_readfirstlane(int v) { return (int)(unsigned int)hostsim_wave_exchange(HOSTSIM_READFIRST, (unsigned int)v, 0); }  // the first lane that is here
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
// ds_permute_b32 (push): lane i sends `data` to lane (addr / 4) % 64; every lane gets what was sent to it (0 if nothing was; the highest sender wins)
static inline int __builtin_amdgcn_ds_permute(int addr, int data) { return (int)(unsigned int)hostsim_wave_exchange(HOSTSIM_PERMUTE, (unsigned int)data, (addr >> 2) & 63); }
// ds_bpermute_b32 (pull): lane i reads `data` of lane (addr / 4) % 64 (0 when that lane is not here)
static inline int __builtin_amdgcn_ds_bpermute(int addr, int data) { return (int)(unsigned int)hostsim_wave_exchange(HOSTSIM_BPERMUTE, (unsigned int)data, (addr >> 2) & 63); }
// v_mov_b32_dpp wave_shr:1 (0x138: lane i <- lane i - 1) / wave_shl:1 (0x130: lane i <- lane i + 1), bound_ctrl 0: the end lane keeps `old`
static inline int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int /*row_mask*/, int /*bank_mask*/, bool /*bound_ctrl*/) {
    const int lane = (int)(hostsim_tid() & 63u);
    const int from = ctrl == 0x138 ? lane - 1 : (ctrl == 0x130 ? lane + 1 : -1000);
    if (from == -1000) { std::fprintf(stderr, "hostsim: DPP control 0x%x is not modelled\n", ctrl); std::abort(); }
    const int got = (int)(unsigned int)hostsim_wave_exchange(HOSTSIM_SHFL, (unsigned int)src, (from + 64) & 63);  // (every lane takes part in the exchange)
    return (from < 0 || from > 63) ? old : got;
}
// v_mbcnt_lo/hi: bits of `mask` below the calling lane (+ base)
static inline unsigned int __builtin_amdgcn_mbcnt_lo(unsigned int mask, unsigned int base) {
    const unsigned int lane = hostsim_tid() & 63u;
    return base + (unsigned int)__builtin_popcount(lane >= 32 ? mask : (mask & ((1u << lane) - 1u)));
}
static inline unsigned int __builtin_amdgcn_mbcnt_hi(unsigned int mask, unsigned int base) {
    const unsigned int lane = hostsim_tid() & 63u;
    return base + (lane > 32 ? (unsigned int)__builtin_popcount(mask & ((1u << (lane - 32)) - 1u)) : 0u);
}
extern size_t hostsim_last_shmem;  // dynamic shared memory the most recent launch asked for (tests read it: rfx_hostsim_last_dynamic_lds)
template <class F>
static void hostsim_launch(dim3 grid, dim3 block, size_t shmem, F body) {
    const long nblocks = (long)grid.x * grid.y * grid.z;
    hostsim_last_shmem = shmem;
#pragma omp parallel for schedule(dynamic, 4)
    for (long b = 0; b < nblocks; b++) {
        static thread_local unsigned char *lds = nullptr;
        static thread_local size_t lds_size = 0;
        if (shmem > lds_size) { std::free(lds); lds = (unsigned char *)std::aligned_alloc(64, (shmem + 63) & ~(size_t)63); lds_size = shmem; }
        hostsim_lds = lds;
        gridDim = {grid.x, grid.y, grid.z};
        blockDim = {block.x, block.y, block.z};
        blockIdx = {(unsigned int)(b % grid.x), (unsigned int)((b / grid.x) % grid.y), (unsigned int)(b / ((long)grid.x * grid.y))};
        F *bp = &body;
        hostsim_run_block(block.x * block.y * block.z, block.x, block.y, [](void *c) { (*(F *)c)(); }, (void *)bp);
    }
}
// the kernel name may arrive parenthesised (`(k3_tiled<T, C>)`: a template-id with a comma) or bare: strip one pair of parentheses if present
#define HOSTSIM_EXTRACT(...) HOSTSIM_EXTRACT __VA_ARGS__
#define HOSTSIM_NOTHING_HOSTSIM_EXTRACT
#define HOSTSIM_PASTE(x, ...) x##__VA_ARGS__
#define HOSTSIM_EVAL_PASTE(x, ...) HOSTSIM_PASTE(x, __VA_ARGS__)
#define HOSTSIM_STRIP(x) HOSTSIM_EVAL_PASTE(HOSTSIM_NOTHING_, HOSTSIM_EXTRACT x)
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    hostsim_launch((grid), (block), (size_t)(shmem), [&]() { HOSTSIM_STRIP(kernel)(__VA_ARGS__); })

// ---------------------------------------------------------------- device intrinsics (hardware approximations -> libm)
static inline float __builtin_amdgcn_exp2f(float x) { return std::exp2(x); }
static inline float __builtin_amdgcn_logf(float x) { return std::log2(x); }
static inline float __builtin_amdgcn_sqrtf(float x) { return std::sqrt(x); }
static inline float __builtin_amdgcn_rsqf(float x) { return 1.0f / std::sqrt(x); }
static inline float __builtin_amdgcn_rcpf(float x) { return 1.0f / x; }
static inline float __builtin_amdgcn_sinf(float rev) { return (float)std::sin((double)rev * 6.283185307179586476925); }
static inline float __builtin_amdgcn_cosf(float rev) { return (float)std::cos((double)rev * 6.283185307179586476925); }
static inline int __builtin_amdgcn_bitop3_b32(int a, int b, int c, unsigned int table) {  // v_bitop3_b32: bit i of the result = table[(a_i << 2) | (b_i << 1) | c_i]
    unsigned int r = 0;
    for (int i = 0; i < 32; i++) {
        const unsigned int idx = ((((unsigned int)a >> i) & 1u) << 2) | ((((unsigned int)b >> i) & 1u) << 1) | (((unsigned int)c >> i) & 1u);
        r |= ((table >> idx) & 1u) << i;
    }
    return (int)r;
}
static inline float __builtin_amdgcn_fmed3f(float a, float b, float c) { return std::fmax(std::fmin(a, b), std::fmin(std::fmax(a, b), c)); }
typedef __fp16 hostsim_half2 __attribute__((ext_vector_type(2)));
static inline unsigned short hostsim_f2h_rtz(float f) {  // v_cvt_pkrtz_f16_f32: truncate, finite overflow saturates at 65504
    uint32_t u;
    std::memcpy(&u, &f, 4);
    const uint32_t sign = (u >> 16) & 0x8000u, mag = u & 0x7fffffffu;
    if (mag > 0x7f800000u) return (unsigned short)(sign | 0x7e00u);           // NaN
    if (mag == 0x7f800000u) return (unsigned short)(sign | 0x7c00u);          // inf
    if (mag >= 0x477fe000u) return (unsigned short)(sign | 0x7bffu);          // >= 65504: saturate
    if (mag < 0x33800000u) return (unsigned short)sign;                       // < 2^-24: zero
    const int e = (int)(mag >> 23) - 127;
    const uint32_t man = (mag & 0x7fffffu) | 0x800000u;
    if (e < -14) return (unsigned short)(sign | (man >> (13 + (-14 - e))));   // subnormal half, truncated
    return (unsigned short)(sign | ((uint32_t)(e + 15) << 10) | ((man & 0x7fffffu) >> 13));
}
static inline hostsim_half2 __builtin_amdgcn_cvt_pkrtz(float a, float b) {
    const uint32_t bits = (uint32_t)hostsim_f2h_rtz(a) | ((uint32_t)hostsim_f2h_rtz(b) << 16);
    hostsim_half2 r;
    std::memcpy(&r, &bits, 4);
    return r;
}
// ballot: the kernels use it only to choose between two forms of the SAME computation for a whole wavefront (a rare guarded path vs the
// common unguarded one) — here every thread decides for itself, which exercises both forms against the oracle
static inline unsigned long long __builtin_amdgcn_ballot_w64(bool p) { return p ? 1ull : 0ull; }
static inline float __builtin_amdgcn_fractf(float x) { return x - std::floor(x); }  // v_fract_f32 (arguments >= 0 here)
static inline float hostsim_half_word(uint32_t w, int sel) {  // one half of a 32-bit word as fp32 (exact)
    const unsigned short h = (unsigned short)(sel ? (w >> 16) : (w & 0xffffu));
    __fp16 v;
    std::memcpy(&v, &h, 2);
    return (float)v;
}
// v_fma_mix_f32 as rfx_device.h uses it: fp32(b.half) - fp32(a.half), one rounding; fma(w, d, fp32(a.half)), one rounding
static inline float hostsim_half_diff(uint32_t b, uint32_t a, int sel) { return hostsim_half_word(b, sel) - hostsim_half_word(a, sel); }
static inline float hostsim_half_fma(float w, float d, uint32_t a, int sel) { return std::fma(w, d, hostsim_half_word(a, sel)); }
static inline float hostsim_vmin(float a, float b) { return std::fmin(a, b); }  // v_min_f32 / v_max_f32 in IEEE mode: the non-NaN operand
static inline float hostsim_vmax(float a, float b) { return std::fmax(a, b); }
static inline int __mul24(int a, int b) { return a * b; }
static inline float __fmaf_rn(float a, float b, float c) { return std::fma(a, b, c); }
static inline unsigned int __float_as_uint(float f) { unsigned int u; std::memcpy(&u, &f, 4); return u; }
static inline int __float_as_int(float f) { int u; std::memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(unsigned int u) { float f; std::memcpy(&f, &u, 4); return f; }
static inline float __int_as_float(int u) { float f; std::memcpy(&f, &u, 4); return f; }
static inline float __builtin_inff_hostsim() { return __builtin_inff(); }
using std::max;
using std::min;
static inline int min(int a, unsigned int b) { return a < (int)b ? a : (int)b; }
static inline unsigned int atomicAdd(unsigned int *p, unsigned int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline unsigned int atomicOr(unsigned int *p, unsigned int v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
static inline unsigned long long atomicAdd(unsigned long long *p, unsigned long long v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
// rfx_peer.hip's flag barrier: the ranks are contexts of ONE process here, each driven by its own host thread (launches run on the calling
// thread), so the barrier kernels of two threads meet through these atomics; the poll sleeps for real, its bound stays a bound
#define __HIP_MEMORY_SCOPE_SYSTEM 0
#define __hip_atomic_store(p, v, order, scope) __atomic_store_n(p, v, order)
#define __hip_atomic_load(p, order, scope) __atomic_load_n(p, order)
static inline void __builtin_amdgcn_s_sleep(int) { struct timespec ts = {0, 1000}; nanosleep(&ts, nullptr); }
static inline int atomicMin(int *p, int v) {  // (idempotent: safe under the replay of a block's passes)
    int old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (v < old && !__atomic_compare_exchange_n(p, &old, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return old;
}
static inline int atomicMax(int *p, int v) {
    int old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (v > old && !__atomic_compare_exchange_n(p, &old, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return old;
}

// ---------------------------------------------------------------- runtime API (host memory stands for device memory; streams are immediate)
typedef int hipError_t;
typedef struct hostsim_stream *hipStream_t;
typedef struct hostsim_event *hipEvent_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2 };
enum { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipHostMallocDefault = 0, hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
enum { hipErrorNotSupported = 801, hipDeviceMallocFinegrained = 1, hipIpcMemLazyEnablePeerAccess = 1 };
struct hipIpcMemHandle_t { char reserved[64]; };
// no second address space to map: the peer-load exchange (rfx_peer.hip) works between contexts of one process (recognised by the blob's
// process id: addresses used directly, the handle is never opened) and reports RFX_EUNSUPPORTED for a blob of another process
static inline hipError_t hipIpcGetMemHandle(hipIpcMemHandle_t *h, void *) { std::memset(h, 0, sizeof *h); return hipSuccess; }
static inline hipError_t hipIpcOpenMemHandle(void **, hipIpcMemHandle_t, unsigned int) { return hipErrorNotSupported; }
static inline hipError_t hipIpcCloseMemHandle(void *) { return hipSuccess; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline const char *hipGetErrorString(hipError_t) { return "hostsim"; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
enum { hipDeviceAttributeMultiprocessorCount = 63 };
static inline hipError_t hipDeviceGetAttribute(int *v, int, int) { *v = 2; return hipSuccess; }  // a small "chip": the persistent K1 grid is 8 workgroups
static inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int *n, const void *, int, size_t) { *n = 4; return hipSuccess; }  // (the persistent K1 grid: 4 workgroups per "CU")
static inline hipError_t hipGetDeviceCount(int *n) { *n = 16; return hipSuccess; }  // every "device" is this host: one rank per device index works
static inline hipError_t hipMalloc(void **p, size_t n) { *p = std::aligned_alloc(256, (n + 255) & ~(size_t)255); return *p ? hipSuccess : hipErrorOutOfMemory; }
static inline hipError_t hipFree(void *p) { std::free(p); return hipSuccess; }
static inline hipError_t hipExtMallocWithFlags(void **p, size_t n, unsigned int) { *p = std::aligned_alloc(256, (n + 255) & ~(size_t)255); return *p ? hipSuccess : hipErrorOutOfMemory; }
static inline hipError_t hipHostMalloc(void **p, size_t n, unsigned int = 0) { return hipMalloc(p, n); }
static inline hipError_t hipHostFree(void *p) { std::free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void *d, const void *s, size_t n, int) { std::memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, int, hipStream_t = nullptr) { std::memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemset(void *d, int v, size_t n) { std::memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t = nullptr) { std::memset(d, v, n); return hipSuccess; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned int) { *s = (hipStream_t)std::malloc(8); return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t s) { std::free(s); return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned int = 0) { return hipSuccess; }
// events carry the wall-clock time of their record (streams are immediate): rfx_time_begin / rfx_time_end report the simulator's own speed
static inline double hostsim_now_ms() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }
static inline hipError_t hipEventCreate(hipEvent_t *e) { *e = (hipEvent_t)std::calloc(1, sizeof(double)); return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned int) { return hipEventCreate(e); }
static inline hipError_t hipEventDestroy(hipEvent_t e) { std::free(e); return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t = nullptr) { *(double *)e = hostsim_now_ms(); return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(*(double *)b - *(double *)a); return hipSuccess; }
template <class K>
static inline hipError_t hipFuncSetAttribute(K, int, int) { return hipSuccess; }

