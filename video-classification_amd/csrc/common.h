// Shared device/host helpers for the vc3d gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;

#define VC_STAT_REPLICAS 32  // per-channel atomic targets are spread over this many copies

// ---- BatchNorm sums ("stats" / "sums" buffers: 256 * C bytes, zeroed by the caller): two accumulation modes.
// Default: float atomics into VC_STAT_REPLICAS copies [32][2][C]; the last bits of a sum then depend on the arrival
// order of the adds (as with cuDNN / MIOpen).
// Deterministic (vc_set_deterministic(1) or VC_DETERMINISTIC=1; bit 0 of the pointer a launcher hands to its kernel is
// the mode flag -- the buffers are 256-byte aligned): every partial sum is added as TWO 64-bit integers, hi = floor(16 v)
// and lo = rint((v - hi / 16) * 2^44), by integer atomics.  Integer addition is associative, so the totals do not
// depend on the order of the adds: results are bit-identical from run to run.  Layout in the same 256 * C bytes:
// int64 [3 limbs: hi, lo, poison][VC_DET_REPLICAS][2][C].  Exact for |v| in [2^-21, 2^36) (a float's 24 bits above the
// 2^-44 quantum); smaller values keep the bits above 2^-44; non-finite or larger values raise the poison count and the
// total reads back as NaN.
#define VC_DET_REPLICAS 4
__device__ __forceinline__ bool vc_stat_is_det(const float* s) { return ((uintptr_t)s & (uintptr_t)1) != 0; }
template <typename T>
__device__ __forceinline__ T* vc_stat_untag(T* s) { return (T*)((uintptr_t)s & ~(uintptr_t)1); }
// (the deterministic paths are out-of-line functions with rolled loops: inlined, their 64-bit arithmetic set the register
// allocation -- and with it the occupancy -- of the bandwidth kernels that call them: chan_reduce 44 -> 86 VGPRs)
__device__ __attribute__((noinline)) void vc_stat_add_det(float* s, unsigned bid, int C, int idx, float v) {
    const size_t limb = (size_t)VC_DET_REPLICAS * 2 * C;
    unsigned long long* w = (unsigned long long*)vc_stat_untag(s) + (size_t)(bid & (VC_DET_REPLICAS - 1)) * 2 * C + idx;
    if (!(fabsf(v) < 0x1p36f)) {  // also NaN
        atomicAdd(w + 2 * limb, 1ull);
        return;
    }
    const double d = (double)v;
    const double h = floor(d * 16.0);
    const long long hi = (long long)h;
    const long long lo = (long long)rint((d - h * 0.0625) * 0x1p44);
    atomicAdd(w, (unsigned long long)hi);
    atomicAdd(w + limb, (unsigned long long)lo);
}
// add v to element idx (= moment * C + channel) of the sums; `bid` (a workgroup or row-phase id) picks the replica
__device__ __forceinline__ void vc_stat_add(float* s, unsigned bid, int C, int idx, float v) {
    if (!vc_stat_is_det(s)) {
        atomicAdd(s + (size_t)(bid & (VC_STAT_REPLICAS - 1)) * 2 * C + idx, v);
        return;
    }
    vc_stat_add_det(s, bid, C, idx, v);
}
__device__ __attribute__((noinline)) float vc_stat_total_det(const float* s, int C, int idx, bool agent) {
    const size_t limb = (size_t)VC_DET_REPLICAS * 2 * C;
    const long long* w = (const long long*)vc_stat_untag(s) + idx;
    long long hi = 0, lo = 0, bad = 0;
#pragma unroll 1
    for (int r = 0; r < VC_DET_REPLICAS; ++r) {
        const long long* q = w + (size_t)r * 2 * C;
        if (agent) {
            hi += __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            lo += __hip_atomic_load(q + limb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            bad += __hip_atomic_load(q + 2 * limb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            hi += q[0], lo += q[limb], bad += q[2 * limb];
        }
    }
    if (bad != 0) return __builtin_nanf("");
    return (float)((double)hi * 0.0625 + (double)lo * 0x1p-44);
}
// the total of element idx over the replicas (AGENT: device-scope loads, for a reader in the adds' own launch)
template <bool AGENT>
__device__ __forceinline__ float vc_stat_total(const float* s, int C, int idx) {
    if (vc_stat_is_det(s)) return vc_stat_total_det(s, C, idx, AGENT);
    float t = 0.f;
#pragma unroll  // all loads in flight at once, summed in replica order
    for (int r = 0; r < VC_STAT_REPLICAS; ++r) {
        const float* q = s + (size_t)r * 2 * C + idx;
        t += AGENT ? __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *q;
    }
    return t;
}

// One tap of an implicit-GEMM gather: source coordinate = a*stride + (ot,oh,ow).
struct VcTap {
    int src_off;  // ((ot*Hs + oh)*Ws + ow)*Cs, element offset added to the row base
    int wk_off;   // first K column of this tap in the weight matrix
    short ot, oh, ow, pad_;
};

// Exact n/d for 0 <= n < 2^31 (mulhi + shift); d == 1 handled by mul == 0.
struct VcFastDiv {
    unsigned mul, shr, d, pad_;
};

__device__ __forceinline__ unsigned vc_div(unsigned n, const VcFastDiv& f) {
    return f.mul ? (__umulhi(n, f.mul) >> f.shr) : n;
}

#ifdef __cplusplus
static inline VcFastDiv vc_make_fastdiv(unsigned d) {
    VcFastDiv f;
    f.d = d;
    f.pad_ = 0;
    if (d <= 1) {
        f.mul = 0;
        f.shr = 0;
        return f;
    }
    unsigned lg = 0;
    while ((1ull << lg) < d) ++lg;  // ceil(log2 d)
    unsigned p = 31 + lg;
    unsigned long long m = ((1ull << p) + d - 1) / d;
    f.mul = (unsigned)m;
    f.shr = p - 32;
    return f;
}
#endif

__device__ __forceinline__ float bf2f(bf16 v) { return (float)v; }
__device__ __forceinline__ bf16 f2bf(float v) { return (bf16)v; }

// thread-local error text for the C ABI
void vc_set_error(const char* fmt, ...);
// deterministic BatchNorm sums (see vc_stat_add): process-wide switch, read by every launcher that hands a sums buffer on
bool vc_det_mode();
template <typename T>
static inline T* vc_det_tag(T* p) { return (p != nullptr && vc_det_mode()) ? (T*)((uintptr_t)p | (uintptr_t)1) : p; }

#if defined(__cplusplus)
#include <mutex>
// One-time set-up per device (kernel attributes such as the 160 KiB dynamic-LDS limit), safe when several host threads
// launch on distinct streams at once: a plain `static bool done[dev]` is a data race there.
struct VcPerDeviceOnce {
    std::once_flag flag[64];
    template <typename F>
    void run(F&& fn) {
        int dev = -1;
        (void)hipGetDevice(&dev);
        if (dev >= 0 && dev < 64) std::call_once(flag[dev], fn);
        else fn();
    }
};
#endif
