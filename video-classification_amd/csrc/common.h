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
