// BatchNorm finalize folded into the launch that produces the per-channel sums.
//
// The sums of a BatchNorm layer ([VC_STAT_REPLICAS][2][C], float atomics) are complete when the last workgroup of
// the producing launch has added its share.  Every workgroup takes a ticket after its own atomics; the one that
// draws the last ticket reads the replicas (device-scope loads: the atomics of workgroups on other XCDs were
// performed below this XCD's L2) and does what the separate one-block finalize launch did, with the same
// arithmetic in the same order.  The ticket word is reset by that workgroup, so a zero-initialised word can be
// reused launch after launch; one word per layer keeps launches on different streams apart.
#pragma once
#include "common.h"

#define VC_TICKET_WORDS 64  // uint32 words behind a ticket pointer (33 used)

struct VcBnFin {  // forward: sums (sum y, sum y^2) -> params [4][C] = mean, invstd, scale, shift (+ running statistics)
    unsigned* ticket;  // nullptr: no fused finalize
    const float* gamma;
    const float* beta;
    float* running_mean;  // nullable
    float* running_var;
    float* params;
    float count, momentum, eps;
    // Shifted sums: the producers accumulate sum (y - shift[c]) and sum (y - shift[c])^2, so that a channel whose mean is
    // far from zero compared with its spread does not lose its variance to cancellation in fp32
    // (var = E[(y-s)^2] - E[y-s]^2 is exact in the limit s = mean; s = 0 is the textbook one-pass form).  Any shift
    // within a few sigma of the mean will do: the engine passes the batch mean of the previous step.
    const float* shift;   // nullable: [C], the shift the sums were taken with (null = 0)
    float* shift_out;     // nullable: [C], receives this batch's mean (may alias shift: the next step's shift)
};
struct VcBnBwdFin {  // backward: sums (sum g, sum g*xhat) -> coef [2][C]; dgamma += sum g*xhat; dbeta += sum g
    unsigned* ticket;
    float* coef;
    float* dgamma;  // nullable
    float* dbeta;   // nullable
    float count;
};

template <bool AGENT>
__device__ __forceinline__ float vc_stat_load(const float* p) {
    return AGENT ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p;
}
// The float-replica form, branch-free: all 2 x VC_STAT_REPLICAS loads of a channel (and, in the batched tails below, of
// several channels) are in flight at once.  With the mode test inside, the compiler cannot hoist the loads across it
// and the last workgroup's tail becomes a chain of dependent round trips below the L2 (R3D-50's 2048-channel
// downsample conv: 42 -> 206 us) -- so the callers test the mode ONCE and take this path or the deterministic one.
template <bool AGENT>
__device__ __forceinline__ void vc_stat_sum_f(const float* stats, int C, int c, float& s, float& q) {
    s = 0.f, q = 0.f;
#pragma unroll
    for (int r = 0; r < VC_STAT_REPLICAS; ++r) {
        s += vc_stat_load<AGENT>(stats + (size_t)r * 2 * C + c);
        q += vc_stat_load<AGENT>(stats + (size_t)r * 2 * C + C + c);
    }
}
template <bool AGENT>
__device__ __forceinline__ void vc_stat_sum(const float* stats, int C, int c, float& s, float& q) {
    if (vc_stat_is_det(stats)) {
        s = vc_stat_total_det(stats, C, c, AGENT);
        q = vc_stat_total_det(stats, C, C + c, AGENT);
    } else {
        vc_stat_sum_f<AGENT>(stats, C, c, s, q);
    }
}

template <bool AGENT>
__device__ __forceinline__ void vc_bn_finalize_channel(const float* stats, int C, int c, float count, const float* gamma,
                                                       const float* beta, float* running_mean, float* running_var,
                                                       float momentum, float eps, float* params,
                                                       const float* shift = nullptr, float* shift_out = nullptr) {
    float s, q;
    vc_stat_sum<AGENT>(stats, C, c, s, q);
    const float m1 = s / count;  // mean of (y - shift)
    const float mean = (shift != nullptr ? shift[c] : 0.f) + m1;
    float var = q / count - m1 * m1;
    var = var > 0.f ? var : 0.f;
    if (shift_out != nullptr) shift_out[c] = mean;
    const float invstd = rsqrtf(var + eps);
    const float scale = gamma[c] * invstd;
    params[c] = mean;
    params[C + c] = invstd;
    params[2 * C + c] = scale;
    params[3 * C + c] = beta[c] - mean * scale;
    if (running_mean != nullptr) {
        const float unbiased = count > 1.f ? var * count / (count - 1.f) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
    }
}

template <bool AGENT>
__device__ __forceinline__ void vc_bn_bwd_finalize_channel(const float* sums, int C, int c, float count, float* coef,
                                                           float* dgamma, float* dbeta) {
    float s, q;
    vc_stat_sum<AGENT>(sums, C, c, s, q);
    coef[c] = s / count;
    coef[C + c] = q / count;
    if (dgamma != nullptr) dgamma[c] += q;
    if (dbeta != nullptr) dbeta[c] += s;
}

// Called by EVERY thread of EVERY workgroup of the launch, after the workgroup's last statistics atomic.
// True (in all threads) for the workgroup that arrived last.
// Producer side: NO release fence.  On gfx950 a device-scope release writes back the XCD's L2, which at this point
// holds the workgroup's freshly written output tile (measured: 5.7 -> 9.8 ms per R3D-18 step), and the bytes handed
// over here need none: they are written by device-scope read-modify-write atomics only, which are performed at the
// memory side (below every XCD's L2; MI355X_MICROARCH.md, rows 'atomics of each flavour' and 'Global float
// atomics').  What the hand-off does need on this side is ORDER: every wave waits for the acknowledgements of its
// own atomics (vmcnt(0)), the workgroup's barrier follows, and only then does one lane draw the ticket.
// Consumer side: the workgroup that draws the last ticket runs an agent-scope ACQUIRE (buffer_inv sc1: drops this
// CU's L1 lines; ~1.7 us, paid by one workgroup per launch) before any of its waves reads the sums -- a reader in
// the adds' own launch needs it whatever the load flavour (same table, row 'read in the same launch') -- and the
// reads are device-scope (sc1) loads on top, so they are served below this XCD's L2 as well.
// Two levels of tickets: a reduction kernel's 2048 workgroups finish within a few microseconds of each other, and
// 2048 returning atomics on ONE word serialise at the memory side (measured: +10 us on a 20 us kernel).  Workgroup b
// draws from word 1 + (b & 31); the one that completes its word draws from word 0; the one that completes word 0
// is the last workgroup of the launch.  `ticket` points at VC_TICKET_WORDS zero words, left zero.
__device__ __forceinline__ bool vc_last_workgroup(unsigned* ticket) {
    __shared__ unsigned s_last;
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned nwg = gridDim.x * gridDim.y * gridDim.z;
        const unsigned bid = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        const unsigned k = bid & 31u;
        const unsigned expect = (nwg - k + 31u) >> 5;  // workgroups with id = k (mod 32)
        unsigned last = 0u;
        const unsigned t = __hip_atomic_fetch_add(ticket + 1 + k, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t == expect - 1u) {
            __hip_atomic_store(ticket + 1 + k, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned groups = nwg < 32u ? nwg : 32u;
            const unsigned t2 = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t2 == groups - 1u) {
                __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                last = 1u;
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the invalidate has completed before the barrier releases the readers
            }
        }
        s_last = last;
    }
    __syncthreads();
    return s_last != 0u;
}

// The last workgroup's loops take VC_FIN_BATCH channels per thread at a time, every load of the batch issued before
// the first use: one channel at a time is a chain of dependent ~2 us round trips below the L2 (8 of them for 2048
// channels and 256 threads: +16 us on a 14 us layer4 pointwise conv), a batch is one.  A batch holds 16 loaded words per
// channel: the GEMM kernels (register budget set by their accumulators) take four channels, the bandwidth kernels,
// whose occupancy hangs on staying within 64 VGPRs, take VC_FIN_BATCH = 1 (they finalize at most a few hundred channels
// per thread block anyway).
template <int VC_FIN_BATCH = 1>
__device__ __forceinline__ void vc_bn_fin_last(const VcBnFin& f, const float* stats, int C) {
    if (f.ticket == nullptr || stats == nullptr) return;
    if (!vc_last_workgroup(f.ticket)) return;
    if (VC_FIN_BATCH == 1) {
        for (int c = threadIdx.x; c < C; c += blockDim.x)
            vc_bn_finalize_channel<true>(stats, C, c, f.count, f.gamma, f.beta, f.running_mean, f.running_var, f.momentum,
                                         f.eps, f.params, f.shift, f.shift_out);
        return;
    }
    if (vc_stat_is_det(stats)) {  // deterministic mode: the plain per-channel loop (no batching needed for a debug mode)
        for (int c = threadIdx.x; c < C; c += blockDim.x)
            vc_bn_finalize_channel<true>(stats, C, c, f.count, f.gamma, f.beta, f.running_mean, f.running_var, f.momentum,
                                         f.eps, f.params, f.shift, f.shift_out);
        return;
    }
    const int step = (int)blockDim.x;
#pragma clang loop unroll(disable)  // unrolled, this tail sets the whole kernel's register allocation (and occupancy)
    for (int base = (int)threadIdx.x; base < C; base += step * VC_FIN_BATCH) {
        float s[VC_FIN_BATCH], q[VC_FIN_BATCH], g[VC_FIN_BATCH], b[VC_FIN_BATCH], rm[VC_FIN_BATCH], rv[VC_FIN_BATCH],
            sh[VC_FIN_BATCH];
#pragma unroll
        for (int k = 0; k < VC_FIN_BATCH; ++k) {
            const int c = base + k * step;
            s[k] = q[k] = g[k] = b[k] = rm[k] = rv[k] = sh[k] = 0.f;
            if (c < C) {
                vc_stat_sum_f<true>(stats, C, c, s[k], q[k]);
                g[k] = f.gamma[c], b[k] = f.beta[c];
                if (f.running_mean != nullptr) rm[k] = f.running_mean[c], rv[k] = f.running_var[c];
                if (f.shift != nullptr) sh[k] = f.shift[c];
            }
        }
#pragma unroll
        for (int k = 0; k < VC_FIN_BATCH; ++k) {  // vc_bn_finalize_channel's arithmetic, on the values loaded above
            const int c = base + k * step;
            if (c >= C) continue;
            const float m1 = s[k] / f.count;
            const float mean = sh[k] + m1;
            float var = q[k] / f.count - m1 * m1;
            var = var > 0.f ? var : 0.f;
            if (f.shift_out != nullptr) f.shift_out[c] = mean;
            const float invstd = rsqrtf(var + f.eps);
            const float scale = g[k] * invstd;
            f.params[c] = mean;
            f.params[C + c] = invstd;
            f.params[2 * C + c] = scale;
            f.params[3 * C + c] = b[k] - mean * scale;
            if (f.running_mean != nullptr) {
                const float unbiased = f.count > 1.f ? var * f.count / (f.count - 1.f) : var;
                f.running_mean[c] = (1.f - f.momentum) * rm[k] + f.momentum * mean;
                f.running_var[c] = (1.f - f.momentum) * rv[k] + f.momentum * unbiased;
            }
        }
    }
}
template <int VC_FIN_BATCH = 1>
__device__ __forceinline__ void vc_bn_bwd_fin_last(const VcBnBwdFin& f, const float* sums, int C) {
    if (f.ticket == nullptr || sums == nullptr) return;
    if (!vc_last_workgroup(f.ticket)) return;
    if (VC_FIN_BATCH == 1) {
        for (int c = threadIdx.x; c < C; c += blockDim.x)
            vc_bn_bwd_finalize_channel<true>(sums, C, c, f.count, f.coef, f.dgamma, f.dbeta);
        return;
    }
    if (vc_stat_is_det(sums)) {
        for (int c = threadIdx.x; c < C; c += blockDim.x)
            vc_bn_bwd_finalize_channel<true>(sums, C, c, f.count, f.coef, f.dgamma, f.dbeta);
        return;
    }
    const int step = (int)blockDim.x;
#pragma clang loop unroll(disable)  // unrolled, this tail sets the whole kernel's register allocation (and occupancy)
    for (int base = (int)threadIdx.x; base < C; base += step * VC_FIN_BATCH) {
        float s[VC_FIN_BATCH], q[VC_FIN_BATCH], dg[VC_FIN_BATCH], db[VC_FIN_BATCH];
#pragma unroll
        for (int k = 0; k < VC_FIN_BATCH; ++k) {
            const int c = base + k * step;
            s[k] = q[k] = dg[k] = db[k] = 0.f;
            if (c < C) {
                vc_stat_sum_f<true>(sums, C, c, s[k], q[k]);
                if (f.dgamma != nullptr) dg[k] = f.dgamma[c];
                if (f.dbeta != nullptr) db[k] = f.dbeta[c];
            }
        }
#pragma unroll
        for (int k = 0; k < VC_FIN_BATCH; ++k) {
            const int c = base + k * step;
            if (c >= C) continue;
            f.coef[c] = s[k] / f.count;
            f.coef[C + c] = q[k] / f.count;
            if (f.dgamma != nullptr) f.dgamma[c] = dg[k] + q[k];
            if (f.dbeta != nullptr) f.dbeta[c] = db[k] + s[k];
        }
    }
}
