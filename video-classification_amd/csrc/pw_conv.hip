// Pointwise (1x1x1, unit stride) Conv3d forward / data-gradient on bf16 MFMA (gfx950): D[m][co] = sum_k X[m][k] W[co][k].
//
// The same tile as the implicit-GEMM kernel (igemm_conv.hip: LDS-DMA operand tiles with a source-side XOR swizzle,
// v_mfma_f32_16x16x32_bf16, output staged through LDS into whole channels-last rows, BatchNorm sums in the epilogue),
// stripped of everything a dense GEMM does not need -- tap tables, bounds masks, stride-parity classes, split-K, the
// row table -- because the Bottleneck's pointwise convs are bandwidth-bound and what limits a bandwidth-bound kernel
// is the bytes it keeps in flight: the general kernel holds ~230 VGPRs (two workgroups per CU); this one is built for
// WPE waves per SIMD so that three or four workgroups fit, each with its tile's loads outstanding.
#include <stdlib.h>

#include "common.h"
#include "kernels.h"

namespace {

constexpr int BK = 64;
constexpr int ROWB = BK * 2;

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __attribute__((aligned(128))) unsigned int vc_pw_zero_page[32];

template <int BCO, int BM>
struct PwCfg {
    static constexpr int WCO = BCO / 64, WM = BM / 64;
    static constexpr int NX = BM / 32, NW = BCO / 32;
    static constexpr int BUF_BYTES = (BCO + BM) * ROWB;
    static constexpr int STAGE_BYTES = BM * (BCO * 2 + 16);
    static constexpr int LDS_1 = BUF_BYTES > STAGE_BYTES ? BUF_BYTES : STAGE_BYTES;      // one K chunk per tile
    static constexpr int LDS_2 = 2 * BUF_BYTES > STAGE_BYTES ? 2 * BUF_BYTES : STAGE_BYTES;  // double buffered
    static_assert(WCO * WM == 4, "4 waves per workgroup");
};

// RED: the instantiation that carries the folded BatchNorm backward reduction instead of the forward statistics
template <int BCO, int BM, int WPE, bool RED>
__global__ __launch_bounds__(256, WPE) void pw_kernel(const VcPwParams p) {
    using Cfg = PwCfg<BCO, BM>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // The Cn / BCO workgroups of a row tile read the same activation rows.  In the plain (x, y) order they sit gridDim.x
    // apart and land on different XCDs (gridDim.x is rarely a multiple of 8), so each XCD's L2 fetched the rows for
    // itself: up to Cn / BCO = 16 reads of the input of a P -> 4P layer.  Here XCD k takes the k-th contiguous share of
    // the (x, y fastest) order: the co blocks of a tile are neighbours in one L2.
    unsigned bx = blockIdx.x, by = blockIdx.y;
    if (p.xcd_map) {
        const unsigned total = gridDim.x * gridDim.y;
        const unsigned lin = bx + gridDim.x * by;
        const unsigned k = lin & 7u, base = total >> 3, rem = total & 7u;
        const unsigned logical = k * base + (k < rem ? k : rem) + (lin >> 3);
        bx = logical / gridDim.y;
        by = logical - bx * gridDim.y;
    }
    const int co0 = (int)by * BCO;
    const int lrow = tid >> 3;
    const int lslot = (tid & 7) ^ ((lrow >> 1) & 7);
    const int wco = wave % Cfg::WCO, wm = wave / Cfg::WCO;
    const int l15 = lane & 15, lg = lane >> 4;
    const int lswz = (lane >> 1) & 7;
    const int cobase = co0 + wco * 64 + lg * 4;
    const int nchunks = p.K >> 6;
    const int ntiles = (p.M + BM - 1) / BM;
    const gptr_t zero = (gptr_t)vc_pw_zero_page;
    // byte offsets from the (uniform) tensor bases, 32 bits each: scalar base + vector offset addressing keeps the
    // loader's state at one register per piece (a tensor stays below 2^31 elements, the C ABI splits the batch above)
    const unsigned woff = (unsigned)((co0 + lrow) * p.K + lslot * 8) * 2u;  // + j * 32 * K * 2 per piece
    const char* const wbase = (const char*)p.wgt;
    const char* const xbase = (const char*)p.src;

    f32x4 st_s[4], st_q[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) st_s[a] = st_q[a] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // RED: the copy-out's decomposition of a tile -- PPR 16-byte pieces per row, thread = (piece rc8, row lane rr0)
    constexpr int PPR = BCO / 8, RSTEP = 256 / PPR, NIT = BM / RSTEP;
    const int rc8_ = tid % PPR, rr0_ = tid / PPR;
    float ra0[8], ra1[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) ra0[k] = ra1[k] = 0.f;

    for (int tile = (int)bx; tile < ntiles; tile += gridDim.x) {
        const int m0 = tile * BM;
        unsigned woff_t = woff;  // (opaque per tile: hoisted out of the tile loop, the pieces' 64-bit addresses were spills)
        int lrow_t = lrow;
        asm volatile("" : "+v"(woff_t), "+v"(lrow_t));
        auto issue_chunk = [&](int c, int buf) {
            char* bW = smem + buf * Cfg::BUF_BYTES + wave * (8 * ROWB);
            char* bX = bW + BCO * ROWB;
#pragma unroll
            for (int j = 0; j < Cfg::NW; ++j)
                __builtin_amdgcn_global_load_lds((gptr_t)(wbase + (woff_t + (unsigned)(j * 32 * p.K + c * 64) * 2u)),
                                                 (lptr_t)(bW + j * 32 * ROWB), 16, 0, 0);
#pragma unroll
            for (int i = 0; i < Cfg::NX; ++i) {
                const int m = m0 + i * 32 + lrow_t;
                const unsigned off = (unsigned)(m * p.K + lslot * 8 + c * 64) * 2u;
                const gptr_t src = m < p.M ? (gptr_t)(xbase + off) : zero;
                __builtin_amdgcn_global_load_lds(src, (lptr_t)(bX + i * 32 * ROWB), 16, 0, 0);
            }
        };
        f32x4 acc[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
        issue_chunk(0, 0);
        int buf = 0;
        for (int c = 0; c < nchunks; ++c) {
            __syncthreads();  // chunk c has landed; the other buffer was read one iteration ago
            if (c + 1 < nchunks) issue_chunk(c + 1, buf ^ 1);
            const char* sW = smem + buf * Cfg::BUF_BYTES;
            const char* sX = sW + BCO * ROWB;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 fb[4];
                const int slot = ((ks << 2) + lg) ^ lswz;
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    fb[b] = *(const bf16x8*)(sX + (wm * 64 + b * 16 + l15) * ROWB + (slot << 4));
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const bf16x8 fa = *(const bf16x8*)(sW + (wco * 64 + a * 16 + l15) * ROWB + (slot << 4));
#pragma unroll
                    for (int b = 0; b < 4; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb[b], acc[a][b], 0, 0, 0);
                }
            }
            buf ^= 1;
        }

        // ---- epilogue: acc[a][b][j] = D[co = cobase + a*16 + j][m = m0 + wm*64 + b*16 + l15]
        if (!RED && p.ep_params != nullptr) {  // inference: BatchNorm scale / shift (+ ReLU) on the fp32 accumulators
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const f32x4 s4 = *(const f32x4*)(p.ep_params + 2 * p.Cn + cobase + a * 16);
                const f32x4 h4 = *(const f32x4*)(p.ep_params + 3 * p.Cn + cobase + a * 16);
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    f32x4 v = acc[a][b] * s4 + h4;
                    if (p.ep_relu && p.ep_res == nullptr) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.f ? v[j] : 0.f;
                    }
                    acc[a][b] = v;
                }
            }
        }
        if (!RED && p.stats != nullptr) {
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const f32x4 sh = p.stat_shift != nullptr ? *(const f32x4*)(p.stat_shift + cobase + a * 16)
                                                         : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    if (m0 + BM > p.M && m0 + wm * 64 + b * 16 + l15 >= p.M) continue;  // zero rows beyond M: no -shift
                    const f32x4 v = acc[a][b] - sh;
                    st_s[a] += v;
                    st_q[a] += v * v;
                }
            }
        }
        constexpr int OPITCH = BCO * 2 + 16;
        bf16x8 ry[NIT];
        unsigned rb[NIT];
        // (opaque copies: everything the folded reduction addresses is recomputed per tile -- hoisted out of the tile
        // loop as invariants, its pointers were 15-20 registers the K loop had to carry, i.e. spills)
        int rc8 = rc8_, rr0 = rr0_;
        int l15e = l15, lge = lg;
        asm volatile("" : "+v"(rc8), "+v"(rr0), "+v"(l15e), "+v"(lge));
        const unsigned e0 = (unsigned)((m0 + rr0) * p.Cn + co0 + rc8 * 8);
        __syncthreads();  // every wave has finished reading the operand tiles
        const bf16* addsrc = p.ep_res != nullptr ? p.ep_res : (p.accumulate ? p.out : nullptr);
        if (addsrc != nullptr) {
#pragma unroll
            for (int it = 0; it < NIT; ++it) {  // a thread's pieces in flight at once (RED: in two halves, for its registers)
                if (RED && it == NIT / 2) asm volatile("" ::: "memory");
                const int r = rr0 + it * RSTEP;
                if (m0 + r < p.M) {
                    const unsigned e = e0 + (unsigned)(it * RSTEP * p.Cn);
                    bf16x8 piece = *(const bf16x8*)((const char*)addsrc + e * 2u);
                    if (p.acc_bits != nullptr) {  // dx = relu_mask(dx) + ...: the eight channels of a piece share one byte of bits
                        const unsigned mb = p.acc_bits[e >> 3];
#pragma unroll
                        for (int k = 0; k < 8; ++k) piece[k] = ((mb >> k) & 1u) ? piece[k] : (bf16)0;
                    }
                    *(bf16x8*)(smem + r * OPITCH + rc8 * 16) = piece;
                }
            }
        }
        if (RED) {  // what the folded reduction reads beside the tile: issued here (after the old tile's pieces have gone
                    // through their registers), used after the tile is staged
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int row = m0 + rr0 + it * RSTEP;
                ry[it] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
                rb[it] = 0u;
                if (row < p.M) {  // (scalar base + 32-bit offset: out and red_y share the element offset e)
                    const unsigned e = e0 + (unsigned)(it * RSTEP * p.Cn);
                    ry[it] = *(const bf16x8*)((const char*)p.red_y + e * 2u);
                    rb[it] = p.red_bits != nullptr ? *(p.red_bits + (e >> 3)) : 0x100u;  // (0x100: a row that counts, no bits yet)
                }
            }
        }
        if (addsrc != nullptr) __syncthreads();
#pragma unroll
        for (int b = 0; b < 4; ++b) {
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                bf16x4* q = (bf16x4*)(smem + (wm * 64 + b * 16 + l15e) * OPITCH + (wco * 64 + a * 16 + lge * 4) * 2);
                f32x4 v = acc[a][b];
                if (addsrc != nullptr) {
                    const bf16x4 old = *q;
                    v[0] += bf2f(old[0]), v[1] += bf2f(old[1]), v[2] += bf2f(old[2]), v[3] += bf2f(old[3]);
                    if (p.ep_res != nullptr && p.ep_relu) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.f ? v[j] : 0.f;
                    }
                }
                bf16x4 o = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
                *q = o;
            }
        }
        __syncthreads();
        if (RED) {
            // the copy-out with the BatchNorm backward sums of the rows it writes: the ROUNDED values, as the separate
            // reduction (chan_reduce_kernel<1, 3>) would read them back, in its access pattern -- a thread owns eight
            // channels and every RSTEP-th row, y and the bits come in as whole 128-byte pieces (fetched above)
            float rmean[8], rinv[8];  // (per tile, from the L1: sixteen registers the K loop does not have to carry)
#pragma unroll
            for (int k4 = 0; k4 < 2; ++k4) {
                const f32x4 m4 = *(const f32x4*)(p.red_params + co0 + rc8 * 8 + k4 * 4);
                const f32x4 i4 = *(const f32x4*)(p.red_params + p.Cn + co0 + rc8 * 8 + k4 * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) rmean[k4 * 4 + j] = m4[j], rinv[k4 * 4 + j] = i4[j];
            }
            if (p.red_bits == nullptr) {  // mask mode 2: the ReLU mask is scale * y + shift > 0, the forward's expression
                float rscale[8], rshift[8];
#pragma unroll
                for (int k4 = 0; k4 < 2; ++k4) {
                    const f32x4 s4 = *(const f32x4*)(p.red_params + 2 * p.Cn + co0 + rc8 * 8 + k4 * 4);
                    const f32x4 h4 = *(const f32x4*)(p.red_params + 3 * p.Cn + co0 + rc8 * 8 + k4 * 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) rscale[k4 * 4 + j] = s4[j], rshift[k4 * 4 + j] = h4[j];
                }
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    unsigned mb = 0u;
#pragma unroll
                    for (int k = 0; k < 8; ++k) mb |= (bf2f(ry[it][k]) * rscale[k] + rshift[k]) > 0.f ? 1u << k : 0u;
                    rb[it] = rb[it] != 0u ? mb : 0u;
                }
            }
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int r = rr0 + it * RSTEP;
                const bf16x8 piece = *(const bf16x8*)(smem + r * OPITCH + rc8 * 16);
                if (m0 + r < p.M) *(bf16x8*)((char*)p.out + (e0 + (unsigned)(it * RSTEP * p.Cn)) * 2u) = piece;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float gk = ((rb[it] >> k) & 1u) ? bf2f(piece[k]) : 0.f;  // rows beyond M: no bits
                    ra0[k] += gk;
                    ra1[k] += gk * (bf2f(ry[it][k]) - rmean[k]) * rinv[k];
                }
            }
        } else {
            for (int i = tid; i < BM * PPR; i += 256) {
                const int r = i / PPR, c8 = i - r * PPR;
                if (m0 + r < p.M)
                    *(bf16x8*)(p.out + (size_t)(m0 + r) * p.Cn + co0 + c8 * 8) = *(const bf16x8*)(smem + r * OPITCH + c8 * 16);
            }
        }
        __syncthreads();  // the staged tile has been read out before the next tile's operands land on it
    }

    // (the thread id again, from the lane count: carried through the tile loop in a register, it was one more spill)
    const int tid_e = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) + wave * 64;
    float* const sums = RED ? p.red_sums : p.stats;
    if (RED) {
        float* sm = (float*)smem;  // [256][17]: chan_reduce_kernel's block reduction over the row lanes
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            sm[tid_e * 17 + k] = ra0[k];
            sm[tid_e * 17 + 8 + k] = ra1[k];
        }
        __syncthreads();
        if (tid_e < 16 * PPR) {
            const int gg = tid_e % PPR, k = tid_e / PPR;
            float v = 0.f;
#pragma unroll 8
            for (int rr = 0; rr < RSTEP; ++rr) v += sm[(rr * PPR + gg) * 17 + k];
            vc_stat_add(sums, (unsigned)bx, p.Cn, (k < 8 ? 0 : p.Cn) + co0 + gg * 8 + (k & 7), v);
        }
    } else if (sums != nullptr) {
        float* sred = (float*)smem;  // [WM][2][BCO]
#pragma unroll
        for (int a = 0; a < 4; ++a) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float ss = st_s[a][j], qq = st_q[a][j];
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) {
                    ss += __shfl_xor(ss, o);
                    qq += __shfl_xor(qq, o);
                }
                if (l15 == 0) {
                    sred[(wm * 2 + 0) * BCO + wco * 64 + a * 16 + lg * 4 + j] = ss;
                    sred[(wm * 2 + 1) * BCO + wco * 64 + a * 16 + lg * 4 + j] = qq;
                }
            }
        }
        __syncthreads();
        for (int i = tid_e; i < 2 * BCO; i += 256) {
            const int mom = i / BCO, c = i - mom * BCO;
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < Cfg::WM; ++w) v += sred[(w * 2 + mom) * BCO + c];
            vc_stat_add(sums, (unsigned)bx, p.Cn, mom * p.Cn + co0 + c, v);
        }
    }
    // ---- BatchNorm finalize, per co block: the sums of channels [co0, co0 + BCO) are complete when the last of the
    // gridDim.x workgroups that share blockIdx.y has added its share, so THAT workgroup finalizes them (ticket word
    // blockIdx.y; bn_fin.h explains the memory-ordering side).  The tail is one round of loads whatever Cn is -- the
    // whole-layer form (one workgroup, Cn / 256 dependent rounds) cost a 2048-channel layer 11 us -- and it needs few
    // registers: unrolled over channels it had set this kernel's allocation at 248 VGPRs, i.e. its occupancy.
    // (the backward sums of the folded reduction finalize the same way: coef, dgamma, dbeta instead of the parameters)
    unsigned* const ticket = RED ? p.red_fin.ticket : (p.stats != nullptr ? p.fin.ticket : nullptr);
    if (ticket == nullptr) return;
    // (the flag lives in the dynamic LDS block: a static __shared__ word on top of 2 x 40 KB of operand buffers would
    // put the 64 x 256 form past the 80 KB it asks for)
    volatile unsigned* s_last_p = (volatile unsigned*)(smem + 8192);
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): this wave's atomics have been performed
    __syncthreads();  // also: every thread is done with the reduction scratch at the head of smem
    if (tid_e == 0) {
        unsigned last = 0u;
        const unsigned t = __hip_atomic_fetch_add(ticket + by, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t == gridDim.x - 1u) {
            __hip_atomic_store(ticket + by, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last = 1u;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        *s_last_p = last;
    }
    __syncthreads();
    if (*s_last_p == 0u) return;
    float* sfin = (float*)smem;  // [2][BCO]: sum and sum of squares over the replicas, in replica order
    if (tid_e < 2 * BCO) {
        const int mom = tid_e / BCO, c = co0 + (tid_e - mom * BCO);
        float v = 0.f;
        if (vc_stat_is_det(sums)) {
            v = vc_stat_total_det(sums, p.Cn, mom * p.Cn + c, true);
        } else {
#pragma unroll
            for (int r = 0; r < VC_STAT_REPLICAS; ++r)
                v += vc_stat_load<true>(sums + (size_t)r * 2 * p.Cn + mom * p.Cn + c);
        }
        sfin[tid_e] = v;
    }
    __syncthreads();
    if (RED) {
        if (tid_e < BCO) {  // vc_bn_bwd_finalize_channel's arithmetic
            const VcBnBwdFin& f = p.red_fin;
            const int c = co0 + tid_e;
            const float s = sfin[tid_e], q = sfin[BCO + tid_e];
            f.coef[c] = s / f.count;
            f.coef[p.Cn + c] = q / f.count;
            if (f.dgamma != nullptr) f.dgamma[c] += q;
            if (f.dbeta != nullptr) f.dbeta[c] += s;
        }
        return;
    }
    if (tid_e < BCO) {
        const VcBnFin& f = p.fin;
        const int c = co0 + tid_e, C = p.Cn;
        const float m1 = sfin[tid_e] / f.count;
        const float mean = (f.shift != nullptr ? f.shift[c] : 0.f) + m1;
        float var = sfin[BCO + tid_e] / f.count - m1 * m1;
        var = var > 0.f ? var : 0.f;
        if (f.shift_out != nullptr) f.shift_out[c] = mean;
        const float invstd = rsqrtf(var + f.eps);
        const float scale = f.gamma[c] * invstd;
        f.params[c] = mean;
        f.params[C + c] = invstd;
        f.params[2 * C + c] = scale;
        f.params[3 * C + c] = f.beta[c] - mean * scale;
        if (f.running_mean != nullptr) {
            const float unbiased = f.count > 1.f ? var * f.count / (f.count - 1.f) : var;
            f.running_mean[c] = (1.f - f.momentum) * f.running_mean[c] + f.momentum * mean;
            f.running_var[c] = (1.f - f.momentum) * f.running_var[c] + f.momentum * unbiased;
        }
    }
}

template <int BCO, int BM, int WPE, bool RED>
int launch_pw(const VcPwParams& p, hipStream_t stream) {
    using Cfg = PwCfg<BCO, BM>;
    const bool one_chunk = p.K == 64;
    const int lds = one_chunk ? Cfg::LDS_1 : Cfg::LDS_2;
    dim3 grid((p.M + BM - 1) / BM, p.Cn / BCO, 1);
    // persistent over tiles: all workgroups resident, same tile count each (the BatchNorm sums stay in registers)
    long per_cu = 160 * 1024 / lds;
    if (per_cu > WPE) per_cu = WPE;
    const long resident = 256l * per_cu;
    const long cap_x = resident / grid.y > 0 ? resident / grid.y : 1;
    if ((long)grid.x > cap_x) {
        const long per = ((long)grid.x + cap_x - 1) / cap_x;
        grid.x = (unsigned)(((long)grid.x + per - 1) / per);
    }
    if (lds > 64 * 1024) {
        static VcPerDeviceOnce once;
        once.run([&] {
            (void)hipFuncSetAttribute((const void*)pw_kernel<BCO, BM, WPE, RED>, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_2);
        });
    }
    VcPwParams q = p;
    q.stats = vc_det_tag(q.stats), q.red_sums = vc_det_tag(q.red_sums);
    static const bool no_xcd_map = getenv("VC_NO_XCD_MAP") != nullptr || getenv("VC_PW_NO_XCD_MAP") != nullptr;
    q.xcd_map = (grid.y > 1 && !no_xcd_map) ? 1 : 0;
    (void)hipGetLastError();
    hipLaunchKernelGGL((pw_kernel<BCO, BM, WPE, RED>), grid, dim3(256), lds, stream, q);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        vc_set_error("pointwise conv launch failed: %s", hipGetErrorString(e));
        return -1;
    }
    return 0;
}

}  // namespace

int vc_launch_pw(const VcPwParams& p, hipStream_t stream) {
    if (p.M <= 0) return 0;
    if (p.Cn % 64 != 0 || p.K % 64 != 0) {
        vc_set_error("pointwise conv: channels must be multiples of 64 (Cn=%d K=%d)", p.Cn, p.K);
        return -2;
    }
    if (p.red_sums != nullptr && (p.stats != nullptr || p.ep_params != nullptr || p.red_y == nullptr || p.red_params == nullptr)) {
        vc_set_error("pointwise conv: the folded BatchNorm backward reduction needs y and params and excludes "
                     "forward statistics and the inference epilogue");
        return -2;
    }
    if (p.red_sums != nullptr)
        return p.Cn % 128 == 0 ? launch_pw<128, 128, VC_PW_WPE, true>(p, stream) : launch_pw<64, 256, VC_PW_WPE, true>(p, stream);
    if (p.Cn % 128 == 0) return launch_pw<128, 128, VC_PW_WPE, false>(p, stream);
    return launch_pw<64, 256, VC_PW_WPE, false>(p, stream);
}
