// Conv3d weight gradient on bf16 MFMA (gfx950).
//
// GEMM view:  dW[co][(tap, ci)] = sum_m dY[m][co] * X[src(m, tap)][ci]
// The reduction index is the output position m, which is the slow (row) index of both
// operands in channels-last memory.  Both operand tiles are therefore kept row-major in
// LDS exactly as they arrive from global memory ([m][channel]) and the MFMA fragments are
// fetched with ds_read_b64_tr_b16, the gfx950 transposing LDS read: a 16-lane group reads
// a 4(m) x 16(channel) block and every lane receives 4 consecutive m of one channel.
// Two such reads form the 8-deep K fragment of v_mfma_f32_16x16x32_bf16.  The m -> k slot
// assignment is the same permutation for both operands, so the sum is unchanged.
//
// Work split: blockIdx.x = group of NB 64-wide column blocks (each block = one tap x 64 ci),
// blockIdx.y = BCO-wide co block, blockIdx.z = slice of m (split-K).  Partial sums are added
// to the fp32 gradient with float atomics (the gradient buffer is zeroed by the caller).
// Global -> registers -> LDS, one LDS buffer, STAGES register sets: the next STAGES - 1 chunks' loads are in flight
// while a chunk is multiplied.
#include <stdlib.h>

#include "common.h"
#include "kernels.h"

namespace {

constexpr int RSX = 160;  // bytes per LDS row of a 64-channel tile: 128 data + 32 pad

template <int BCO, int NB>
struct WgCfg {
    static constexpr int RSY = BCO * 2 + 32;
    static constexpr int NY = BCO / 32;  // 16-B dY loads per thread per chunk
    static constexpr int NXL = NB * 2;   // 16-B X loads per thread per chunk
    static constexpr int LDS_BYTES = 64 * RSY + NB * 64 * RSX;
    static_assert((BCO / 64) * NB == 4, "4 waves per workgroup");
};

template <int BCO, int NB, int MODE, int STAGES>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(const VcWgradParams p) {
    using Cfg = WgCfg<BCO, NB>;
    typedef s16x4 __attribute__((address_space(3))) * lds_s16x4_p;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sY = smem;
    char* sX = smem + 64 * Cfg::RSY;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    // Workgroups are dealt round-robin to the 8 XCDs, each with its own L2, and the workgroups of one slice of m -- the
    // co blocks and the tap / ci groups -- read the same dY and x rows: XCD k takes the k-th contiguous share of the
    // (z, y, x) order, so that a slice's workgroups meet in one L2 (as in wgrad9_conv.hip).  Without it the pointwise
    // layers of R3D-50 drew 1.7x their algorithmic bytes from below the L2.
    unsigned bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (p.xcd_map) {
        const unsigned total = gridDim.x * gridDim.y * gridDim.z;
        const unsigned lin = bx + gridDim.x * (by + gridDim.y * bz);
        const unsigned k = lin & 7u, base = total >> 3, rem = total & 7u;
        const unsigned logical = k * base + (k < rem ? k : rem) + (lin >> 3);
        bx = logical % gridDim.x;
        const unsigned t = logical / gridDim.x;
        by = t % gridDim.y;
        bz = t / gridDim.y;
    }
    const int co0 = by * BCO;
    const int mb = bz * p.rows_per_split;
    const int mend = min(p.M, mb + p.rows_per_split);
    const int nchunks = (mend - mb + 63) >> 6;

    const int cpt = (MODE == 0) ? (p.Cs >> 6) : 1;
    const int nblk = (MODE == 0) ? p.ntaps * cpt : (p.Kw >> 6);

    // per-block (uniform) gather description
    int b_soff[NB], b_ot[NB], b_oh[NB], b_ow[NB], b_col[NB];
    bool b_ok[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int cb = (int)bx * NB + b;
        b_ok[b] = cb < nblk;
        if (MODE == 0) {
            const int tap = b_ok[b] ? cb / cpt : 0;
            const int ci0 = (cb - tap * cpt) * 64;
            b_soff[b] = p.taps[tap].src_off + ci0;
            b_ot[b] = p.taps[tap].ot;
            b_oh[b] = p.taps[tap].oh;
            b_ow[b] = p.taps[tap].ow;
            b_col[b] = p.taps[tap].wk_off + ci0;
        } else {
            b_soff[b] = 0;
            b_ot[b] = b_oh[b] = b_ow[b] = 0;
            b_col[b] = cb * 64;
        }
    }

    const int lrow = tid >> 2;  // 0..63: the m row this thread stages
    const int lsub = tid & 3;

    // stem: per-piece source offsets (depend on the K group only)
    int st_soff[NB][2];
    if (MODE == 1) {
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int g = ((int)bx * NB + b) * 8 + lsub + 4 * h;
                const int j = g / 6;
                const int within = (g - j * 6) * 8;
                st_soff[b][h] = (((j >> 2) * p.Hs + (j & 3)) * p.Ws) * p.Cs + within;
            }
    }

    // STAGES register sets of one chunk each: the loads of the next STAGES - 1 chunks are in flight while one is
    // multiplied (a workgroup with ONE chunk in flight waits out most of every HBM round trip: 24-40 KB per ~2 us)
    bf16x8 yrs[STAGES][Cfg::NY];
    bf16x8 xrs[STAGES][NB][2];

    auto load_chunk = [&](int c, bf16x8 (&yr)[Cfg::NY], bf16x8 (&xr)[NB][2]) {
        const int m = mb + c * 64 + lrow;
        const bool rowok = m < mend;
        int base = 0, s_t = 0, s_h = 0, s_w = 0;
        if (rowok) {
            unsigned t1 = vc_div((unsigned)m, p.dWa);
            int aw = m - (int)t1 * p.Wa;
            unsigned t2 = vc_div(t1, p.dHa);
            int ah = (int)t1 - (int)t2 * p.Ha;
            unsigned n = vc_div(t2, p.dTa);
            int at = (int)t2 - (int)n * p.Ta;
            s_t = at * p.st;
            s_h = ah * p.sh;
            s_w = aw * p.sw;
            base = ((((int)n * p.Ts + s_t) * p.Hs + s_h) * p.Ws + s_w) * p.Cs;
        }
#pragma unroll
        for (int h = 0; h < Cfg::NY; ++h) {
            bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (rowok) v = *(const bf16x8*)(p.dy + ((size_t)m * p.Co + co0 + (lsub + 4 * h) * 8));
            yr[h] = v;
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            bool ok = rowok && b_ok[b];
            if (MODE == 0) {
                const int tt = s_t + b_ot[b], hh = s_h + b_oh[b], ww = s_w + b_ow[b];
                ok = ok && (unsigned)tt < (unsigned)p.Ts && (unsigned)hh < (unsigned)p.Hs &&
                     (unsigned)ww < (unsigned)p.Ws;
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
                if (ok) {
                    if (MODE == 0) {
                        v = *(const bf16x8*)(p.x + (base + b_soff[b] + (lsub + 4 * h) * 8));
                    } else {
                        const bf16x4* q = (const bf16x4*)(p.x + (base + st_soff[b][h]));
                        bf16x4 lo = q[0], hi = q[1];
                        v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                    }
                }
                xr[b][h] = v;
            }
        }
    };
    auto store_chunk = [&](const bf16x8 (&yr)[Cfg::NY], const bf16x8 (&xr)[NB][2]) {
#pragma unroll
        for (int h = 0; h < Cfg::NY; ++h) *(bf16x8*)(sY + lrow * Cfg::RSY + (lsub + 4 * h) * 16) = yr[h];
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int h = 0; h < 2; ++h) *(bf16x8*)(sX + (b * 64 + lrow) * RSX + (lsub + 4 * h) * 16) = xr[b][h];
    };

    // wave roles
    const int wco = wave % (BCO / 64);
    const int wb = wave / (BCO / 64);
    const int l15 = lane & 15;
    const int lg = lane >> 4;
    const int q = l15 >> 2, pp = l15 & 3;
    // address of this lane's row/column inside a 4x16 transposed-read block
    const char* aY = sY + (4 * lg + q) * Cfg::RSY + (wco * 64 + 4 * pp) * 2;
    const char* aX = sX + (wb * 64 + 4 * lg + q) * RSX + (4 * pp) * 2;

    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int s = 0; s < STAGES; ++s)
        if (s < nchunks) load_chunk(s, yrs[s], xrs[s]);

    for (int c0 = 0; c0 < nchunks; c0 += STAGES) {
#pragma unroll
      for (int s = 0; s < STAGES; ++s) {
        const int c = c0 + s;
        if (c >= nchunks) break;  // (uniform)
        store_chunk(yrs[s], xrs[s]);
        __syncthreads();
        if (c + STAGES < nchunks) load_chunk(c + STAGES, yrs[s], xrs[s]);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 fa[4], fb[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const char* r0 = aY + (ks * 32) * Cfg::RSY + a * 32;
                s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(r0));
                s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(r0 + 16 * Cfg::RSY));
                fa[a] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const char* r0 = aX + (ks * 32) * RSX + b * 32;
                s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(r0));
                s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(r0 + 16 * RSX));
                fb[b] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[a], fb[b], acc[a][b], 0, 0, 0);
        }
        __syncthreads();
      }
    }

    // ---- epilogue: acc[a][b][j] = dW[co0 + wco*64 + a*16 + lg*4 + j][col(wb) + b*16 + l15]
    bool okb = false;
    int colb = 0;
#pragma unroll
    for (int b = 0; b < NB; ++b)
        if (b == wb) {
            okb = b_ok[b];
            colb = b_col[b];
        }
    if (!okb) return;
    float* dst = p.dw;
    int how = 0;  // 0 atomics, 1 plain += (single split owns the element), 2 plain store into a slab
    if (p.partial != nullptr && p.splits > 1) {
        dst = p.partial + (size_t)bz * p.partial_stride;
        how = 2;
    } else if (p.splits <= 1) {
        how = 1;
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float* row = dst + (size_t)(co0 + wco * 64 + a * 16 + lg * 4 + j) * p.Kw + colb + l15;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                if (how == 0) atomicAdd(row + b * 16, acc[a][b][j]);
                else if (how == 1) row[b * 16] += acc[a][b][j];
                else row[b * 16] = acc[a][b][j];
            }
        }
}

template <int BCO, int NB, int MODE, int STAGES>
int launch_wgrad(const VcWgradParams& p, int splits, hipStream_t stream) {
    using Cfg = WgCfg<BCO, NB>;
    const int nblk = (MODE == 0) ? p.ntaps * (p.Cs / 64) : p.Kw / 64;
    dim3 grid((nblk + NB - 1) / NB, p.Co / BCO, splits);
    (void)hipGetLastError();  // drop stale errors left by other users of the runtime
    hipLaunchKernelGGL((wgrad_kernel<BCO, NB, MODE, STAGES>), grid, dim3(256), Cfg::LDS_BYTES, stream, p);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        vc_set_error("wgrad launch failed: %s", hipGetErrorString(e));
        return -1;
    }
    return 0;
}

}  // namespace

// one 64-wide column block (a pointwise conv from 64 channels) onto a multiple of 256 output channels: the 256 x 64 tile
static bool wgrad_one_block(int Co, int nblk) {
    static const bool off = getenv("VC_WGRAD_NO_256") != nullptr;
    return nblk == 1 && Co % 256 == 0 && !off;
}

int vc_wgrad_splits(int M, int Co, int nblk, int* rows_per_split) {
    const bool wide = (Co % 128 == 0);
    const bool one = wgrad_one_block(Co, nblk);
    const int nb = one ? 1 : wide ? 2 : 4;
    const int bco = one ? 256 : wide ? 128 : 64;
    const long tiles = (long)((nblk + nb - 1) / nb) * (Co / bco);
    // aim for ~192 workgroups, at least 4 chunks (256 rows) per split.  Every split costs a slab of the whole gradient
    // (written, then read by the reduce): in isolation 512 against 1024 measured 15-25 % faster on the 1x1x1 and the
    // strided 3x3x3 shapes of R3D-18/50; in the step these launches run on the side stream beside the main stream's
    // kernels, and 192 -- which leaves CUs to those and writes fewer slabs -- beat the 512 that fill every CU twice:
    // R3D-50 step 10.49 -> 10.37 ms, R3D-18 4.87 -> 4.86 (sweep 128 .. 768)
    static const long target = getenv("VC_WGRAD_TARGET") ? atol(getenv("VC_WGRAD_TARGET")) : 192;
    long splits = (target + tiles - 1) / tiles;
    const long maxsplits = (M + 255) / 256;
    if (splits > maxsplits) splits = maxsplits;
    if (splits < 1) splits = 1;
    if (splits > 1024) splits = 1024;
    long rps = (M + splits - 1) / splits;
    rps = (rps + 63) / 64 * 64;
    splits = (M + rps - 1) / rps;
    *rows_per_split = (int)rps;
    return (int)splits;
}

int vc_launch_wgrad(const VcWgradParams& p_in, int mode, hipStream_t stream) {
    VcWgradParams p = p_in;
    if (p.M <= 0) return 0;
    if (p.Co % 64 != 0 || (mode == 0 && p.Cs % 64 != 0) || p.Kw % 64 != 0) {
        vc_set_error("wgrad: channels must be multiples of 64 (Co=%d Cs=%d Kw=%d)", p.Co, p.Cs, p.Kw);
        return -2;
    }
    const bool wide = (p.Co % 128 == 0);
    const int nblk = (mode == 0) ? p.ntaps * (p.Cs / 64) : p.Kw / 64;
    int rps = 0;
    const int splits = vc_wgrad_splits(p.M, p.Co, nblk, &rps);
    p.rows_per_split = rps;
    p.splits = splits;
    static const bool no_xcd_map = getenv("VC_NO_XCD_MAP") != nullptr || getenv("VC_WGRAD_NO_XCD_MAP") != nullptr;
    p.xcd_map = no_xcd_map ? 0 : 1;
    static const int stages = getenv("VC_WGRAD_STAGES") ? atoi(getenv("VC_WGRAD_STAGES")) : 3;
    if (mode == 1) return launch_wgrad<64, 4, 1, 1>(p, splits, stream);
    if (mode == 0 && wgrad_one_block(p.Co, nblk)) {
        // one 64-wide column block (a pointwise conv from 64 channels): four waves side by side along co, none idle,
        // and x read once per 256 co instead of once per 128
        if (stages <= 1) return launch_wgrad<256, 1, 0, 1>(p, splits, stream);
        if (stages == 2) return launch_wgrad<256, 1, 0, 2>(p, splits, stream);
        return launch_wgrad<256, 1, 0, 3>(p, splits, stream);
    }
    if (wide) {
        if (stages <= 1) return launch_wgrad<128, 2, 0, 1>(p, splits, stream);
        if (stages == 2) return launch_wgrad<128, 2, 0, 2>(p, splits, stream);
        return launch_wgrad<128, 2, 0, 3>(p, splits, stream);
    }
    if (stages <= 1) return launch_wgrad<64, 4, 0, 1>(p, splits, stream);
    if (stages == 2) return launch_wgrad<64, 4, 0, 2>(p, splits, stream);
    return launch_wgrad<64, 4, 0, 3>(p, splits, stream);
}
