// Implicit-GEMM Conv3d forward / data-gradient on bf16 MFMA (gfx950).
//
// One kernel serves both directions.  GEMM view:  D[co][m] = sum_k W[co][k] * X[m][k]
//   m  : a position of the "M-space" grid (N, Ta, Ha, Wa)   (forward: output positions;
//        dgrad: input positions, one launch per stride-parity class)
//   k  : (tap, source channel); the source row for (m, tap) is a*stride + tap offset
//   co : produced channel (forward: Cout, dgrad: Cin)
// The MFMA "A" operand is the weight tile (rows = co) and the "B" operand the gathered
// activation tile (cols = m), so each lane ends up with 4 consecutive channels of one
// position: an 8-byte store into the channels-last (NTHWC) output.
//
// Data path per 64-wide K chunk: global -> registers (16 B per lane, zero-filled where the
// tap falls outside the source) -> XOR-swizzled LDS tile -> ds_read_b128 fragments ->
// v_mfma_f32_16x16x32_bf16.  The next chunk's global loads are issued before the current
// chunk's MFMAs and written to LDS after them.
//
// Epilogue: bf16 store (optionally accumulating into the existing output), and optional
// per-channel sum / sum-of-squares (the BatchNorm batch statistics) reduced over the tile
// from the fp32 accumulators and added to one of VC_STAT_REPLICAS copies with float atomics.
#include "common.h"
#include "kernels.h"

namespace {

constexpr int BK = 64;           // K elements per chunk
constexpr int ROWB = BK * 2;     // bytes per LDS tile row

template <int BCO, int BM>
struct IgemmCfg {
    static constexpr int WCO = BCO / 64;           // waves along co
    static constexpr int WM = BM / 64;             // waves along m
    static constexpr int NX = BM / 32;             // 16-B X loads per thread per chunk
    static constexpr int NW = BCO / 32;            // 16-B W loads per thread per chunk
    static constexpr int LDS_BYTES = (BCO + BM) * ROWB + BM * 12;
    static_assert(WCO * WM == 4, "4 waves per workgroup");
};

__device__ __forceinline__ int swz_off(int row, int slot) {
    return row * ROWB + ((slot ^ ((row >> 1) & 7)) << 4);
}

template <int BCO, int BM, int MODE>  // MODE 0: tap table with bounds masks; 1: stem (padded s2d source)
__global__ __launch_bounds__(256, 2) void igemm_kernel(const VcIgemmParams p) {
    using Cfg = IgemmCfg<BCO, BM>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sW = smem;
    char* sX = smem + BCO * ROWB;
    int* sRowBase = (int*)(sX + BM * ROWB);
    unsigned* sRowMask = (unsigned*)(sRowBase + BM);
    int* sOutOff = (int*)(sRowMask + BM);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int m0 = blockIdx.x * BM;
    const int co0 = blockIdx.y * BCO;

    // ---- per-row geometry, one row per thread, shared through LDS ----
    for (int r = tid; r < BM; r += 256) {
        const int m = m0 + r;
        int base = 0, ooff = -1;
        unsigned mask = 0;
        if (m < p.M) {
            unsigned t1 = vc_div((unsigned)m, p.dWa);
            int aw = m - (int)t1 * p.Wa;
            unsigned t2 = vc_div(t1, p.dHa);
            int ah = (int)t1 - (int)t2 * p.Ha;
            unsigned n = vc_div(t2, p.dTa);
            int at = (int)t2 - (int)n * p.Ta;
            const int s_t = at * p.st, s_h = ah * p.sh, s_w = aw * p.sw;
            base = ((((int)n * p.Ts + s_t) * p.Hs + s_h) * p.Ws + s_w) * p.Cs;
            if (MODE == 0) {
                for (int i = 0; i < p.ntaps; ++i) {
                    const int tt = s_t + p.taps[i].ot, hh = s_h + p.taps[i].oh, ww = s_w + p.taps[i].ow;
                    const bool ok = (unsigned)tt < (unsigned)p.Ts && (unsigned)hh < (unsigned)p.Hs &&
                                    (unsigned)ww < (unsigned)p.Ws;
                    mask |= ok ? (1u << i) : 0u;
                }
            } else {
                mask = 0xffffffffu;
            }
            ooff = p.out_base + (int)n * p.out_n + at * p.out_t + ah * p.out_h + aw * p.out_w;
        }
        sRowBase[r] = base;
        sRowMask[r] = mask;
        sOutOff[r] = ooff;
    }
    __syncthreads();

    // ---- loader state ----
    const int lrow = tid >> 3;  // 0..31
    const int lslot = tid & 7;  // 16-B slot inside the 128-B chunk row
    int rb[Cfg::NX];
    unsigned rm[Cfg::NX];
#pragma unroll
    for (int i = 0; i < Cfg::NX; ++i) {
        rb[i] = sRowBase[i * 32 + lrow] + (MODE == 0 ? lslot * 8 : 0);
        rm[i] = sRowMask[i * 32 + lrow];
    }
    int wb[Cfg::NW];
#pragma unroll
    for (int j = 0; j < Cfg::NW; ++j) wb[j] = (co0 + j * 32 + lrow) * p.Kw + lslot * 8;

    bf16x8 xr[Cfg::NX];
    bf16x8 wr[Cfg::NW];

    const int cpt = (MODE == 0) ? (p.Cs >> 6) : 1;               // chunks per tap
    const int nchunks = (MODE == 0) ? p.ntaps * cpt : (p.Kw >> 6);

    auto load_chunk = [&](int ti, int c0, int chunk) {
        if (MODE == 0) {
            const int soff = p.taps[ti].src_off + c0;
            const int woff = p.taps[ti].wk_off + c0;
#pragma unroll
            for (int i = 0; i < Cfg::NX; ++i) {
                bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
                if ((rm[i] >> ti) & 1u) v = *(const bf16x8*)(p.src + (rb[i] + soff));
                xr[i] = v;
            }
#pragma unroll
            for (int j = 0; j < Cfg::NW; ++j) wr[j] = *(const bf16x8*)(p.wgt + (wb[j] + woff));
        } else {
            // stem: K = 28 macro-taps (kt, a_h) x 48 contiguous elements (a_w, b_h, b_w, c)
            const int g = chunk * 8 + lslot;  // 8-element K group
            const int j = g / 6;
            const int within = (g - j * 6) * 8;
            const int kt = j >> 2, ahh = j & 3;
            const int soff = ((kt * p.Hs + ahh) * p.Ws) * p.Cs + within;
#pragma unroll
            for (int i = 0; i < Cfg::NX; ++i) {
                bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
                if (rm[i]) {
                    const bf16x4* q = (const bf16x4*)(p.src + (rb[i] + soff));  // 8-B aligned only
                    bf16x4 lo = q[0], hi = q[1];
                    v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                }
                xr[i] = v;
            }
#pragma unroll
            for (int jj = 0; jj < Cfg::NW; ++jj) wr[jj] = *(const bf16x8*)(p.wgt + (wb[jj] + chunk * 64));
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int i = 0; i < Cfg::NX; ++i) *(bf16x8*)(sX + swz_off(i * 32 + lrow, lslot)) = xr[i];
#pragma unroll
        for (int j = 0; j < Cfg::NW; ++j) *(bf16x8*)(sW + swz_off(j * 32 + lrow, lslot)) = wr[j];
    };

    // ---- MFMA state ----
    const int wco = wave % Cfg::WCO;
    const int wm = wave / Cfg::WCO;
    const int l15 = lane & 15;
    const int lg = lane >> 4;
    const int lswz = (lane >> 1) & 7;  // == ((row >> 1) & 7) for row = 16*x + l15
    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    int cbeg = 0, cend = nchunks;
    if (p.splits > 1) {
        cbeg = blockIdx.z * p.cps;
        cend = min(nchunks, cbeg + p.cps);
    }
    int ti = 0, c0 = 0;
    if (MODE == 0) {
        ti = cbeg / cpt;
        c0 = (cbeg - ti * cpt) * 64;
    }
    if (cbeg < cend) {
        load_chunk(ti, c0, cbeg);
        store_chunk();
    }
    __syncthreads();

    for (int c = cbeg; c < cend; ++c) {
        int nti = ti, nc0 = c0 + 64;
        if (MODE == 0 && nc0 >= p.Cs) {
            nc0 = 0;
            nti = ti + 1;
        }
        const bool more = (c + 1 < cend);
        if (more) load_chunk(nti, nc0, c + 1);

#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 fa[4], fb[4];
            const int slot = ((ks << 2) + lg) ^ lswz;
#pragma unroll
            for (int a = 0; a < 4; ++a)
                fa[a] = *(const bf16x8*)(sW + (wco * 64 + a * 16 + l15) * ROWB + (slot << 4));
#pragma unroll
            for (int b = 0; b < 4; ++b)
                fb[b] = *(const bf16x8*)(sX + (wm * 64 + b * 16 + l15) * ROWB + (slot << 4));
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[a], fb[b], acc[a][b], 0, 0, 0);
        }
        __syncthreads();
        if (more) {
            store_chunk();
            __syncthreads();
        }
        ti = nti;
        c0 = nc0;
    }

    // ---- epilogue ----
    // acc[a][b][j] = D[co = co0 + wco*64 + a*16 + lg*4 + j][m = m0 + wm*64 + b*16 + l15]
    int ooff[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) ooff[b] = sOutOff[wm * 64 + b * 16 + l15];
    const int cobase = co0 + wco * 64 + lg * 4;

    if (p.partial != nullptr) {
        float* slab = p.partial + (size_t)blockIdx.z * p.partial_stride;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            if (ooff[b] < 0) continue;
#pragma unroll
            for (int a = 0; a < 4; ++a) *(f32x4*)(slab + (size_t)ooff[b] + cobase + a * 16) = acc[a][b];
        }
        return;
    }

    if (p.stats != nullptr) {
        float* srep = p.stats + (size_t)(blockIdx.x & (VC_STAT_REPLICAS - 1)) * 2 * p.Cn;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            float s[4], q[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float ss = 0.f, qq = 0.f;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const float v = acc[a][b][j];
                    ss += v;
                    qq += v * v;
                }
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) {
                    ss += __shfl_xor(ss, o);
                    qq += __shfl_xor(qq, o);
                }
                s[j] = ss;
                q[j] = qq;
            }
            if (l15 == 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    atomicAdd(srep + cobase + a * 16 + j, s[j]);
                    atomicAdd(srep + p.Cn + cobase + a * 16 + j, q[j]);
                }
            }
        }
    }

#pragma unroll
    for (int b = 0; b < 4; ++b) {
        if (ooff[b] < 0) continue;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            bf16* dst = p.out + (size_t)ooff[b] + cobase + a * 16;
            f32x4 v = acc[a][b];
            if (p.accumulate) {
                const bf16x4 old = *(const bf16x4*)dst;
                v[0] += bf2f(old[0]);
                v[1] += bf2f(old[1]);
                v[2] += bf2f(old[2]);
                v[3] += bf2f(old[3]);
            }
            bf16x4 o = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
            *(bf16x4*)dst = o;
        }
    }
}

template <int BCO, int BM, int MODE>
int launch_igemm(const VcIgemmParams& p, hipStream_t stream) {
    using Cfg = IgemmCfg<BCO, BM>;
    dim3 grid((p.M + BM - 1) / BM, p.Cn / BCO, p.splits > 1 ? p.splits : 1);
    (void)hipGetLastError();  // drop stale errors left by other users of the runtime
    hipLaunchKernelGGL((igemm_kernel<BCO, BM, MODE>), grid, dim3(256), Cfg::LDS_BYTES, stream, p);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        vc_set_error("igemm launch failed: %s", hipGetErrorString(e));
        return -1;
    }
    return 0;
}

}  // namespace

int vc_igemm_splits(int M, int Cn, int nchunks) {
    const bool wide = (Cn % 128 == 0);
    const long wgs = (long)((M + (wide ? 127 : 255)) / (wide ? 128 : 256)) * (Cn / (wide ? 128 : 64));
    if (wgs >= 384 || nchunks < 8) return 1;
    long s = (768 + wgs - 1) / wgs;
    if (s > nchunks / 4) s = nchunks / 4;
    if (s > 32) s = 32;
    return s < 2 ? 1 : (int)s;
}

// Host dispatch.  mode 0 = tap table, 1 = stem.
int vc_launch_igemm(const VcIgemmParams& p, int mode, hipStream_t stream) {
    if (p.M <= 0) return 0;
    if (p.Cn % 64 != 0 || (mode == 0 && p.Cs % 64 != 0) || p.Kw % 64 != 0) {
        vc_set_error("igemm: channels must be multiples of 64 (Cn=%d Cs=%d Kw=%d)", p.Cn, p.Cs, p.Kw);
        return -2;
    }
    if (mode == 0 && (p.ntaps < 1 || p.ntaps > 32)) {
        vc_set_error("igemm: ntaps=%d out of range", p.ntaps);
        return -2;
    }
    if (mode == 1) return launch_igemm<64, 256, 1>(p, stream);
    if (p.Cn % 128 == 0 && p.Cn >= 128) return launch_igemm<128, 128, 0>(p, stream);
    return launch_igemm<64, 256, 0>(p, stream);
}
