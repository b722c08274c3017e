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
#ifndef VC_IGEMM_PIPE
#define VC_IGEMM_PIPE 1  // two fragment sets per chunk with a pinned read / MFMA interleave (0: the unpinned loop)
#endif
#include "kernels.h"

namespace {

constexpr int BK = 64;           // K elements per chunk
constexpr int ROWB = BK * 2;     // bytes per LDS tile row

// masked (out-of-bounds) lanes of an LDS-DMA load fetch zeros from here
__device__ __attribute__((aligned(128))) unsigned int vc_zero_page[32];

template <int BCO, int BM>
struct IgemmCfg {
    static constexpr int WCO = BCO / 64;           // waves along co
    static constexpr int WM = BM / 64;             // waves along m
    static constexpr int NX = BM / 32;             // 1-KiB X pieces per wave per chunk
    static constexpr int NW = BCO / 32;            // 1-KiB W pieces per wave per chunk
    static constexpr int BUF_BYTES = (BCO + BM) * ROWB;
    static constexpr int LDS_BYTES = 2 * BUF_BYTES;  // double buffered
    static constexpr int STAGE_BYTES = BM * (BCO * 2 + 16) + BM * 4;  // epilogue: staged output tile + row offsets
    // a launch whose workgroups see a single K chunk (1x1x1 convs with 64 source channels) needs one buffer only:
    // four workgroups per CU instead of two, i.e. twice the bytes in flight for what is a bandwidth-bound GEMM
    static constexpr int LDS_BYTES_1 = BUF_BYTES > STAGE_BYTES ? BUF_BYTES : STAGE_BYTES;
    static_assert(WCO * WM == 4, "4 waves per workgroup");
    static_assert(BM * 12 <= BUF_BYTES, "row table must fit in one buffer");
};

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// Data path: global --(LDS-DMA, 16 B per lane, 1 KiB per wave instruction)--> LDS --> ds_read_b128
// fragments --> MFMA.  An LDS-DMA piece lands lane-linear (8 rows x 128 B), so the XOR swizzle that
// keeps the fragment reads conflict-free is applied to the SOURCE address: the lane that fills
// 16-B slot p of row r fetches logical slot p ^ ((r >> 1) & 7).
template <int BCO, int BM, int MODE>  // MODE 0: tap table with bounds masks; 1: stem (padded s2d source)
__global__ __launch_bounds__(256, 2) void igemm_kernel(const VcIgemmParams p) {
    using Cfg = IgemmCfg<BCO, BM>;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int co0 = blockIdx.y * BCO;
    // output geometry: the launch's own, or that of this block's class
    int M = p.M, Ta = p.Ta, Ha = p.Ha, Wa = p.Wa, out_base = p.out_base, tap0 = 0, ntaps = p.ntaps;
    int zsplit = blockIdx.z, cps = p.cps;
    VcFastDiv dWa = p.dWa, dHa = p.dHa, dTa = p.dTa;
    if (MODE == 0 && p.ncls > 1) {
        const int zs = p.splits > 1 ? p.splits : 1;
        const int c = blockIdx.z / zs;
        zsplit = blockIdx.z - c * zs;
        M = p.cls[c].M, Ta = p.cls[c].Ta, Ha = p.cls[c].Ha, Wa = p.cls[c].Wa;
        dWa = p.cls[c].dWa, dHa = p.cls[c].dHa, dTa = p.cls[c].dTa;
        out_base = p.cls[c].out_base, tap0 = p.cls[c].tap0, ntaps = p.cls[c].ntaps;
        cps = (ntaps * (p.Cs >> 6) + zs - 1) / zs;
    }
    const int ntiles = (M + BM - 1) / BM;  // (the grid of a class launch is sized for the largest class)

    // ---- tile-independent lane roles
    // loader: this lane fills LDS position (row = piece*32 + lrow, physical slot lane&7)
    const int lrow = tid >> 3;                        // 0..31 = wave*8 + (lane>>3)
    const int lslot = (tid & 7) ^ ((lrow >> 1) & 7);  // logical (source) slot fetched by this lane
    int wb[Cfg::NW];
#pragma unroll
    for (int j = 0; j < Cfg::NW; ++j) wb[j] = (co0 + j * 32 + lrow) * p.Kw + lslot * 8;
    // MFMA / epilogue
    const int wco = wave % Cfg::WCO;
    const int wm = wave / Cfg::WCO;
    const int l15 = lane & 15;
    const int lg = lane >> 4;
    const int lswz = (lane >> 1) & 7;  // == ((row >> 1) & 7) for row = 16*x + l15
    const int cobase = co0 + wco * 64 + lg * 4;
    const int cpt = (MODE == 0) ? (p.Cs >> 6) : 1;  // chunks per tap
    const int nchunks = (MODE == 0) ? ntaps * cpt : (p.Kw >> 6);
    const gptr_t zero = (gptr_t)vc_zero_page;
    int cbeg = 0, cend = nchunks;
    if (p.splits > 1) {
        cbeg = zsplit * cps;
        cend = min(nchunks, cbeg + cps);
    }

    int rb[Cfg::NX];
    unsigned rm[Cfg::NX];
    auto issue_chunk = [&](int ti, int c0, int chunk, int buf) {
        char* bW = smem + buf * Cfg::BUF_BYTES + wave * (8 * ROWB);
        char* bX = bW + BCO * ROWB;
        int soff, woff;
        if (MODE == 0) {
            soff = p.taps[tap0 + ti].src_off + c0;
            woff = p.taps[tap0 + ti].wk_off + c0;
        } else {
            // stem: K = 28 macro-taps (kt, a_h) x 48 contiguous elements (a_w, b_h, b_w, c)
            const int g = chunk * 8 + lslot;  // 8-element K group fetched by this lane
            const int j = g / 6;
            soff = (((j >> 2) * p.Hs + (j & 3)) * p.Ws) * p.Cs + (g - j * 6) * 8;
            woff = chunk * 64;
        }
#pragma unroll
        for (int j = 0; j < Cfg::NW; ++j)
            __builtin_amdgcn_global_load_lds((gptr_t)(p.wgt + (wb[j] + woff)), (lptr_t)(bW + j * 32 * ROWB), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < Cfg::NX; ++i) {
            const bool ok = (MODE == 0) ? (((rm[i] >> ti) & 1u) != 0) : (rm[i] != 0);
            const gptr_t src = ok ? (gptr_t)(p.src + (rb[i] + soff)) : zero;
            __builtin_amdgcn_global_load_lds(src, (lptr_t)(bX + i * 32 * ROWB), 16, 0, 0);
        }
    };

    // BatchNorm statistics: per-lane partial sums over this workgroup's tiles, reduced across lanes and added to
    // the replicated accumulators once, after the tile loop (device-scope float atomics are the expensive part:
    // one set per tile held a 64 -> 256 pointwise conv to half its bandwidth)
    f32x4 st_s[4], st_q[4], st_h[4];  // st_h: the shift of the sums (bn_fin.h), 0 without one
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        st_s[a] = st_q[a] = (f32x4){0.f, 0.f, 0.f, 0.f};
        st_h[a] = (p.stats != nullptr && p.stat_shift != nullptr) ? *(const f32x4*)(p.stat_shift + cobase + a * 16)
                                                                  : (f32x4){0.f, 0.f, 0.f, 0.f};
    }

    // A workgroup walks tiles blockIdx.x, blockIdx.x + gridDim.x, ... (the host caps gridDim.x for single-geometry
    // launches so that all workgroups are resident and own the same number of tiles).
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int m0 = tile * BM;
        // ---- per-row geometry, one row per thread, staged in buffer 0 (consumed before the first chunk is issued)
        int* sRowBase = (int*)smem;
        unsigned* sRowMask = (unsigned*)(sRowBase + BM);
        int* sOutOff = (int*)(sRowMask + BM);
        int my_ooff = -1;  // out offset of row tid (BM <= 256), put back into LDS for the staged store
        int ooff[4];
        if (MODE == 0 && p.pointwise) {
            // 1x1x1, unit stride, dense output: no gather, no bounds -- the row table (and its two barriers) is arithmetic
            if (tid < BM) my_ooff = m0 + tid < M ? out_base + (m0 + tid) * p.Cn : -1;
#pragma unroll
            for (int i = 0; i < Cfg::NX; ++i) {
                const int m = m0 + i * 32 + lrow;
                rb[i] = (m < M ? m : 0) * p.Cs + lslot * 8;
                rm[i] = m < M ? 1u : 0u;
            }
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int m = m0 + wm * 64 + b * 16 + l15;
                ooff[b] = m < M ? out_base + m * p.Cn : -1;
            }
        } else {
        for (int r = tid; r < BM; r += 256) {
            const int m = m0 + r;
            int base = 0, ooff = -1;
            unsigned mask = 0;
            if (m < M) {
                unsigned t1 = vc_div((unsigned)m, dWa);
                int aw = m - (int)t1 * Wa;
                unsigned t2 = vc_div(t1, dHa);
                int ah = (int)t1 - (int)t2 * Ha;
                unsigned n = vc_div(t2, dTa);
                int at = (int)t2 - (int)n * Ta;
                const int s_t = at * p.st, s_h = ah * p.sh, s_w = aw * p.sw;
                base = ((((int)n * p.Ts + s_t) * p.Hs + s_h) * p.Ws + s_w) * p.Cs;
                if (MODE == 0) {
                    for (int i = 0; i < ntaps; ++i) {
                        const int tt = s_t + p.taps[tap0 + i].ot, hh = s_h + p.taps[tap0 + i].oh,
                                  ww = s_w + p.taps[tap0 + i].ow;
                        const bool ok = (unsigned)tt < (unsigned)p.Ts && (unsigned)hh < (unsigned)p.Hs &&
                                        (unsigned)ww < (unsigned)p.Ws;
                        mask |= ok ? (1u << i) : 0u;
                    }
                } else {
                    mask = 0xffffffffu;
                }
                ooff = out_base + (int)n * p.out_n + at * p.out_t + ah * p.out_h + aw * p.out_w;
            }
            sRowBase[r] = base;
            sRowMask[r] = mask;
            sOutOff[r] = ooff;
            my_ooff = ooff;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < Cfg::NX; ++i) {
            rb[i] = sRowBase[i * 32 + lrow] + (MODE == 0 ? lslot * 8 : 0);
            rm[i] = sRowMask[i * 32 + lrow];
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) ooff[b] = sOutOff[wm * 64 + b * 16 + l15];
        __syncthreads();  // the row table has been consumed; buffer 0 may be overwritten
        }

        f32x4 acc[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

        int ti = 0, c0 = 0;
        if (MODE == 0) {
            ti = cbeg / cpt;
            c0 = (cbeg - ti * cpt) * 64;
        }
        if (cbeg < cend) issue_chunk(ti, c0, cbeg, 0);

        int buf = 0;
        for (int c = cbeg; c < cend; ++c) {
            // chunk c has landed for every wave (the barrier drains this wave's LDS-DMA first), and every
            // wave is done reading the other buffer (it was read one iteration ago)
            __syncthreads();
            int nti = ti, nc0 = c0 + 64;
            if (MODE == 0 && nc0 >= p.Cs) {
                nc0 = 0;
                nti = ti + 1;
            }
            if (c + 1 < cend) issue_chunk(nti, nc0, c + 1, buf ^ 1);

            const char* sW = smem + buf * Cfg::BUF_BYTES;
            const char* sX = sW + BCO * ROWB;
#if VC_IGEMM_PIPE
            {   // both k-steps' fragments as two sets, the second read behind the first's MFMAs (pinned)
                bf16x8 fa[2][4], fb[2][4];
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const int slot = ((ks << 2) + lg) ^ lswz;
#pragma unroll
                    for (int a = 0; a < 4; ++a)
                        fa[ks][a] = *(const bf16x8*)(sW + (wco * 64 + a * 16 + l15) * ROWB + (slot << 4));
#pragma unroll
                    for (int b = 0; b < 4; ++b)
                        fb[ks][b] = *(const bf16x8*)(sX + (wm * 64 + b * 16 + l15) * ROWB + (slot << 4));
                }
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int a = 0; a < 4; ++a)
#pragma unroll
                        for (int b = 0; b < 4; ++b)
                            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[ks][a], fb[ks][b], acc[a][b], 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                    if (i < 4) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                }
            }
#else
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
#endif
            ti = nti;
            c0 = nc0;
            buf ^= 1;
        }

        // ---- epilogue ----
        // acc[a][b][j] = D[co = co0 + wco*64 + a*16 + lg*4 + j][m = m0 + wm*64 + b*16 + l15]
        if (p.partial != nullptr) {
            float* slab = p.partial + (size_t)zsplit * p.partial_stride;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                if (ooff[b] < 0) continue;
#pragma unroll
                for (int a = 0; a < 4; ++a) *(f32x4*)(slab + (size_t)ooff[b] + cobase + a * 16) = acc[a][b];
            }
            __syncthreads();  // the operand tiles have been read before the next tile's row table lands on them
            continue;
        }

        if (p.ep_params != nullptr) {
            // inference: BatchNorm with fixed (running) statistics is a per-channel scale and shift, applied with the
            // ReLU on the fp32 accumulators, so the conv output is rounded to bf16 once and no separate pass reads it back
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const f32x4 s4 = *(const f32x4*)(p.ep_params + 2 * p.Cn + cobase + a * 16);
                const f32x4 h4 = *(const f32x4*)(p.ep_params + 3 * p.Cn + cobase + a * 16);
#pragma unroll
                for (int b = 0; b < 4; ++b)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float v = acc[a][b][j] * s4[j] + h4[j];
                        if (p.ep_relu && p.ep_res == nullptr) v = v > 0.f ? v : 0.f;  // with a residual: after the add
                        acc[a][b][j] = v;
                    }
            }
        }
        if (p.stats != nullptr) {
            // four channels at a time (packed fp32 adds and fmas): this accumulation sits on the tile's critical path
            if (m0 + BM <= M) {  // whole tile inside the tensor (uniform): no row tests
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        const f32x4 v = acc[a][b] - st_h[a];
                        st_s[a] += v;
                        st_q[a] += v * v;
                    }
            } else {  // rows beyond M hold exact zeros, which a shift would turn into -shift: skipped
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        if (ooff[b] < 0) continue;
                        const f32x4 v = acc[a][b] - st_h[a];
                        st_s[a] += v;
                        st_q[a] += v * v;
                    }
            }
        }

        // ---- store through LDS: the accumulator layout gives a lane 4 channels (8 bytes) of one position; staged,
        // every global access is a whole 16-byte piece of a channels-last row, 16 (or 8) lanes per row.  With
        // `accumulate` the old output takes the same way in, and the add and the single rounding happen on the fp32
        // accumulators.
        constexpr int OPITCH = BCO * 2 + 16;  // bytes per staged row
        constexpr int PPR = BCO / 8;          // 16-byte pieces per row
        static_assert(Cfg::STAGE_BYTES <= Cfg::LDS_BYTES, "staging tile and row table must fit");
        __syncthreads();  // every wave has finished reading the operand tiles
        int* sOut = (int*)(smem + BM * OPITCH);
        if (tid < BM) sOut[tid] = my_ooff;
        __syncthreads();
        const bf16* addsrc = p.ep_res != nullptr ? p.ep_res : (p.accumulate ? p.out : nullptr);  // what is added in
        if (addsrc != nullptr) {
            for (int i = tid; i < BM * PPR; i += 256) {
                const int r = i / PPR, c8 = i - r * PPR;
                const int oo = sOut[r];
                if (oo >= 0) *(bf16x8*)(smem + r * OPITCH + c8 * 16) = *(const bf16x8*)(addsrc + (size_t)oo + co0 + c8 * 8);
            }
            __syncthreads();
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) {
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                bf16x4* q = (bf16x4*)(smem + (wm * 64 + b * 16 + l15) * OPITCH + (wco * 64 + a * 16 + lg * 4) * 2);
                f32x4 v = acc[a][b];
                if (addsrc != nullptr) {
                    const bf16x4 old = *q;
                    v[0] += bf2f(old[0]);
                    v[1] += bf2f(old[1]);
                    v[2] += bf2f(old[2]);
                    v[3] += bf2f(old[3]);
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
        for (int i = tid; i < BM * PPR; i += 256) {
            const int r = i / PPR, c8 = i - r * PPR;
            const int oo = sOut[r];
            if (oo >= 0) *(bf16x8*)(p.out + (size_t)oo + co0 + c8 * 8) = *(const bf16x8*)(smem + r * OPITCH + c8 * 16);
        }
        __syncthreads();  // the staged tile has been read out before the next tile's row table is written
    }

    if (p.stats != nullptr && p.partial == nullptr) {
        // lanes -> wave (shuffles) -> workgroup (LDS: the waves that share a co range) -> one float atomic per channel
        // and moment per workgroup.  (Device-scope float atomics cost ~40 ps each chip-wide: 512 per workgroup, as
        // one set per wave made, were 18 us of a 65 us pointwise conv.)
        float* sred = (float*)smem;  // [WM][2][BCO]; the staged tile was read out before the loop's last barrier
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
        for (int i = tid; i < 2 * BCO; i += 256) {
            const int mom = i / BCO, c = i - mom * BCO;
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < Cfg::WM; ++w) v += sred[(w * 2 + mom) * BCO + c];
            vc_stat_add(p.stats, blockIdx.x, p.Cn, mom * p.Cn + co0 + c, v);
        }
    }
    if (p.partial == nullptr) vc_bn_fin_last<4>(p.fin, p.stats, p.Cn);
}

template <int BCO, int BM, int MODE>
int launch_igemm(const VcIgemmParams& p, hipStream_t stream) {
    using Cfg = IgemmCfg<BCO, BM>;
    dim3 grid((p.M + BM - 1) / BM, p.Cn / BCO, (p.splits > 1 ? p.splits : 1) * (p.ncls > 1 ? p.ncls : 1));
    const bool one_chunk = MODE == 0 && p.ncls <= 1 && p.ntaps * (p.Cs >> 6) == 1;
    if (p.ncls <= 1 && p.splits <= 1) {
        // persistent over tiles: all workgroups resident (2 per CU, 4 with a single buffer), same tile count each
        const long resident = 256l * (one_chunk ? 4 : 2);
        const long cap_x = resident / grid.y > 0 ? resident / grid.y : 1;
        if ((long)grid.x > cap_x) {
            const long per = ((long)grid.x + cap_x - 1) / cap_x;
            grid.x = (unsigned)(((long)grid.x + per - 1) / per);
        }
    }
    (void)hipGetLastError();  // drop stale errors left by other users of the runtime
    hipLaunchKernelGGL((igemm_kernel<BCO, BM, MODE>), grid, dim3(256), one_chunk ? Cfg::LDS_BYTES_1 : Cfg::LDS_BYTES, stream, p);
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
    if (s > nchunks / 16) s = nchunks / 16;  // slabs cost output-sized traffic: only split long K loops
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
    if (mode == 0 && p.ncls > 1) {
        int tot = 0;
        for (int c = 0; c < p.ncls; ++c) tot += p.cls[c].ntaps;
        if (p.ncls > 8 || tot > 32) {
            vc_set_error("igemm: %d classes with %d taps do not fit one launch", p.ncls, tot);
            return -2;
        }
    } else if (mode == 0 && (p.ntaps < 1 || p.ntaps > 32)) {
        vc_set_error("igemm: ntaps=%d out of range", p.ntaps);
        return -2;
    }
    if (mode == 1) {
        VcIgemmParams s1 = p;
        s1.stats = vc_det_tag(s1.stats);
        return launch_igemm<64, 256, 1>(s1, stream);
    }
    VcIgemmParams q = p;
    q.stats = vc_det_tag(q.stats);
    q.pointwise = p.ncls <= 1 && p.ntaps == 1 && p.taps[0].ot == 0 && p.taps[0].oh == 0 && p.taps[0].ow == 0 &&
                  p.taps[0].src_off == 0 && p.st == 1 && p.sh == 1 && p.sw == 1 && p.Ts == p.Ta && p.Hs == p.Ha &&
                  p.Ws == p.Wa && p.out_w == p.Cn && p.out_h == p.Wa * p.Cn && p.out_t == p.Ha * p.Wa * p.Cn &&
                  p.out_n == p.Ta * p.Ha * p.Wa * p.Cn && getenv("VC_NO_POINTWISE") == nullptr;
    if (q.pointwise && p.splits <= 1 && p.partial == nullptr && getenv("VC_NO_PW_KERNEL") == nullptr) {
        VcPwParams w = {};  // the dense GEMM kernel (pw_conv.hip)
        w.src = p.src, w.wgt = p.wgt, w.out = p.out + p.out_base, w.stats = q.stats, w.stat_shift = p.stat_shift;
        w.fin = p.fin, w.ep_params = p.ep_params, w.ep_relu = p.ep_relu;
        w.ep_res = p.ep_res != nullptr ? p.ep_res + p.out_base : nullptr;
        w.M = p.M, w.K = p.Cs, w.Cn = p.Cn, w.accumulate = p.accumulate;
        return vc_launch_pw(w, stream);
    }
    if (p.Cn % 128 == 0 && p.Cn >= 128) return launch_igemm<128, 128, 0>(q, stream);
    return launch_igemm<64, 256, 0>(q, stream);
}
