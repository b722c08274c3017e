// Stem (7x7x7, 3 -> 64, stride (1,2,2)) forward on bf16 MFMA (gfx950), image-resident form.
//
// Y[m][co] = sum_k Xs[pos(m) + off(k)] * Ws[co][k],  k = kt*192 + ah*48 + j,  j = aw*12 + c12, over the
// space-to-depth source xs [N][T+6][Hp][Wp][12] (24 bytes per position).  For a fixed (kt, ah) the 48 values of
// j are contiguous in xs, so the 8 consecutive k of an MFMA B-fragment lane are 16 contiguous bytes of the raw
// image row (8-byte aligned: two ds_read_b64).  The implicit-GEMM kernel this replaces gathered an im2col tile
// through the vector-memory pipe in 8-byte pieces and was bound by that pipe; here the raw rows of the 7
// source frames sit in LDS (a ring of 10 plane slots filled by LDS-DMA, each plane = RH+3 padded rows = one
// contiguous global range) and no im2col tile exists.
//
// Persistent workgroups of 7 waves.  Work unit = (sample, group of RH output rows, range of TR frames); frames
// are processed in pairs: a chunk of RH x Wo <= 224 positions per frame, wave w owns positions [32w, 32w+32)
// of both frames x all 64 co = 16 accumulator tiles.  The weights stream through LDS one temporal tap at a
// time (64 x 192 bf16, 416-byte row pitch produced on the DMA's source side, three tiles in flight with a counted
// s_waitcnt vmcnt, one barrier per tap); A = weights (so a lane ends with 4 consecutive channels of one position), B = image.
// Epilogue per frame pair: bf16 channels-last stores straight from the accumulators; BatchNorm sum / sum of
// squares are kept in registers over the workgroup's whole share and leave as 128 atomics per workgroup.
#include <stdlib.h>
#include <string.h>

#include "common.h"
#include "kernels.h"

namespace {

__device__ __attribute__((aligned(128))) unsigned int vc_stemf_zero_page[32];  // zero source of out-of-range DMA lanes

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

#ifndef VC_STEMF_ABLATE
#define VC_STEMF_ABLATE 0  // timing experiments only: 1 = weight tiles not refilled, 2 = image planes not refilled,
                           // 4 / 8 = no image / weight fragment reads, 16 = no MFMAs, 32 = no per-tap barrier
#endif
#ifndef VC_STEMF_B128
#define VC_STEMF_B128 0  // image fragments as one 8-byte-aligned ds_read_b128 instead of two ds_read_b64
#endif
#ifndef VC_STEMF_PIPE
#define VC_STEMF_PIPE 1  // explicit two-set fragment pipeline over the k-steps of a tap (0: the plain loop; 0.532 -> 0.510 ms in the step)
#endif
#ifndef VC_STEMF_NT
#define VC_STEMF_NT 2  // 16-position tiles per wave and frame: 2 -> 7 waves of 32 positions, 4 -> 4 waves of 64
#endif
#ifndef VC_STEMF_FG
#define VC_STEMF_FG 1  // frame pairs in flight: 2 -> two groups of waves, each on its own pair of frames (with NT = 4)
#endif
constexpr int NT = VC_STEMF_NT;
constexpr int FG = VC_STEMF_FG;
constexpr int PW = NT * 16;          // positions per wave and frame
constexpr int WPG = NT == 2 ? 7 : 4; // waves per group: they share a chunk of WPG * PW positions of the group's two frames
constexpr int NWAVE = WPG * FG;
constexpr int WPITCH = 416;          // bytes per weight row in LDS: 384 data + 32 pad (conflict-free ds_read_b128)
constexpr int WPIECES = 26;          // 64 rows x 26 slots of 16 bytes = 26 pieces of 1 KiB
constexpr int WBUF = WPIECES * 1024;
constexpr int NSLOT = 6 + 2 * FG;    // plane ring: the 8 planes of a frame pair (10 of two pairs); the next step's new
                                     // planes replace the oldest ones once the taps that read them are done
constexpr int NWB = FG == 2 ? 2 : 3; // weight tiles in flight: tap g is computed while g+1 (and g+2) stream in
constexpr int WPW = (WPIECES + NWAVE - 1) / NWAVE;  // weight DMA instructions per wave and tile (26 pieces + dummies)
constexpr int W_OFF = 0;
constexpr int DUMMY_OFF = NWB * WBUF;  // 1 KiB that the dummy DMA instructions land in
constexpr int PL_OFF = NWB * WBUF + 1024;

__global__ __launch_bounds__(NWAVE * 64) __attribute__((amdgpu_waves_per_eu(NWAVE > 4 ? 2 : 1, 2))) void stem_fwd_kernel(
    const VcStemFwdParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lg = lane >> 4;
    const int RP = p.Wp * 24;
    const int PLB = p.plane_pieces * 1024;
    const int CP = p.RH * p.Wo;
    const gptr_t zero = (gptr_t)vc_stemf_zero_page;

    // this lane's two positions (B columns) of the chunk, and the k offsets of its 8-element group per k-step
    int mloc[NT], posoff[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        mloc[nt] = (wave % WPG) * PW + nt * 16 + l15;
        const int hl = mloc[nt] / p.Wo;
        const int w = mloc[nt] - hl * p.Wo;
        posoff[nt] = mloc[nt] < CP ? hl * RP + w * 24 : 0;
    }
    int koff[6];
#pragma unroll
    for (int ks = 0; ks < 6; ++ks) {
        const int k = ks * 32 + lg * 8;
        const int ah = k / 48;
        koff[ks] = ah * RP + (k - ah * 48) * 2;
    }
    const int aoff = l15 * WPITCH + lg * 16;
    const int fg2 = 2 * (wave / WPG);  // this wave's frames of a step: t + fg2, t + fg2 + 1

    f32x4 ssum[4], ssq[4];  // BatchNorm partial sums of this lane's channels co = a*16 + lg*4 + j (packed fp32 math)
    f32x4 sshift[4];        // their shift (bn_fin.h), 0 without one
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        ssum[a] = ssq[a] = (f32x4){0.f, 0.f, 0.f, 0.f};
        sshift[a] = (p.stats != nullptr && p.stat_shift != nullptr) ? *(const f32x4*)(p.stat_shift + a * 16 + (lane >> 4) * 4)
                                                                    : (f32x4){0.f, 0.f, 0.f, 0.f};
    }

    auto issue_plane = [&](int n, int tp, int h0) {
        char* dst = smem + PL_OFF + (tp % NSLOT) * PLB;
        const int rows_valid = min(p.RH + 3, p.Hp - h0);
        const int valid_bytes = tp < p.Tp ? rows_valid * RP : 0;
        const char* src0 = (const char*)(p.xs + ((size_t)((n * p.Tp + min(tp, p.Tp - 1)) * p.Hp + h0) * p.Wp) * 12);
        for (int pc = wave; pc < p.plane_pieces; pc += NWAVE) {
            const int o = pc * 1024 + lane * 16;
            const gptr_t src = o < valid_bytes ? (gptr_t)(src0 + o) : zero;
            __builtin_amdgcn_global_load_lds(src, (lptr_t)(dst + pc * 1024), 16, 0, 0);
        }
    };
    // every wave issues exactly WPW instructions per tile (the last two go to a dummy KiB), so that a counted
    // s_waitcnt vmcnt(WPW) means "everything but the youngest tile has landed" in every wave
    auto issue_w = [&](int kt, int buf) {
        char* dst = smem + W_OFF + buf * WBUF;
#pragma unroll
        for (int i = 0; i < WPW; ++i) {
            const int pc = wave + NWAVE * i;
            const unsigned S = (unsigned)(pc * 64 + lane);
            const unsigned row = (S * 2521u) >> 16;  // S / 26 for S < 1664
            const unsigned part = S - row * 26u;
            const bool real = pc < WPIECES;
            const gptr_t src = real && part < 24u ? (gptr_t)(p.ws + ((size_t)row * 1344 + kt * 192 + part * 8)) : zero;
            __builtin_amdgcn_global_load_lds(src, (lptr_t)(real ? dst + pc * 1024 : smem + DUMMY_OFF), 16, 0, 0);
        }
    };

    // Workgroups are dealt round-robin to the 8 XCDs; neighbouring units share source rows (3 of a row group's RH + 3
    // rows, 6 of a frame range's TR + 6 frames), so XCD k takes the k-th contiguous eighth of each round of units and
    // the shared rows are hits in its own L2 (the source was read 2.2x from HBM with the plain order).
    unsigned bx = blockIdx.x;
    if (p.xcd_map) bx = (bx & 7u) * (gridDim.x >> 3) + (bx >> 3);
    for (int u = (int)bx; u < p.units; u += gridDim.x) {
        const int tr = u % p.ntr;
        const int hg = (u / p.ntr) % p.nhg;
        const int n = u / (p.ntr * p.nhg);
        const int t0 = tr * p.TR;
        const int t1 = min(p.T, t0 + p.TR);
        const int h0 = hg * p.RH;
        const int cpv = min(p.RH, p.Ho - h0) * p.Wo;  // valid positions of this row group
        __syncthreads();  // the previous unit's last reads are done: ring and weight buffers may be refilled
        for (int tp = t0; tp < t0 + NSLOT; ++tp) issue_plane(n, tp, h0);
        const int nsteps = ((t1 - t0 + 2 * FG - 1) / (2 * FG)) * 7;  // temporal-tap steps of this unit
        issue_w(0, 0);
        if (NWB == 3 && nsteps > 1) issue_w(1, 1);
        int g = 0;  // step counter of the unit: weight tile of step g sits in buffer g % NWB
        for (int t = t0; t < t1; t += 2 * FG) {
            f32x4 acc[4][2][NT];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int f = 0; f < 2; ++f)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[a][f][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
            for (int kt = 0; kt < 7; ++kt, ++g) {
                // tile g has landed once at most the younger tile g+1 is in flight; the step after the plane
                // refill (kt == 3) and the unit's last step drain everything
                static_assert(WPW == 4 || WPW == 7, "counted wait below");
                // (two frame pairs: one tile ahead only, and the plane refills of the step before are waited for as well)
                if (FG == 2 || kt == 3 || g + 1 >= nsteps) __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0) lgkmcnt(0)
                else __builtin_amdgcn_s_waitcnt(WPW == 4 ? 0x0074 : 0x0077);         // vmcnt(WPW) lgkmcnt(0)
                if (!(VC_STEMF_ABLATE & 32)) __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                const int wbuf = g % NWB;
#if !(VC_STEMF_ABLATE & 2)
                if (FG == 1) {
                    if (kt == 2 && t + 2 < t1) {  // planes t and t+1 are dead after tap 1: the next pair's new planes go there
                        issue_plane(n, t + 8, h0);
                        issue_plane(n, t + 9, h0);
                    }
                } else if (t + 4 < t1) {
                    // planes t .. t+3 are dead after tap 3 (plane t+j is read by frame t+i at tap j-i): the next step's four
                    // new planes take their slots, two per step so that the DMA queue is not hit by all of them at once
                    if (kt == 4) issue_plane(n, t + 10, h0), issue_plane(n, t + 11, h0);
                    if (kt == 5) issue_plane(n, t + 12, h0), issue_plane(n, t + 13, h0);
                }
#endif
#if !(VC_STEMF_ABLATE & 1)
                if (NWB == 3) {
                    if (g + 2 < nsteps) issue_w((kt + 2) % 7, (g + 2) % NWB);  // buffer of tile g-1, read one step ago
                } else {
                    if (g + 1 < nsteps) issue_w((kt + 1) % 7, (g + 1) % NWB);  // buffer of tile g-1, read one step ago
                }
#endif
                const char* sW = smem + W_OFF + wbuf * WBUF + aoff;
                const char* pl0 = smem + PL_OFF + ((t + fg2 + kt) % NSLOT) * PLB;
                const char* pl1 = smem + PL_OFF + ((t + fg2 + 1 + kt) % NSLOT) * PLB;
#if VC_STEMF_PIPE
                // explicit software pipeline over the six k-steps of the tap: the fragments of k-step ks+1 are read while
                // the MFMAs of ks run (two fragment sets), the interleave pinned below
                bf16x8 fa[2][4], fb[2][2][NT];
                auto load_ks = [&](int ks, bf16x8 (&qa)[4], bf16x8 (&qb)[2][NT]) {
#pragma unroll
                    for (int a = 0; a < 4; ++a) qa[a] = *(const bf16x8*)(sW + a * 16 * WPITCH + ks * 64);
#pragma unroll
                    for (int f = 0; f < 2; ++f)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) {
                            const char* bp = (f ? pl1 : pl0) + posoff[nt] + koff[ks];
                            const bf16x4 lo = *(const bf16x4*)bp;
                            const bf16x4 hi = *(const bf16x4*)(bp + 8);
                            qb[f][nt] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                        }
                };
                load_ks(0, fa[0], fb[0]);
#pragma unroll
                for (int ks = 0; ks < 6; ++ks) {
                    if (ks + 1 < 6) load_ks(ks + 1, fa[(ks + 1) & 1], fb[(ks + 1) & 1]);
#pragma unroll
                    for (int f = 0; f < 2; ++f)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                            for (int a = 0; a < 4; ++a)
                                acc[a][f][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[ks & 1][a], fb[ks & 1][f][nt], acc[a][f][nt], 0, 0, 0);
                }
                if (NT == 2) {
                    __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);  // k-step 0: 4 weight + 8 image reads
#pragma unroll
                    for (int i = 0; i < 24; ++i) {  // 24 groups of 4 MFMAs; the 60 reads of k-steps 1..5 ride along, 3 per group
                        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                        if (i < 20) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
                    }
                }
#else
#pragma unroll
                for (int ks = 0; ks < 6; ++ks) {
                    bf16x8 fa[4];
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        if (VC_STEMF_ABLATE & 8) asm volatile("" : "=v"(fa[a]));  // timing only: no weight fragment reads
                        else fa[a] = *(const bf16x8*)(sW + a * 16 * WPITCH + ks * 64);
                    }
#pragma unroll
                    for (int f = 0; f < 2; ++f)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) {
                            const char* bp = (f ? pl1 : pl0) + posoff[nt] + koff[ks];
                            bf16x8 fb;
                            if (VC_STEMF_ABLATE & 4) {  // timing only: no image fragment reads
                                asm volatile("" : "=v"(fb));
                            } else {
#if VC_STEMF_B128
                                typedef bf16x8 __attribute__((aligned(8))) bf16x8_a8;
                                fb = *(const bf16x8_a8*)bp;  // one 16-byte read at 8-byte alignment
#else
                                const bf16x4 lo = *(const bf16x4*)bp;
                                const bf16x4 hi = *(const bf16x4*)(bp + 8);
                                fb = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
#endif
                            }
#pragma unroll
                            for (int a = 0; a < 4; ++a) {
                                if (VC_STEMF_ABLATE & 16) asm volatile("" : "+v"(acc[a][f][nt]) : "v"(fa[a]), "v"(fb));  // no MFMA
                                else acc[a][f][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[a], fb, acc[a][f][nt], 0, 0, 0);
                            }
                        }
                }
#endif
                if (NT == 4) {
                    // pin the read / MFMA interleave of the tap (one basic block): per k-step 4 weight + 16 image reads for
                    // 32 MFMAs; the first k-step's weights and two image fragments go out up front, then 7 reads per
                    // three fragments' 12 MFMAs (left alone the compiler reads each fragment right before its MFMAs)
                    __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
                    for (int i = 0; i < 48; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                        if (i % 3 == 0) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
                        else __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    }
                }
            }
            // ---- acc[a][f][nt][j] = Y[frame t+f][position mloc[nt]][co = a*16 + lg*4 + j]
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                if (t + fg2 + f >= t1) continue;
                const long mrow = ((long)(n * p.T + t + fg2 + f) * p.Ho + h0) * p.Wo;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    if (mloc[nt] >= cpv) continue;
                    bf16* dst = p.y + (mrow + mloc[nt]) * 64 + lg * 4;
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        const f32x4 v = acc[a][f][nt];
                        *(bf16x4*)(dst + a * 16) = (bf16x4){f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
                        const f32x4 u = v - sshift[a];  // shifted sums (bn_fin.h)
                        ssum[a] += u;
                        ssq[a] += u * u;
                    }
                }
            }
        }
    }

    // ---- BatchNorm sums: 16 positions (l15) -> wave -> workgroup -> one atomic per channel and moment
    if (p.stats != nullptr) {
        __syncthreads();
        float* sred = (float*)smem;  // [NWAVE][2][64]
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float s = ssum[a][j], q2 = ssq[a][j];
#pragma unroll
                for (int o = 8; o >= 1; o >>= 1) {
                    s += __shfl_xor(s, o, 16);
                    q2 += __shfl_xor(q2, o, 16);
                }
                if (l15 == 0) {
                    sred[wave * 128 + a * 16 + lg * 4 + j] = s;
                    sred[wave * 128 + 64 + a * 16 + lg * 4 + j] = q2;
                }
            }
        __syncthreads();
        if (tid < 128) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < NWAVE; ++w) s += sred[w * 128 + tid];
            vc_stat_add(p.stats, blockIdx.x, 64, (tid < 64 ? 0 : 64) + (tid & 63), s);
        }
    }
    vc_bn_fin_last(p.fin, p.stats, 64);
}

}  // namespace

// Geometry / plan.  Returns 0 when the image-resident form does not apply (then the implicit GEMM runs).
int vc_stem_fwd_plan(int N, int T, int Ho, int Wo, int Hp, int Wp, int Co, VcStemFwdParams* out) {
    static const bool off = getenv("VC_NO_STEM_FWD2") != nullptr;
    if (off || Co != 64 || (Wp & 1)) return 0;
    VcStemFwdParams p;
    memset(&p, 0, sizeof(p));
    const int RP = Wp * 24;
    int best_rh = 0;
    double best_eff = 0.0;
    for (int RH = 1; RH <= Ho && RH * Wo <= WPG * PW; ++RH) {
        const int pieces = ((RH + 3) * RP + 1023) / 1024;
        const long lds = (long)NWB * WBUF + 1024 + (long)NSLOT * pieces * 1024;
        if (lds > 160 * 1024) break;
        const int nhg = (Ho + RH - 1) / RH;
        const double eff = ((double)Ho * Wo / ((double)nhg * WPG * PW)) * ((double)RH / (RH + 1.0));
        if (eff > best_eff) best_eff = eff, best_rh = RH;
    }
    if (best_rh == 0) return 0;
    p.N = N, p.T = T, p.Ho = Ho, p.Wo = Wo, p.Tp = T + 6, p.Hp = Hp, p.Wp = Wp;
    p.RH = best_rh;
    p.plane_pieces = ((best_rh + 3) * RP + 1023) / 1024;
    p.nhg = (Ho + best_rh - 1) / best_rh;
    const int target = getenv("VC_STEM_FWD_WGS") ? atoi(getenv("VC_STEM_FWD_WGS")) : 256;
    double best_cost = 1e30;
    for (int cand = 1; cand <= T; ++cand) {
        const int TR = (T + cand - 1) / cand;
        const int nr = (T + TR - 1) / TR;
        const long units = (long)N * p.nhg * nr;
        const long wgs = units < target ? units : target;
        const long per = (units + wgs - 1) / wgs;
        const double cost = (double)per * ((TR + 2 * FG - 1) / (2 * FG) + 0.7);  // steps (FG frame pairs each) + ring start-up
        if (cost < best_cost) best_cost = cost, p.ntr = nr, p.TR = TR;
    }
    p.units = N * p.nhg * p.ntr;
    p.grid = p.units < target ? p.units : target;
    p.lds_bytes = NWB * WBUF + 1024 + NSLOT * p.plane_pieces * 1024;
    static const bool no_xcd = getenv("VC_NO_XCD_MAP") != nullptr || getenv("VC_STEM_NO_XCD_MAP") != nullptr;
    p.xcd_map = (p.grid % 8 == 0 && p.grid >= 16 && !no_xcd) ? 1 : 0;
    *out = p;
    return 1;
}

int vc_launch_stem_fwd(VcStemFwdParams& p, hipStream_t stream) {
    static VcPerDeviceOnce once;  // thread-safe, once per device
    once.run([&]() { (void)hipFuncSetAttribute((const void*)stem_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
    p.stats = vc_det_tag(p.stats);
    (void)hipGetLastError();
    hipLaunchKernelGGL(stem_fwd_kernel, dim3(p.grid), dim3(NWAVE * 64), p.lds_bytes, stream, p);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        vc_set_error("stem fwd launch failed: %s", hipGetErrorString(e));
        return -1;
    }
    return 0;
}
