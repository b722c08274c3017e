// Stem (7x7x7, 3 -> 64, stride (1,2,2)) weight gradient on bf16 MFMA (gfx950), image-resident form.
//
// GEMM view:  dWs[co][k] = sum_m dY[m][co] * Xs[pos(m) + off(k)],  k = (kt*4 + ah)*48 + j,  j = aw*12 + c12
// over the space-to-depth source xs [N][T+6][Hp][Wp][12] (24 bytes per position).  For a fixed (kt, ah) the 48
// values of j are CONTIGUOUS in xs, and consecutive output positions of one output row are 24 bytes apart: the
// im2col matrix of a run of output positions is just the raw image row read with a 24-byte row pitch.  So no
// im2col tile is ever built (the generic wgrad kernel gathers one through the vector-memory pipe and is bound
// by it): the raw rows sit in LDS and ds_read_b64_tr_b16 reads the B fragments from them directly.
//
// Workgroup = 7 waves, wave kt owns columns [kt*192, kt*192 + 192) of dWs for all 64 co: 48 accumulator tiles
// (192 VGPRs) that stay in registers over the workgroup's whole share of the batch (persistent workgroups);
// dY is read from HBM exactly once, xs ~1.4x.  Work unit = (sample n, group of RH output rows, range of TR output
// frames); per output frame t the chunk is RH x Wo positions.  LDS holds a ring of 8 source planes (frame t+kt
// for wave kt; each plane = RH+3 padded rows, one contiguous global range, filled by LDS-DMA), two dY chunks
// (LDS-DMA, 16-byte slots XOR-swizzled on the source side so that the transposing reads are conflict-free at a
// 128-byte pitch) and a per-row offset table.  One barrier per frame.
// Each workgroup ends by storing its 64 x 1344 partial into its own fp32 slab; the split-K reduce kernel sums them.
#include <stdlib.h>
#include <string.h>

#include "common.h"
#include "kernels.h"
#include "elementwise.h"

namespace {

__device__ __attribute__((aligned(128))) unsigned int vc_stem_zero_page[32];  // zero source of out-of-range DMA lanes

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef s16x4 __attribute__((address_space(3))) * lds_s16x4_p;

constexpr int NWAVE = 7;
constexpr int TAB_BYTES = 1024;  // up to 256 chunk rows

__global__ __launch_bounds__(NWAVE * 64) __attribute__((amdgpu_waves_per_eu(2, 2))) void stem_wgrad_kernel(
    const VcStemWgradParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // = kt
    const int l15 = lane & 15, lg = lane >> 4;
    const int q = l15 >> 2, pp = l15 & 3;
    const int RP = p.Wp * 24;              // bytes per source row
    const int DYB = p.KS * 32 * 128;       // bytes per dY chunk buffer
    const int DY_OFF = TAB_BYTES;
    const int PL_OFF = TAB_BYTES + 2 * DYB;
    const int PLB = p.plane_pieces * 1024;
    const gptr_t zero = (gptr_t)vc_stem_zero_page;

    // byte offset of chunk row r inside a plane (row-group-relative); rows past the chunk point at position 0
    // (their dY rows are zero-filled, the source value only has to be finite)
    int* tab = (int*)smem;
    for (int r = tid; r < p.KS * 32; r += NWAVE * 64) {
        const int hl = r / p.Wo;
        const int w = r - hl * p.Wo;
        tab[r] = r < p.RH * p.Wo ? hl * RP + w * 24 : 0;
    }

    // A operand (dY^T) lane addresses: row 4*lg + q of a 16-row group, 8 bytes at channel a*16 + 4*pp;
    // the 16-byte slot index is XORed with ((row >> 1) & 3) << 1 (the swizzle the DMA applied)
    int addrA[4];
    {
        const int row = 4 * lg + q;
        const int f = ((row >> 1) & 3) << 1;
#pragma unroll
        for (int a = 0; a < 4; ++a) addrA[a] = row * 128 + (((2 * a + (pp >> 1)) ^ f) << 4) + (pp & 1) * 8;
    }
    const int tabidx = (4 * lg + q) * 4;  // byte index into the table for k-step 0, half 0

    f32x4 acc[4][12];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 12; ++c) acc[a][c] = (f32x4){0.f, 0.f, 0.f, 0.f};

    auto issue_plane = [&](int n, int tp, int h0) {
        char* dst = smem + PL_OFF + (tp & 7) * PLB;
        const int rows_valid = min(p.RH + 3, p.Hp - h0);
        const int valid_bytes = rows_valid * RP;
        const char* src0 = (const char*)(p.xs + ((size_t)((n * p.Tp + tp) * p.Hp + h0) * p.Wp) * 12);
        for (int pc = wave; pc < p.plane_pieces; pc += NWAVE) {
            const int o = pc * 1024 + lane * 16;
            const gptr_t src = o < valid_bytes ? (gptr_t)(src0 + o) : zero;
            __builtin_amdgcn_global_load_lds(src, (lptr_t)(dst + pc * 1024), 16, 0, 0);
        }
    };
    auto issue_dy = [&](int n, int t, int h0, int buf) {
        char* dst = smem + DY_OFF + buf * DYB;
        const long m0 = ((long)(n * p.T + t) * p.Ho + h0) * p.Wo;
        const int cpv = min(p.RH, p.Ho - h0) * p.Wo;
        for (int pc = wave; pc < p.KS * 4; pc += NWAVE) {
            const int r = pc * 8 + (lane >> 3);
            const int s = (lane & 7) ^ (((r >> 1) & 3) << 1);
            const gptr_t src = r < cpv ? (gptr_t)(p.dy + ((m0 + r) * 64 + s * 8)) : zero;
            __builtin_amdgcn_global_load_lds(src, (lptr_t)(dst + pc * 1024), 16, 0, 0);
        }
    };

    unsigned bx = blockIdx.x;  // XCD k takes the k-th contiguous eighth of each round of units (stem_fwd.hip)
    if (p.xcd_map) bx = (bx & 7u) * (gridDim.x >> 3) + (bx >> 3);
    for (int u = (int)bx; u < p.units; u += gridDim.x) {
        const int tr = u % p.ntr;
        const int hg = (u / p.ntr) % p.nhg;
        const int n = u / (p.ntr * p.nhg);
        const int t0 = tr * p.TR;
        const int t1 = min(p.T, t0 + p.TR);
        const int h0 = hg * p.RH;
        __syncthreads();  // the previous unit's last frame (and the table build) is done: the ring may be refilled
        for (int tp = t0; tp < t0 + 7; ++tp) issue_plane(n, tp, h0);
        issue_dy(n, t0, h0, 0);
        for (int t = t0; t < t1; ++t) {
            __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0) lgkmcnt(0): everything issued so far has landed
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (t + 1 < t1) {
                issue_plane(n, t + 7, h0);  // into the slot of plane t-1, which every wave has left
                issue_dy(n, t + 1, h0, (t + 1 - t0) & 1);
            }
            const char* plane = smem + PL_OFF + ((t + wave) & 7) * PLB + pp * 8;
            const char* dyb = smem + DY_OFF + ((t - t0) & 1) * DYB;
#pragma unroll 1
            for (int ks = 0; ks < p.KS; ++ks) {
                const int po0 = *(const int*)(smem + tabidx + ks * 128);
                const int po1 = *(const int*)(smem + tabidx + ks * 128 + 64);
                bf16x8 fa[4];
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const char* r0 = dyb + addrA[a] + ks * (32 * 128);
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(r0));
                    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(r0 + 16 * 128));
                    fa[a] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
                }
#pragma unroll
                for (int ah = 0; ah < 4; ++ah) {
                    const char* b0 = plane + po0 + ah * RP;
                    const char* b1 = plane + po1 + ah * RP;
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(b0 + j * 32));
                        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(b1 + j * 32));
                        const bf16x8 fb = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
                        for (int a = 0; a < 4; ++a)
                            acc[a][ah * 3 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[a], fb, acc[a][ah * 3 + j], 0, 0, 0);
                    }
                }
            }
        }
    }

    // ---- acc[a][c][j] = dWs[co = a*16 + lg*4 + j][k = kt*192 + c*16 + l15] of this workgroup's share
    float* slab = p.slabs + (size_t)blockIdx.x * (64 * 1344);
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float* row = slab + (size_t)(a * 16 + lg * 4 + j) * 1344 + wave * 192 + l15;
#pragma unroll
            for (int c = 0; c < 12; ++c) row[c * 16] = acc[a][c][j];
        }
}

}  // namespace

// Geometry / plan.  Returns 0 when the image-resident form does not apply (then the generic kernel runs).
int vc_stem_wgrad_plan(int N, int T, int Ho, int Wo, int Hp, int Wp, int Co, VcStemWgradParams* out) {
    static const bool off = getenv("VC_NO_STEM_WGRAD2") != nullptr;
    if (off || Co != 64 || (Wp & 1)) return 0;
    VcStemWgradParams p;
    memset(&p, 0, sizeof(p));
    const int RP = Wp * 24;
    int best_rh = 0;
    double best_eff = 0.0;
    for (int RH = 1; RH <= Ho && RH <= 16; ++RH) {
        const int CP = RH * Wo;
        const int KS = (CP + 31) / 32;
        if (KS * 32 > 256) break;
        const int pieces = ((RH + 3) * RP + 1023) / 1024;
        const long lds = TAB_BYTES + 2L * KS * 32 * 128 + 8L * pieces * 1024;
        if (lds > 160 * 1024) break;
        const int nhg = (Ho + RH - 1) / RH;
        // useful share of the MFMA work (chunk padding, ragged last row group) and of the source reads
        const double eff = ((double)Ho * Wo / ((double)nhg * KS * 32)) * ((double)RH / (RH + 1.0));
        if (eff > best_eff) best_eff = eff, best_rh = RH;
    }
    if (best_rh == 0) return 0;
    p.N = N, p.T = T, p.Ho = Ho, p.Wo = Wo, p.Tp = T + 6, p.Hp = Hp, p.Wp = Wp;
    p.RH = best_rh;
    p.KS = (best_rh * Wo + 31) / 32;
    p.plane_pieces = ((best_rh + 3) * RP + 1023) / 1024;
    p.nhg = (Ho + best_rh - 1) / best_rh;
    // frame ranges: as few as give every workgroup the same number of units (each range re-reads 6 planes)
    const int target = getenv("VC_STEM_WGRAD_WGS") ? atoi(getenv("VC_STEM_WGRAD_WGS")) : 256;
    int ntr = 1;
    double best_cost = 1e30;
    for (int cand = 1; cand <= T; ++cand) {
        const int TR = (T + cand - 1) / cand;
        const int nr = (T + TR - 1) / TR;
        const long units = (long)N * p.nhg * nr;
        const long wgs = units < target ? units : target;
        const long per = (units + wgs - 1) / wgs;  // units of the busiest workgroup
        const double cost = (double)per * (TR + 1.5);  // frames + ring start-up
        if (cost < best_cost) best_cost = cost, ntr = nr, p.TR = TR;
    }
    p.ntr = ntr;
    p.units = N * p.nhg * ntr;
    p.grid = p.units < target ? p.units : target;
    static const bool no_xcd = getenv("VC_NO_XCD_MAP") != nullptr || getenv("VC_STEM_NO_XCD_MAP") != nullptr;
    p.xcd_map = (p.grid % 8 == 0 && p.grid >= 16 && !no_xcd) ? 1 : 0;
    p.lds_bytes = TAB_BYTES + 2 * p.KS * 32 * 128 + 8 * p.plane_pieces * 1024;
    *out = p;
    return 1;
}

size_t vc_stem_wgrad_slab_bytes(const VcStemWgradParams& p) { return (size_t)p.grid * 64 * 1344 * sizeof(float); }

int vc_launch_stem_wgrad(VcStemWgradParams& p, float* dws, hipStream_t stream) {
    static VcPerDeviceOnce once;  // thread-safe, once per device
    once.run([&]() { (void)hipFuncSetAttribute((const void*)stem_wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
    (void)hipGetLastError();
    hipLaunchKernelGGL(stem_wgrad_kernel, dim3(p.grid), dim3(NWAVE * 64), p.lds_bytes, stream, p);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        vc_set_error("stem wgrad launch failed: %s", hipGetErrorString(e));
        return -1;
    }
    // dws += sum of the workgroups' slabs
    return vc_ew_splitk_reduce_wgrad(p.slabs, p.grid, (long long)64 * 1344, dws, 64L * 1344, 0, 0, 0xffffffffu, stream);
}
