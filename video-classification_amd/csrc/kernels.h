// Internal launch interfaces between capi.hip and the kernel translation units.
#pragma once
#include "common.h"
#include "bn_fin.h"

struct VcIgemmParams {
    const bf16* src;  // [N][Ts][Hs][Ws][Cs]
    const bf16* wgt;  // [Cn][Kw]
    bf16* out;
    float* stats;  // nullable: [VC_STAT_REPLICAS][2][Cn]
    const float* stat_shift;  // nullable: [Cn], subtracted from every value before it enters the sums (bn_fin.h)
    VcBnFin fin;   // fin.ticket nullable: the last workgroup finalizes the BatchNorm parameters (bn_fin.h)
    const float* ep_params;  // nullable: BatchNorm params [4][Cn] applied in the epilogue (out = scale * acc + shift)
    int ep_relu;             // ... followed by ReLU
    const bf16* ep_res;      // nullable (with ep_params): residual, laid out like out, added before the ReLU
    int M, Ta, Ha, Wa;
    VcFastDiv dWa, dHa, dTa;
    int Ts, Hs, Ws, Cs;
    int st, sh, sw;
    int ntaps, Kw, Cn;
    int out_n, out_t, out_h, out_w, out_base;
    int accumulate;
    int pointwise;  // set by vc_launch_igemm: one centred tap, unit stride, dense output -- row m is row m of src and out
    // split-K: blockIdx.z handles chunks [z*cps, (z+1)*cps); partial sums go to fp32 slabs
    // partial[z][same element offsets as out]; a finalize kernel sums, rounds and takes BN statistics
    int splits, cps;
    float* partial;
    long long partial_stride;
    // several output classes in one launch (the stride-parity classes of a strided dgrad): blockIdx.z =
    // class * max(splits, 1) + split; class c has its own output geometry and taps [tap0, tap0 + ntaps).
    // ncls <= 1: the single geometry above.
    int ncls;
    struct Class {
        int M, Ta, Ha, Wa;
        VcFastDiv dWa, dHa, dTa;
        int out_base, tap0, ntaps;
    } cls[8];
    VcTap taps[32];
};

// pw_conv.hip: the dense case of the above (one centred tap, unit stride, contiguous rows)
#ifndef VC_PW_WPE
#define VC_PW_WPE 3  // waves per SIMD the pointwise kernel's register allocation leaves room for
#endif
struct VcPwParams {
    const bf16* src;  // [M][K]
    const bf16* wgt;  // [Cn][K]
    bf16* out;        // [M][Cn]
    float* stats;
    const float* stat_shift;
    VcBnFin fin;
    const float* ep_params;
    int ep_relu;
    const bf16* ep_res;
    int M, K, Cn;
    int accumulate;
    const uint8_t* acc_bits;  // nullable (with accumulate): ReLU bits of out's elements; the old value counts where its bit is set
    // The reduction of a BatchNorm backward folded into the (data-gradient) launch that completes its incoming
    // gradient: out is the gradient reaching `relu(bn(red_y) [+ residual])`; the epilogue adds sum g and sum g * xhat
    // over its tile (g = out where the ReLU passed, xhat = (red_y - mean) * invstd) to red_sums -- what
    // chan_reduce_kernel<1, 3 or 2> would re-read out, red_y and the mask for.  red_sums nullable (with stats == nullptr
    // only); red_fin as fin.
    const bf16* red_y;          // [M][Cn]
    const float* red_params;    // [4][Cn]
    const uint8_t* red_bits;    // [M * Cn / 8] (mask mode 3), or null: scale * red_y + shift > 0 (mask mode 2)
    float* red_sums;            // [VC_STAT_REPLICAS][2][Cn]
    VcBnBwdFin red_fin;
    int xcd_map;  // 1: the co blocks of a row tile are neighbours in an XCD-contiguous workgroup order (pw_conv.hip)
};
int vc_launch_pw(const VcPwParams& p, hipStream_t stream);

struct VcWgradParams {
    const bf16* x;   // [N][Ts][Hs][Ws][Cs]
    const bf16* dy;  // [M][Co]
    float* dw;       // [Co][Kw] fp32, accumulated with atomics
    int M, Ta, Ha, Wa;
    VcFastDiv dWa, dHa, dTa;
    int Ts, Hs, Ws, Cs;
    int st, sh, sw;
    int ntaps, Kw, Co;
    int rows_per_split;  // multiple of 64
    int splits;
    float* partial;  // nullable: fp32 slabs [splits][Co][Kw] instead of atomics
    long long partial_stride;
    VcTap taps[32];
    int xcd_map;  // 1: XCD k takes the k-th contiguous share of the (z, y, x) workgroup order (wgrad_conv.hip)
};

// halo_conv.hip: stride-1 conv with the activation halo resident in LDS
struct VcHaloParams {
    const bf16* src;  // [NP*H*W][Cs]
    const bf16* wgt;  // [Cn][Kw]
    bf16* out;        // [NP*H*W][Cn]
    float* stats;     // nullable: [VC_STAT_REPLICAS][2][Cn]
    const float* stat_shift;  // nullable: [Cn], subtracted from every value before it enters the sums (bn_fin.h)
    VcBnFin fin;      // fin.ticket nullable: the last workgroup finalizes the BatchNorm parameters (bn_fin.h)
    const float* ep_params;  // nullable: BatchNorm params [4][Cn] applied in the epilogue (out = scale * acc + shift)
    int ep_relu;             // ... followed by ReLU
    const bf16* ep_res;      // nullable (with ep_params, not with accumulate): residual laid out like out, added before the ReLU
    int NP, T, H, W;  // NP = N*T planes of H*W rows
    int Cs, Cn, Kw, ntaps;
    int TT, SEG, HR, halo_rows, npieces, ntg, nseg;  // filled by vc_launch_halo from the plan
    int canon;        // +1 / -1: canonical 3x3x3 tap order (forward / mirrored), 0: generic table
    VcFastDiv dSEG, dT, dW;
    int pad, R, Wp;   // zero-padded image form: R h-rows per segment, Wp = W + 2, HR = (R + 2) * Wp
    VcFastDiv dHR, dWp;
    int accumulate;
    int splits, cps;
    float* partial;
    long long partial_stride;
    long long* dbg;   // nullable: 8 s_memtime stamps per workgroup (tuning aid)
    int xcd_map;      // set by the launcher: blockIdx.x -> tile so that each XCD works on a contiguous range of tiles
    VcTap taps[32];   // ot/oh/ow in [-1,1]; src_off is rewritten to the row shift inside the LDS image
};
struct VcHaloPlan {
    int NW, TT, SEG, splits, CK;  // CK: channels per K chunk (64, or 32 for two workgroups per CU)
    int pad;                      // zero-padded LDS image (no per-tap validity selects)
    int NB;                       // 16-position blocks per wave (4; 7 or 8 in the two-workgroups-per-CU forms)
};
int vc_halo_plan(int NP, int T, int H, int W, int Cs, int Cn, int ntaps, VcHaloPlan* plan);
int vc_launch_halo(VcHaloParams& p, const VcHaloPlan& plan, hipStream_t stream);

// wgrad9_conv.hip: 3x3x3 stride-1 weight gradient, nine (dh,dw) taps per LDS image
struct VcWgrad9Params {
    const bf16* x;   // [NP*H*W][Cs]
    const bf16* dy;  // [NP*H*W][Co]
    float* dw;       // [Co][27*Cs] fp32
    float* partial;  // nullable: fp32 slabs [splits][Co][27*Cs]
    long long partial_stride;
    int NP, T, H, W, Cs, Co, Kw;
    int rows_per_split, splits;
    int dt0, ndt;    // temporal offsets computed: dt = dt0 - 1 .. dt0 + ndt - 2 (all three, or only dt = 0 when T == 1)
    int xcd_map;     // set by the launcher: workgroup id -> (slice, co, dt/ci) so that a slice's workgroups share an XCD
    long long* dbg;  // nullable: per-workgroup phase clocks (tuning aid)
    VcFastDiv dHW, dT, dW;
};
int vc_wgrad9_plan(long M, int W, int Cs, int Co, int ndt, int* rows_per_split);  // splits; 0 = geometry not supported
int vc_launch_wgrad9(VcWgrad9Params& p, hipStream_t stream);

// stem_wgrad.hip: stem weight gradient with the raw space-to-depth rows resident in LDS (persistent workgroups)
struct VcStemWgradParams {
    const bf16* xs;  // [N][Tp][Hp][Wp][12], Wp even
    const bf16* dy;  // [N*T*Ho*Wo][64]
    float* slabs;    // [grid][64][1344] fp32 partials
    int N, T, Ho, Wo, Tp, Hp, Wp;
    int RH, KS;      // output rows per chunk, 32-row k-steps per chunk
    int TR, nhg, ntr, units;  // frames per unit, row groups, frame ranges, units = N * nhg * ntr
    int plane_pieces, grid, lds_bytes;
    int xcd_map;
};
int vc_stem_wgrad_plan(int N, int T, int Ho, int Wo, int Hp, int Wp, int Co, VcStemWgradParams* out);  // 0 = not applicable
size_t vc_stem_wgrad_slab_bytes(const VcStemWgradParams& p);
int vc_launch_stem_wgrad(VcStemWgradParams& p, float* dws, hipStream_t stream);

// stem_fwd.hip: stem forward with the raw space-to-depth rows resident in LDS (persistent workgroups)
struct VcStemFwdParams {
    const bf16* xs;  // [N][Tp][Hp][Wp][12], Wp even
    const bf16* ws;  // [64][1344]
    bf16* y;         // [N*T*Ho*Wo][64]
    float* stats;    // nullable: [VC_STAT_REPLICAS][2][64]
    const float* stat_shift;  // nullable: [64], subtracted from every value before it enters the sums (bn_fin.h)
    VcBnFin fin;     // fin.ticket nullable: the last workgroup finalizes the BatchNorm parameters (bn_fin.h)
    int N, T, Ho, Wo, Tp, Hp, Wp;
    int RH;          // output rows per chunk (RH * Wo <= 224)
    int TR, nhg, ntr, units;
    int plane_pieces, grid, lds_bytes;
    int xcd_map;     // XCD k takes the k-th contiguous eighth of each round of units
};
int vc_stem_fwd_plan(int N, int T, int Ho, int Wo, int Hp, int Wp, int Co, VcStemFwdParams* out);  // 0 = not applicable
int vc_launch_stem_fwd(VcStemFwdParams& p, hipStream_t stream);

int vc_launch_igemm(const VcIgemmParams& p, int mode, hipStream_t stream);
int vc_launch_wgrad(const VcWgradParams& p, int mode, hipStream_t stream);
// split-K planning (shared by the workspace query and the launch)
int vc_igemm_splits(int M, int Cn, int nchunks);
int vc_wgrad_splits(int M, int Co, int nblk, int* rows_per_split);
