// Internal launch interfaces between capi.hip and the kernel translation units.
#pragma once
#include "common.h"

struct VcIgemmParams {
    const bf16* src;  // [N][Ts][Hs][Ws][Cs]
    const bf16* wgt;  // [Cn][Kw]
    bf16* out;
    float* stats;  // nullable: [VC_STAT_REPLICAS][2][Cn]
    int M, Ta, Ha, Wa;
    VcFastDiv dWa, dHa, dTa;
    int Ts, Hs, Ws, Cs;
    int st, sh, sw;
    int ntaps, Kw, Cn;
    int out_n, out_t, out_h, out_w, out_base;
    int accumulate;
    // split-K: blockIdx.z handles chunks [z*cps, (z+1)*cps); partial sums go to fp32 slabs
    // partial[z][same element offsets as out]; a finalize kernel sums, rounds and takes BN statistics
    int splits, cps;
    float* partial;
    long long partial_stride;
    VcTap taps[32];
};

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
};

int vc_launch_igemm(const VcIgemmParams& p, int mode, hipStream_t stream);
int vc_launch_wgrad(const VcWgradParams& p, int mode, hipStream_t stream);
// split-K planning (shared by the workspace query and the launch)
int vc_igemm_splits(int M, int Cn, int nchunks);
int vc_wgrad_splits(int M, int Co, int nblk, int* rows_per_split);
