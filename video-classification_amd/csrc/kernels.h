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
    VcTap taps[32];
};

int vc_launch_igemm(const VcIgemmParams& p, int mode, hipStream_t stream);
int vc_launch_wgrad(const VcWgradParams& p, int mode, hipStream_t stream);
