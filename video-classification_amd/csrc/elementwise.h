// Launch wrappers of elementwise.hip (internal; the C ABI in capi.hip validates and forwards).
#pragma once
#include "common.h"
#include "bn_fin.h"

struct VcPoolGeom {
    int N, C, T, H, W;
    int To, Ho, Wo;
    int kt, kh, kw, st, sh, sw, pt, ph, pw;
};

int vc_ew_bn_finalize(const float* stats, int C, float count, const float* gamma, const float* beta, float* rmean,
                      float* rvar, float momentum, float eps, float* params, const float* shift, float* shift_out,
                      hipStream_t s);
// lstm.hip
int vc_ew_lstm_pointwise_fwd(const void* gates, const float* b_ih, const float* b_hh, const float* c_prev, float* h, float* c,
                             void* h_bf16, float* saved, int N, int H, hipStream_t s);
int vc_ew_lstm_pointwise_bwd(const float* dh, const float* dc_next, const float* saved, const float* c_prev, void* dgates,
                             float* dc_prev, float* db_ih, float* db_hh, int N, int H, hipStream_t s);
int vc_ew_bn_eval_params(int C, const float* gamma, const float* beta, const float* rmean, const float* rvar, float eps,
                         float* params, hipStream_t s);
// bits (nullable; needs relu): one byte per 8 channels, bit k = (out[k] > 0)
int vc_ew_bn_apply(const void* y, const float* params, const void* res, void* out, long n, int C, int relu,
                   uint8_t* bits, hipStream_t s);
// shift_out (nullable): the sums are taken relative to row 0 of y, which is written there for the finalize
int vc_ew_bn_stats(const void* y, float* sums, long rows, int C, float* shift_out, hipStream_t s);
int vc_ew_bn_bwd_reduce(const void* y, const void* dz, const void* aux, const float* params, float* sums, long rows,
                        int C, int mask, const VcBnBwdFin* fin, hipStream_t s);
int vc_ew_bn_bwd_finalize(const float* sums, int C, float count, float* coef, float* dgamma, float* dbeta,
                          hipStream_t s);
int vc_ew_bn_bwd_apply(const void* y, const void* dz, const void* aux, const float* params, const float* coef, void* dy,
                       void* gout, long n, int C, int mask, hipStream_t s);
int vc_ew_splitk_finalize(const float* slabs, int splits, long long stride, void* out, int accumulate, float* stats,
                          long rows, int C, const VcBnFin* fin, hipStream_t s);
int vc_ew_splitk_finalize_red(const float* slabs, int splits, long long stride, void* out, int accumulate, long rows, int C,
                              const void* y, const float* params, const uint8_t* bits, float* sums,
                              const VcBnBwdFin& fin, hipStream_t s);
// live: bit t set = tap t of the [Co][taps][Ci] gradient was computed (entries of dropped taps are skipped); taps = 0: all
int vc_ew_splitk_reduce_wgrad(const float* slabs, int splits, long long stride, float* dw, long n, int Ci, int taps,
                              unsigned live, hipStream_t s);
int vc_ew_add(const void* a, const void* b, void* out, long n, hipStream_t s);
int vc_ew_mask_bits(void* x, const uint8_t* bits, long n, hipStream_t s);
int vc_ew_maxpool_fwd(const void* y, const float* params, void* out, uint8_t* idx, const VcPoolGeom& g, hipStream_t s);
int vc_ew_maxpool_bwd(const void* dout, const uint8_t* idx, void* din, const VcPoolGeom& g, hipStream_t s);
int vc_ew_avgpool_fwd(const void* in, float* out, int N, int S, int C, hipStream_t s);
int vc_ew_avgpool_bwd(const float* dfeat, void* din, int N, int S, int C, hipStream_t s);
int vc_ew_softmax_ce(const float* logits, const long long* y, float* loss, float* dlogits, int N, int K, hipStream_t s);
int vc_ew_linear_fwd(const float* feat, const float* W, const float* bias, float* out, int N, int F, int K,
                     hipStream_t s);
int vc_ew_linear_bwd(const float* dlog, const float* feat, const float* W, float* dfeat, float* dW, float* db, int N,
                     int F, int K, hipStream_t s);
int vc_ew_stem_pack_input(const void* x, int x_is_bf16, void* xs, int N, int T, int H, int W, hipStream_t s);
int vc_ew_clip_pack_u8(const uint8_t* frames, const int* crops, void* xs, int N, int T, int H, int W, int Ts, int Hs,
                       int Ws, int tstep, const float* a, const float* b, hipStream_t s);
int vc_ew_ncs_to_nsc(const void* in, int in_is_bf16, void* out, int N, int C, long S, hipStream_t s);
int vc_ew_nsc_to_ncs(const void* in, void* out, int out_is_bf16, int N, int C, long S, hipStream_t s);
int vc_ew_pack_weight(const float* w, void* wf, void* wd, int Co, int taps, int Ci, hipStream_t s);
int vc_ew_stem_pack_weight(const float* w, void* ws, int Co, hipStream_t s);
int vc_ew_stem_unpack_wgrad(const float* dws, float* dw, int Co, hipStream_t s);
int vc_ew_sgd_step(float* p, const float* g, float* mom, void* shadow, long n, float lr, float momentum, float wd,
                   int nesterov, hipStream_t s);
int vc_ew_adam_step(float* p, const float* g, float* m, float* v, void* shadow, long n, float step_size, float b1, float b2,
                    float eps, float wd, float bias2_sqrt, hipStream_t s);
int vc_ew_shadow(const float* p, void* shadow, long n, hipStream_t s);
int vc_ew_pack_dgrad_batched(const void* shadow, void* wd, const int* table, int nconv, int tiles, hipStream_t s);
int vc_ew_bn_bwd_reduce_pooled(const void* y, const void* dout, const uint8_t* idx, const void* pooled, const float* params,
                               float* sums, const VcPoolGeom& g, const VcBnBwdFin* fin, hipStream_t s);
int vc_ew_bn_bwd_apply_pooled(const void* y, const void* dout, const uint8_t* idx, const float* params, const float* coef,
                              void* dy, const VcPoolGeom& g, hipStream_t s);
