/* vc3d.h -- C ABI of libvc3d.so: the MI355X (gfx950) kernels of the 3D-ResNet hot path.
 *
 * The reference (bfshi/Video-Classification) has no FFI: its hot path is the torch.nn
 * calls inside lib/models/backbone.py.  Each entry point below replaces one of those
 * calls (3D-inflated) and cites it.  A binding only needs ctypes/cffi:
 * plain pointers, ints and floats; no torch types cross this boundary.
 *
 * Conventions
 *  - every function returns 0 on success, a negative code on error; vc_last_error()
 *    returns the thread-local message.  Nothing throws across the boundary.
 *  - all pointers are DEVICE pointers owned by the caller.  The library allocates
 *    nothing, holds no state, never synchronises and launches on `stream`
 *    (a hipStream_t passed as void*; NULL = the default stream).
 *  - activations are channels-last bf16: [N][T][H][W][C]; C is a multiple of 64
 *    (the stem's 3-channel clip goes through vc_stem_pack_input instead).
 *  - conv weights (bf16) are [Cout][kt][kh][kw][Cin] for forward / wgrad and
 *    [Cin][kt][kh][kw][Cout] for dgrad; vc_pack_weight makes both from the fp32 master
 *    (which is torch's channels_last_3d storage of a [Cout][Cin][kt][kh][kw] parameter).
 *  - "stats"/"sums" buffers are 256 * C bytes (float [VC_STAT_REPLICAS][2][C]), 256-byte aligned, zeroed by the
 *    caller before the kernel that accumulates into them; their content is the library's (float replicas by
 *    default, 64-bit integer limbs in deterministic mode: vc_set_deterministic) and is read back only by the
 *    library's finalize steps.
 *  - "bn params" are float [4][C]: mean, invstd, scale = gamma*invstd, shift = beta-mean*scale.
 */
#ifndef VC3D_H
#define VC3D_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VC_STAT_REPLICAS 32
#define VC_ABI_VERSION 2

typedef struct vc_conv3d_desc {
    int32_t N, Ci, Ti, Hi, Wi; /* input  [N][Ti][Hi][Wi][Ci] */
    int32_t Co;                /* output channels            */
    int32_t kt, kh, kw;        /* kernel                     */
    int32_t st, sh, sw;        /* stride                     */
    int32_t pt, ph, pw;        /* zero padding               */
} vc_conv3d_desc;

typedef struct vc_pool3d_desc {
    int32_t N, C, T, H, W;
    int32_t kt, kh, kw, st, sh, sw, pt, ph, pw;
} vc_pool3d_desc;

const char* vc_last_error(void);
int vc_abi_version(void);

/* Deterministic-reduction mode (the reference has no counterpart -- cuDNN's BatchNorm sums are
 * order-dependent too).  Process-wide; default off, or on when VC_DETERMINISTIC=1 is in the environment at the first
 * launch.  On: every BatchNorm sum (forward statistics in the conv epilogues, split-K finalize, stand-alone and folded
 * backward reductions) is accumulated as exact 64-bit fixed-point integers instead of float atomics, so results do not
 * depend on the order in which workgroups finish: two runs of the same step are bit-identical.  Set it before the first
 * step and do not flip it between a kernel that accumulates into a buffer and the finalize that reads it.  Exact for
 * partial sums of magnitude [2^-21, 2^36); a non-finite or larger partial sum makes the total NaN. */
int vc_set_deterministic(int32_t on);
int vc_get_deterministic(void);

/* ---- Conv3d (bf16 MFMA implicit GEMM).  Replaces nn.Conv2d at
 * reference lib/models/backbone.py:19-20 (conv3x3), :70-76 (Bottleneck 1x1/3x3/1x1),
 * :131-132 (downsample 1x1) -- forward, and autograd's dgrad/wgrad of the same calls
 * (reference lib/core/function.py:81 loss.backward()). kt*kh*kw <= 32. */
int vc_conv3d_out_dims(const vc_conv3d_desc* d, int32_t* To, int32_t* Ho, int32_t* Wo);
/* `workspace` (device, caller-owned, nullable) holds the fp32 partial slabs of the split-K form used
 * when the plain grid would not fill the chip; vc_conv3d_workspace_bytes(d, op) reports the need
 * (op: 0 forward, 1 dgrad, 2 wgrad; 0 bytes = never used).  Without it the unsplit kernel runs
 * (wgrad: float atomics).  With it results are bitwise reproducible. */
size_t vc_conv3d_workspace_bytes(const vc_conv3d_desc* d, int32_t op);
int vc_conv3d_fwd(const vc_conv3d_desc* d, const void* x, const void* w, void* y,
                  float* bn_stats /* nullable: fused sum / sum-of-squares of y */, void* workspace,
                  size_t ws_bytes, void* stream);
/* BatchNorm finalize folded into the launch that produces the sums (reference backbone.py:39-42,114: the
 * BatchNorm that follows every conv).  The workgroup that finishes last does vc_bn_finalize's work -- same
 * arithmetic, one launch less per layer.  `ticket`: VC_TICKET_WORDS (64) uint32 words in device memory, zero before their
 * first use; the launch leaves them zero, so one zero-initialised block per layer is reused step after step.
 * Launches that share a ticket block must not overlap.  `count` = elements per channel (N*T*H*W of y). */
#define VC_TICKET_WORDS 64
typedef struct vc_bn_fin {
    uint32_t* ticket;
    const float* gamma;
    const float* beta;
    float* running_mean; /* nullable, with running_var */
    float* running_var;
    float* params;       /* out [4][C]: mean, invstd, scale, shift (as vc_bn_finalize) */
    float count, momentum, eps;
    /* Shifted sums (numerics): with `shift` ([C], nullable) the launch accumulates sum (y - shift) and sum (y - shift)^2,
     * and the finalize adds the shift back: var = E[(y-s)^2] - E[y-s]^2 does not cancel when |mean| >> sigma as long as
     * s is within a few sigma of the mean.  `shift_next` ([C], nullable, may alias shift) receives this batch's mean:
     * pass the same buffer step after step and the shift tracks the mean (first step: zeros, or vc_bn_stats' row-0
     * shift for a robust first estimate). */
    const float* shift;
    float* shift_next;
} vc_bn_fin;
int vc_conv3d_fwd_bn(const vc_conv3d_desc* d, const void* x, const void* w, void* y, float* bn_stats,
                     const vc_bn_fin* fin, void* workspace, size_t ws_bytes, void* stream);
/* Inference form (reference function.py:117-134, model.eval() under no_grad): BatchNorm with running statistics is
 * a per-channel scale and shift, so it is applied with the ReLU in the conv's epilogue on the fp32 accumulators:
 * y = relu(scale * conv(x) + shift [+ residual]), bn_params = [4][Co] as vc_bn_eval_params writes them, residual
 * (nullable, laid out like y, not y itself) = the block input of `out += residual` (backbone.py:59-60,100-101).
 * No separate vc_bn_apply pass.  (Split-K geometries apply it in place after the conv.) */
int vc_conv3d_fwd_act(const vc_conv3d_desc* d, const void* x, const void* w, void* y, const float* bn_params,
                      const void* residual, int32_t relu, void* workspace, size_t ws_bytes, void* stream);
/* No counterpart in the reference: vc_conv3d_fwd for callers that cannot allocate (code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc) and so have no workspace to bring.  The split-K partial
 * slabs live in a file-scope device array of VC_STATIC_SCRATCH_BYTES that the library holds on every GPU it runs on
 * until the process ends; the call itself allocates nothing.  Same kernels, same split plan and therefore the same
 * bits in y as vc_conv3d_fwd with a workspace of vc_conv3d_workspace_bytes(d, 0).  A geometry whose plan needs more
 * than VC_STATIC_SCRATCH_BYTES is rejected (-3, message in vc_last_error) -- it is never silently run unsplit.
 * All calls on one GPU share that GPU's array: calls that overlap on one GPU (two streams, two threads) race;
 * serialise them or use vc_conv3d_fwd.  Nothing is kept between calls (the array needs no clearing).
 * The array's device address is looked up once per device, on the first call there; vc_static_scratch_bytes() with
 * the device current does that lookup ahead of time -- call it before a stream capture begins. */
#define VC_STATIC_SCRATCH_BYTES (64u << 20)
size_t vc_static_scratch_bytes(void);
int vc_conv3d_fwd_noalloc(const vc_conv3d_desc* d, const void* x, const void* w, void* y,
                          float* bn_stats /* nullable */, void* stream);
int vc_conv3d_dgrad(const vc_conv3d_desc* d, const void* dy, const void* w_dgrad, void* dx,
                    int accumulate /* dx += instead of dx = */, void* workspace, size_t ws_bytes,
                    void* stream);
/* vc_conv3d_dgrad with the two elementwise neighbours of a residual block's data gradient folded in (both optional):
 *
 * red -- the reduction of the BatchNorm backward that consumes dx.  When dx is the gradient that reaches `relu(bn(y))`
 * (backbone.py:50-55,88-95) or `relu(bn(y) + residual)` (backbone.py:56-61,97-102: the block's first conv hands it to
 * the previous block), the next thing the backward does is vc_bn_bwd_reduce_fin(y, dx, relu_bits, params, sums, rows,
 * C = Ci, mask, ticket, coef, dgamma, dbeta) -- a pass that reads dx back.  This entry point produces the same sums,
 * coef, dgamma += and dbeta += with the conv.  relu_bits given: mask mode 3; null: mask mode 2 (the mask is recomputed
 * from y).  sums must be zero on entry; with accumulate the sums are over the accumulated dx.
 *
 * acc_relu_bits (with accumulate = 1) -- dx holds the gradient g that reached the block's `relu(... + residual)`, and
 * what has to be added to the conv's data gradient is the part of it that passes that ReLU into the identity shortcut
 * (backbone.py:59-61,100-102): dx = (dx where its bit of acc_relu_bits is set, else 0) + dgrad.  The masked copy of g
 * that vc_bn_bwd_apply would write (g_out) and this call would read back is never made.
 *
 * Dense geometries (1x1x1, unit stride, unsplit) take both in the GEMM's own launch -- the mask where the old dx tile
 * is staged, the sums from the finished tile in the epilogue; any other geometry runs them as separate launches
 * inside this call (same results). */
typedef struct {
    const void* y;            /* what the BatchNorm normalised: [N*Ti*Hi*Wi][Ci] bf16 */
    const float* params;      /* [4][Ci] of that BatchNorm */
    const uint8_t* relu_bits; /* vc_bn_apply_mask's bits of the BatchNorm's output, or null (mask mode 2) */
    float* sums;              /* [32][2][Ci], zero on entry */
    uint32_t* ticket;         /* as in vc_bn_fin */
    float* coef;              /* [2][Ci] */
    float* dgamma;            /* += nullable */
    float* dbeta;             /* += nullable */
} vc_bn_bwd_red;
int vc_conv3d_dgrad_fused(const vc_conv3d_desc* d, const void* dy, const void* w_dgrad, void* dx, int accumulate,
                          const uint8_t* acc_relu_bits /* nullable */, const vc_bn_bwd_red* red /* nullable */,
                          void* workspace, size_t ws_bytes, void* stream);
int vc_conv3d_wgrad(const vc_conv3d_desc* d, const void* x, const void* dy,
                    float* dw /* fp32 [Co][kt][kh][kw][Ci], += */, void* workspace, size_t ws_bytes,
                    void* stream);
int vc_pack_weight(const float* w_master, void* w_fwd, void* w_dgrad /* nullable */, int32_t Co,
                   int32_t taps, int32_t Ci, void* stream);

/* ---- Optimizer step and weight repack over FLAT buffers (all parameters of the model in one fp32
 * array, gradients in another of the same layout).  Replaces optimizer.step() at reference
 * lib/core/function.py:82 for the SGD that lib/utils/utils.py:22-29 builds (torch.optim.SGD arithmetic,
 * dampening 0): one launch for the whole model, which also writes the bf16 shadow of the new
 * parameters -- a conv weight's slice of the shadow is its forward / wgrad operand.
 * vc_pack_dgrad_batched then makes every conv's dgrad operand in one launch; the table (the same
 * int32 [nconv][6] on the host and on the device) holds per conv: element offset in the shadow,
 * element offset in w_dgrad_flat, Co, taps, Ci, index of its first 64x64 tile. */
int vc_sgd_step(float* params, const float* grads, float* momentum_buf /* nullable */,
                void* shadow_bf16 /* nullable */, int64_t n, float lr, float momentum, float weight_decay,
                int32_t nesterov, void* stream);
/* torch.optim.Adam's step (the reference's DEFAULT optimizer: config.py:43, utils.py:30-33; betas / eps as passed,
 * amsgrad off) over the same flat buffers, one launch, also writing the bf16 shadow.  `step` = 1 for the first call
 * (bias corrections 1 - beta^step are taken on the host in double, as torch does); exp_avg / exp_avg_sq start at zero. */
int vc_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, void* shadow_bf16 /* nullable */,
                 int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay, int64_t step,
                 void* stream);
int vc_shadow_bf16(const float* params, void* shadow_bf16, int64_t n, void* stream);
int vc_pack_dgrad_batched(const void* shadow_bf16, void* w_dgrad_flat, const int32_t* table_dev,
                          const int32_t* table_host, int32_t nconv, void* stream);

/* ---- Stem: 7x7x7, 3 -> Co (multiple of 64), stride (1,2,2), pad 3.  Replaces
 * reference lib/models/backbone.py:112-113,145.  The clip [N][3][T][H][W] (fp32 or bf16,
 * H and W even) is packed once into a zero-padded space-to-depth bf16 source
 * [N][T+6][H/2+3][Wp][12], Wp = W/2+3 rounded up to even; the weight into [Co][1344]. The stem has no dgrad. */
size_t vc_stem_packed_input_elems(int32_t N, int32_t T, int32_t H, int32_t W);
int vc_stem_pack_input(const void* x, int32_t x_is_bf16, void* xs, int32_t N, int32_t T, int32_t H,
                       int32_t W, void* stream);
/* Clip input pipeline (the reference's get_dataset, lib/dataset/dataset.py:6-12, is empty and
 * its loop takes ready clips at lib/core/function.py:58,68): decoded uint8 RGB frames [N][Ts][Hs][Ws][3] ->
 * temporal sampling (frame t0 + t*tstep), spatial crop at (y0, x0), optional horizontal flip, per-channel
 * normalisation v = u8 * (1 / (255 std[c])) - mean[c] / std[c] (one fp32 fma), rounded to bf16 and written
 * straight into the packed stem source of vc_stem_pack_input (same layout, same padding) in one pass.
 * crops: int32 [N][4] = {t0, y0, x0, flip}, the same table on the device and on the host (validated on the host:
 * the kernel does no bounds tests); mean/std: 3 host floats each, in units of u8/255. */
int vc_clip_pack_u8(const uint8_t* frames, const int32_t* crops_dev, const int32_t* crops_host, void* xs,
                    int32_t N, int32_t T, int32_t H, int32_t W, int32_t Ts, int32_t Hs, int32_t Ws,
                    int32_t tstep, const float* mean, const float* std, void* stream);
int vc_stem_pack_weight(const float* w_master /* [Co][7][7][7][3] */, void* ws, int32_t Co, void* stream);
int vc_stem_fwd(const void* xs, const void* ws, void* y /* [N][T][H/2][W/2][Co] */, float* bn_stats,
                int32_t N, int32_t T, int32_t H, int32_t W, int32_t Co, void* stream);
int vc_stem_fwd_bn(const void* xs, const void* ws, void* y, float* bn_stats, const vc_bn_fin* fin, int32_t N,
                   int32_t T, int32_t H, int32_t W, int32_t Co, void* stream);
/* `workspace` (device, caller-owned, nullable): per-workgroup fp32 partials of the image-resident stem weight
 * gradient; vc_stem_wgrad_workspace_bytes reports the need (0: the generic kernel runs, as it does when the
 * workspace is absent or too small). */
size_t vc_stem_wgrad_workspace_bytes(int32_t N, int32_t T, int32_t H, int32_t W, int32_t Co);
int vc_stem_wgrad(const void* xs, const void* dy, float* dws /* fp32 [Co][1344], += */, int32_t N,
                  int32_t T, int32_t H, int32_t W, int32_t Co, void* workspace, size_t ws_bytes, void* stream);
int vc_stem_unpack_wgrad(const float* dws, float* dw_master /* [Co][7][7][7][3], += */, int32_t Co,
                         void* stream);

/* ---- BatchNorm3d (+ReLU, +residual).  Replaces nn.BatchNorm2d / nn.ReLU / `out += residual`
 * at reference lib/models/backbone.py:39-42,50-51,54,59-60 (BasicBlock), :71-79,87-101
 * (Bottleneck), :114-115,146-147 (stem), :133 (downsample); BN_MOMENTUM :13. */
/* shift_out (nullable, [C]): the sums are taken relative to row 0 of y (a value within a few sigma of the channel
 * mean, known to every workgroup without a pass of its own) and that row is written to shift_out for vc_bn_finalize. */
int vc_bn_stats(const void* y, float* sums, float* shift_out, int64_t rows, int32_t C, void* stream);
/* shift (nullable): what the sums were taken relative to; shift_next (nullable, may alias): receives the mean */
int vc_bn_finalize(const float* stats, int32_t C, float count, const float* gamma, const float* beta,
                   float* running_mean /* nullable */, float* running_var, float momentum, float eps,
                   float* params, const float* shift, float* shift_next, void* stream);
int vc_bn_eval_params(int32_t C, const float* gamma, const float* beta, const float* running_mean,
                      const float* running_var, float eps, float* params, void* stream);
int vc_bn_apply(const void* y, const float* params, const void* residual /* nullable */, void* out,
                int64_t elems, int32_t C, int32_t relu, void* stream);
/* vc_bn_apply with relu = 1 that also writes the ReLU mask of its output, one bit per element (relu_bits: elems / 8
 * bytes; bit k of byte i = out[8 i + k] > 0): the backward of `out += residual; relu` (backbone.py:59-61,100-102)
 * then reads 1 bit per element instead of the 16-bit activation (mask mode 3 below). */
int vc_bn_apply_mask(const void* y, const float* params, const void* residual /* nullable */, void* out,
                     uint8_t* relu_bits, int64_t elems, int32_t C, void* stream);
/* mask: 0 none; 1 gradient passes where aux (saved post-activation output) > 0;
 *       2 where y*scale+shift > 0 (recomputed); 3 where the bit of aux = vc_bn_apply_mask's relu_bits is set */
int vc_bn_bwd_reduce(const void* y, const void* dz, const void* aux, const float* params, float* sums,
                     int64_t rows, int32_t C, int32_t mask, void* stream);
int vc_bn_bwd_finalize(const float* sums, int32_t C, float count, float* coef /* [2][C] */,
                       float* dgamma /* += nullable */, float* dbeta /* += nullable */, void* stream);
/* vc_bn_bwd_reduce with vc_bn_bwd_finalize folded in (the last workgroup does it; `ticket` as in vc_bn_fin;
 * count = rows). */
int vc_bn_bwd_reduce_fin(const void* y, const void* dz, const void* aux, const float* params, float* sums,
                         int64_t rows, int32_t C, int32_t mask, uint32_t* ticket, float* coef /* [2][C] */,
                         float* dgamma /* += nullable */, float* dbeta /* += nullable */, void* stream);
int vc_bn_bwd_apply(const void* y, const void* dz, const void* aux, const float* params,
                    const float* coef, void* dy, void* g_out /* nullable: masked dz */, int64_t elems,
                    int32_t C, int32_t mask, void* stream);
int vc_add(const void* a, const void* b, void* out, int64_t elems, void* stream);

/* ---- Pooling.  Replaces nn.MaxPool2d(3,2,1) at reference lib/models/backbone.py:116,148 and
 * nn.AvgPool2d(7)+Flatten at :122-125,155 (global average in 3D). */
int vc_pool3d_out_dims(const vc_pool3d_desc* d, int32_t* To, int32_t* Ho, int32_t* Wo);
int vc_maxpool3d_fwd(const vc_pool3d_desc* d, const void* y,
                     const float* bn_params /* nullable: apply BN+ReLU to y first */, void* out,
                     uint8_t* argmax, void* stream);
int vc_maxpool3d_bwd(const vc_pool3d_desc* d, const void* dout, const uint8_t* argmax, void* din,
                     void* stream);
/* BatchNorm+ReLU backward THROUGH the max pool (the stem, reference backbone.py:146-148 in reverse): the
 * gradient entering BN+ReLU is gathered from the pooled gradient and the arg-max map on the fly instead of
 * being materialised by vc_maxpool3d_bwd.  `d` describes the pool (its input = y's geometry); ReLU mask
 * recomputed from y (mask mode 2).  Use with vc_bn_bwd_finalize in between.  `pooled` (nullable) is the
 * pool's own output from vc_maxpool3d_fwd with bn_params: the reduce then solves y at the arg-max from it
 * instead of gathering y element by element (channels with |beta| > 16 |gamma| keep the gather). */
int vc_bn_bwd_reduce_pooled(const vc_pool3d_desc* d, const void* y, const void* dpooled, const uint8_t* argmax,
                            const void* pooled, const float* params, float* sums, void* stream);
int vc_bn_bwd_reduce_pooled_fin(const vc_pool3d_desc* d, const void* y, const void* dpooled, const uint8_t* argmax,
                                const void* pooled, const float* params, float* sums, uint32_t* ticket,
                                float* coef, float* dgamma, float* dbeta, void* stream);
int vc_bn_bwd_apply_pooled(const vc_pool3d_desc* d, const void* y, const void* dpooled, const uint8_t* argmax,
                           const float* params, const float* coef, void* dy, void* stream);
int vc_gavgpool_fwd(const void* in, float* out, int32_t N, int32_t S, int32_t C, void* stream);
int vc_gavgpool_bwd(const float* dfeat, void* din, int32_t N, int32_t S, int32_t C, void* stream);

/* ---- Classifier head (fp32).  Replaces nn.Linear at reference lib/models/model.py:24,38. */
int vc_linear_fwd(const float* feat, const float* W, const float* bias, float* out, int32_t N,
                  int32_t F, int32_t K, void* stream);
int vc_linear_bwd(const float* dlogits, const float* feat, const float* W, float* dfeat /* nullable */,
                  float* dW /* += nullable */, float* db /* += nullable */, int32_t N, int32_t F,
                  int32_t K, void* stream);

/* ---- Cross-entropy of the class scores (fp32), the first term of the reference's Loss (lib/core/loss.py:10,25:
 * nn.CrossEntropyLoss, mean reduction): loss[0] = mean_n (logsumexp(logits[n]) - logits[n][labels[n]]) and, when dlogits
 * is given, d loss / d logits = (softmax - onehot) / N, in one launch instead of torch's chain of log-softmax / NLL /
 * fill / NLL-backward / softmax-backward launches between the forward and the backward.  K <= 1024 (a row lives in
 * registers; more is refused).  A label outside [0, K) makes the loss NaN (torch raises; that needs the labels on the host). */
int vc_softmax_ce(const float* logits, const int64_t* labels, float* loss, float* dlogits /* nullable */, int32_t N,
                  int32_t K, void* stream);

/* ---- LSTM cell.  Replaces nn.LSTMCell at reference lib/models/lstm.py:16,25-27.  The cell's two
 * GEMMs, gates[N][4H] = [x h] [W_ih W_hh]^T, are a 1x1x1 conv over N positions (vc_conv3d_fwd / _dgrad / _wgrad with
 * T = H = W = 1, Ci = Fin + H, Co = 4H; bf16 MFMA).  These two calls are the element-wise rest, one pass each way.
 * gates / dgates: bf16 [N][4H], torch's gate order i, f, g, o.  saved: fp32 [N][5H] = the four activations and
 * tanh(c').  h_bf16 (nullable): h' once more in bf16, the next step's GEMM operand.  Backward: dh and dc_next are the
 * gradients of h' and c' (either nullable); db_ih / db_hh (nullable) are accumulated into (+=). */
int vc_lstm_pointwise_fwd(const void* gates, const float* b_ih, const float* b_hh, const float* c_prev, float* h, float* c,
                          void* h_bf16, float* saved, int32_t N, int32_t H, void* stream);
int vc_lstm_pointwise_bwd(const float* dh, const float* dc_next, const float* saved, const float* c_prev, void* dgates,
                          float* dc_prev, float* db_ih, float* db_hh, int32_t N, int32_t H, void* stream);

/* ---- Layout helpers at the API edge (NCTHW <-> NTHWC). */
int vc_ncthw_to_nthwc(const void* in, int32_t in_is_bf16, void* out, int32_t N, int32_t C, int64_t S,
                      void* stream);
int vc_nthwc_to_ncthw(const void* in, void* out, int32_t out_is_bf16, int32_t N, int32_t C, int64_t S,
                      void* stream);

/* ---- Command lists: a recorded launch sequence replayed by one call (no counterpart in the reference, whose step is
 * enqueued layer by layer from Python: lib/core/function.py:71-82).  A command is one call of an entry point of this
 * library (any one that returns int and takes only integers, pointers and at most VC_CMD_MAX_FLT floats: `fn`, its
 * integer / pointer arguments in order in iarg, its float arguments in order in farg; descriptor structs are passed by
 * pointer and must stay alive and unchanged), or a cross-stream dependency: VC_CMD_EVENT_RECORD (iarg[0] = index into
 * `events`, iarg[1] = stream) / VC_CMD_STREAM_WAIT (same operands: the stream waits for the event).  vc_replay issues
 * the commands in order, exactly as the recorded calls were issued, and stops at the first failure (*failed_at = its
 * index, the entry point's message in vc_last_error).  time_cmd >= 0: HIP events time_start / time_stop (created with
 * timing) are recorded on that command's stream around it.  The caller guarantees what made the recording valid:
 * every pointer in the list still names the buffer it named.  x86-64 System V only (replay.hip). */
#define VC_CMD_MAX_INT 16
#define VC_CMD_MAX_FLT 8
#define VC_CMD_CALL 0
#define VC_CMD_EVENT_RECORD 1
#define VC_CMD_STREAM_WAIT 2
typedef struct vc_cmd {
    void* fn;
    int32_t kind, nint, nflt, pad_;
    int64_t iarg[VC_CMD_MAX_INT];
    float farg[VC_CMD_MAX_FLT];
} vc_cmd;
int vc_replay(const vc_cmd* cmds, int32_t n, void* const* events, int32_t n_events, int32_t time_cmd /* -1: none */,
              void* time_start, void* time_stop, int32_t* failed_at /* nullable */);
int vc_replay_events_create(int32_t n, int32_t timing, void** out);
int vc_replay_events_destroy(int32_t n, void** events);
int vc_replay_event_elapsed_ms(void* start, void* stop, float* ms);
/* small helpers so that a whole step is expressible as commands */
int vc_memset_async(void* ptr, int32_t value, size_t bytes, void* stream);
int vc_incr_i64(const void* ptr_table /* device: int64* [n] */, int32_t n, void* stream); /* *ptr[i] += 1 (BatchNorm's num_batches_tracked) */

#ifdef __cplusplus
}
#endif
#endif /* VC3D_H */
