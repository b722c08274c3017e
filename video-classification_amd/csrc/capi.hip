// extern "C" entry points of libvc3d.so (declared in include/vc3d.h): validate shapes on the
// host, build the gather descriptions and launch.  Nothing here allocates or synchronises.
#include <stdarg.h>
#include <atomic>
#include <cmath>
#include <mutex>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/vc3d.h"
#include "common.h"
#include "elementwise.h"
#include "kernels.h"

static thread_local char g_err[512] = "";

void vc_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- deterministic BatchNorm sums (common.h: vc_stat_add): a process-wide mode
static std::atomic<int> g_det_mode{-1};  // -1: not decided yet (VC_DETERMINISTIC read at first use)
bool vc_det_mode() {
    int m = g_det_mode.load(std::memory_order_relaxed);
    if (m < 0) {
        const char* e = getenv("VC_DETERMINISTIC");
        m = (e != nullptr && e[0] != '\0' && e[0] != '0') ? 1 : 0;
        g_det_mode.store(m, std::memory_order_relaxed);
    }
    return m == 1;
}
extern "C" int vc_set_deterministic(int32_t on) {
    g_det_mode.store(on ? 1 : 0, std::memory_order_relaxed);
    return 0;
}
extern "C" int vc_get_deterministic(void) { return vc_det_mode() ? 1 : 0; }

#define VC_REQUIRE(cond, ...)         \
    do {                              \
        if (!(cond)) {                \
            vc_set_error(__VA_ARGS__); \
            return -2;                \
        }                             \
    } while (0)

static inline hipStream_t S(void* s) { return (hipStream_t)s; }

static bool pow2_channels(int C) { return C >= 64 && C <= 2048 && (C & (C - 1)) == 0; }

static int conv_dims(const vc_conv3d_desc* d, int& To, int& Ho, int& Wo) {
    VC_REQUIRE(d != nullptr, "null descriptor");
    VC_REQUIRE(d->N > 0 && d->Ci > 0 && d->Co > 0 && d->Ti > 0 && d->Hi > 0 && d->Wi > 0, "conv3d: bad dims");
    VC_REQUIRE(d->kt > 0 && d->kh > 0 && d->kw > 0 && d->st > 0 && d->sh > 0 && d->sw > 0, "conv3d: bad kernel/stride");
    VC_REQUIRE(d->pt >= 0 && d->ph >= 0 && d->pw >= 0, "conv3d: bad padding");
    To = (d->Ti + 2 * d->pt - d->kt) / d->st + 1;
    Ho = (d->Hi + 2 * d->ph - d->kh) / d->sh + 1;
    Wo = (d->Wi + 2 * d->pw - d->kw) / d->sw + 1;
    VC_REQUIRE(To > 0 && Ho > 0 && Wo > 0 && d->Ti + 2 * d->pt >= d->kt && d->Hi + 2 * d->ph >= d->kh &&
                   d->Wi + 2 * d->pw >= d->kw,
               "conv3d: empty output");
    return 0;
}

static int conv_check(const vc_conv3d_desc* d, int& To, int& Ho, int& Wo) {
    int rc = conv_dims(d, To, Ho, Wo);
    if (rc) return rc;
    VC_REQUIRE(d->Ci % 64 == 0 && d->Co % 64 == 0, "conv3d: Ci=%d and Co=%d must be multiples of 64", d->Ci, d->Co);
    VC_REQUIRE(d->kt * d->kh * d->kw <= 32, "conv3d: at most 32 taps (got %d)", d->kt * d->kh * d->kw);
    const long long in_s = (long long)d->Ti * d->Hi * d->Wi * d->Ci;
    const long long out_s = (long long)To * Ho * Wo * d->Co;
    VC_REQUIRE(in_s < (1ll << 31) && out_s < (1ll << 31),
               "conv3d: one sample exceeds 2^31 elements (in=%lld out=%lld)", in_s, out_s);
    return 0;
}

// The kernels index a tensor with 32-bit element offsets.  A batch whose input or output exceeds 2^31 elements (64 clips
// of 64 x 224 x 224 at the stem's output) is run as consecutive sub-batches inside the entry point: samples do not
// interact in a conv, BatchNorm sums accumulate across the launches (atomics into the same buffer), weight gradients
// accumulate by construction, and the fused finalize (one launch's ticket) becomes the separate finalize launch after
// the last sub-batch.  Returns the largest sub-batch (== N when no split is needed).
// element limit of one tensor per launch (32-bit element offsets in the conv kernels).  VC_MAX_TENSOR_ELEMS lowers
// it: tests use that to drive the sub-batch paths at sizes a reference fits in memory for.
static long long max_tensor_elems() {
    const char* e = getenv("VC_MAX_TENSOR_ELEMS");
    const long long lim = (1ll << 31) - 1;
    if (e == nullptr) return lim;
    const long long v = atoll(e);
    return v > 0 && v < lim ? v : lim;
}
// A sub-launch of a sub-batched call takes the SHIFT of the sums (so that every sub-launch accumulates sum (y - s) with
// the s the final finalize assumes) but neither the fused finalize (ticket) nor a finalize launch of its own (params).
static VcBnFin shift_only(const VcBnFin* fin) {
    VcBnFin f;
    memset(&f, 0, sizeof(f));
    if (fin != nullptr) f.shift = fin->shift;
    return f;
}
static int conv_max_batch(const vc_conv3d_desc* d, int To, int Ho, int Wo) {
    const long long in_s = (long long)d->Ti * d->Hi * d->Wi * d->Ci;
    const long long out_s = (long long)To * Ho * Wo * d->Co;
    const long long big = in_s > out_s ? in_s : out_s;
    long long nmax = max_tensor_elems() / big;
    if (nmax < 1) nmax = 1;
    return nmax < d->N ? (int)nmax : d->N;
}

static int halo_run(const vc_conv3d_desc* d, bool dgrad, const void* src, const void* w, void* out, float* stats,
                    const VcBnFin* fin, const float* ep_params, int ep_relu, const void* ep_res,
                    int accumulate, void* workspace, size_t ws_bytes, void* stream,
                    const vc_bn_bwd_red* red = nullptr, bool* folded = nullptr);

extern "C" {

const char* vc_last_error(void) { return g_err; }
int vc_abi_version(void) { return VC_ABI_VERSION; }

int vc_conv3d_out_dims(const vc_conv3d_desc* d, int32_t* To, int32_t* Ho, int32_t* Wo) {
    int a, b, c;
    int rc = conv_dims(d, a, b, c);
    if (rc) return rc;
    *To = a;
    *Ho = b;
    *Wo = c;
    return 0;
}

// A tap that reads the zero padding from EVERY output position (its offset along some axis never lands inside
// the input, e.g. the outer temporal taps once T has shrunk to 1) is dropped from the tap tables of all three
// passes: the products it would add are exact zeros, so results are unchanged; its weight gradient stays 0.
static bool axis_live(int a, int p, int s, int in, int out) {
    for (int o = 0; o < out; ++o) {
        const int i = o * s - p + a;
        if (i >= 0 && i < in) return true;
    }
    return false;
}
static bool tap_live(const vc_conv3d_desc* d, int To, int Ho, int Wo, int a, int b, int c) {
    return axis_live(a, d->pt, d->st, d->Ti, To) && axis_live(b, d->ph, d->sh, d->Hi, Ho) &&
           axis_live(c, d->pw, d->sw, d->Wi, Wo);
}
static int live_taps(const vc_conv3d_desc* d) {
    const int To = (d->Ti + 2 * d->pt - d->kt) / d->st + 1, Ho = (d->Hi + 2 * d->ph - d->kh) / d->sh + 1,
              Wo = (d->Wi + 2 * d->pw - d->kw) / d->sw + 1;
    int n = 0;
    for (int a = 0; a < d->kt; ++a)
        for (int b = 0; b < d->kh; ++b)
            for (int c = 0; c < d->kw; ++c) n += tap_live(d, To, Ho, Wo, a, b, c) ? 1 : 0;
    return n;
}
static int fwd_nchunks(const vc_conv3d_desc* d) { return live_taps(d) * (d->Ci / 64); }

// stride 1, "same" padding, more than one tap, every tap within one plane / row / column of the output
// position: served by the halo kernel (halo_conv.hip)
static bool halo_ok(const vc_conv3d_desc* d) {
    static const bool off = getenv("VC_NO_HALO") != nullptr;
    if (off) return false;
    if (d->st != 1 || d->sh != 1 || d->sw != 1) return false;
    if (d->kt * d->kh * d->kw < 2) return false;
    if (d->kt != 2 * d->pt + 1 || d->kh != 2 * d->ph + 1 || d->kw != 2 * d->pw + 1) return false;
    if (d->kt > 3 || d->kh > 3 || d->kw > 3) return false;
    return true;
}
// canonical 3x3x3 / stride 1 / pad 1: weight gradient by wgrad9_conv.hip
static bool wgrad9_ok(const vc_conv3d_desc* d) {
    static const bool off = getenv("VC_NO_WGRAD9") != nullptr;
    return !off && d->kt == 3 && d->kh == 3 && d->kw == 3 && d->st == 1 && d->sh == 1 && d->sw == 1 && d->pt == 1 &&
           d->ph == 1 && d->pw == 1;
}
// A tap of a stride-1 conv whose offset along an axis is at least the extent of that axis only ever reads the zero
// padding (e.g. the two outer temporal taps when T == 1, as in layer4 at 16-frame clips): it is dropped from the
// tap table.  The result is unchanged (those products are exact zeros), the work is not done.
static bool halo_tap_live(const vc_conv3d_desc* d, int a, int b, int c) {
    return tap_live(d, d->Ti, d->Hi, d->Wi, a, b, c);  // stride 1, "same" padding: output dims = input dims
}
static int halo_ntaps(const vc_conv3d_desc* d) {
    int n = 0;
    for (int a = 0; a < d->kt; ++a)
        for (int b = 0; b < d->kh; ++b)
            for (int c = 0; c < d->kw; ++c) n += halo_tap_live(d, a, b, c) ? 1 : 0;
    return n;
}
static int halo_plan(const vc_conv3d_desc* d, bool dgrad, VcHaloPlan* plan) {
    return vc_halo_plan(d->N * d->Ti, d->Ti, d->Hi, d->Wi, dgrad ? d->Co : d->Ci, dgrad ? d->Ci : d->Co, halo_ntaps(d), plan);
}
static void halo_fill(VcHaloParams& p, const vc_conv3d_desc* d, bool dgrad) {
    memset(&p, 0, sizeof(p));
    p.NP = d->N * d->Ti, p.T = d->Ti, p.H = d->Hi, p.W = d->Wi;
    p.Cs = dgrad ? d->Co : d->Ci;
    p.Cn = dgrad ? d->Ci : d->Co;
    p.Kw = d->kt * d->kh * d->kw * p.Cs;
    int i = 0, n = 0;
    for (int a = 0; a < d->kt; ++a)
        for (int b = 0; b < d->kh; ++b)
            for (int c = 0; c < d->kw; ++c, ++i) {
                if (!halo_tap_live(d, a, b, c)) continue;
                VcTap& t = p.taps[n++];
                // forward: source = out + (a - p); data gradient: source (in dy) = in + (p - a)
                t.ot = (short)(dgrad ? d->pt - a : a - d->pt);
                t.oh = (short)(dgrad ? d->ph - b : b - d->ph);
                t.ow = (short)(dgrad ? d->pw - c : c - d->pw);
                t.wk_off = i * p.Cs;
            }
    p.ntaps = n;
}

// largest tap count over the stride-parity classes of the data gradient
static int dgrad_max_taps(const vc_conv3d_desc* d, bool* has_empty_class) {
    int best = 0;
    bool empty = false;
    const int To = (d->Ti + 2 * d->pt - d->kt) / d->st + 1, Ho = (d->Hi + 2 * d->ph - d->kh) / d->sh + 1,
              Wo = (d->Wi + 2 * d->pw - d->kw) / d->sw + 1;
    for (int rt = 0; rt < d->st; ++rt)
        for (int rh = 0; rh < d->sh; ++rh)
            for (int rw = 0; rw < d->sw; ++rw) {
                int cnt = 0;
                for (int a = 0; a < d->kt; ++a)
                    for (int b = 0; b < d->kh; ++b)
                        for (int c = 0; c < d->kw; ++c)
                            if ((((rt + d->pt - a) % d->st) + d->st) % d->st == 0 &&
                                (((rh + d->ph - b) % d->sh) + d->sh) % d->sh == 0 &&
                                (((rw + d->pw - c) % d->sw) + d->sw) % d->sw == 0 && tap_live(d, To, Ho, Wo, a, b, c))
                                ++cnt;
                if (cnt == 0) empty = true;
                if (cnt > best) best = cnt;
            }
    if (has_empty_class) *has_empty_class = empty;
    return best;
}

static int dgrad_splits(const vc_conv3d_desc* d) {
    bool empty = false;
    const int taps = dgrad_max_taps(d, &empty);
    if (empty) return 1;  // a class with no tap is zero-filled, not computed: keep the direct path
    const int Ta = (d->Ti + d->st - 1) / d->st, Ha = (d->Hi + d->sh - 1) / d->sh, Wa = (d->Wi + d->sw - 1) / d->sw;
    return vc_igemm_splits(d->N * Ta * Ha * Wa, d->Ci, taps * (d->Co / 64));
}

static size_t workspace_bytes_one(const vc_conv3d_desc* d, int32_t op);

size_t vc_conv3d_workspace_bytes(const vc_conv3d_desc* d, int32_t op) {
    int To, Ho, Wo;
    if (conv_check(d, To, Ho, Wo)) return 0;
    const int nb = conv_max_batch(d, To, Ho, Wo);
    if (nb == d->N) return workspace_bytes_one(d, op);
    vc_conv3d_desc a = *d, b = *d;  // full sub-batches and the remainder
    a.N = nb;
    b.N = d->N % nb;
    const size_t wa = workspace_bytes_one(&a, op), wb = b.N > 0 ? workspace_bytes_one(&b, op) : 0;
    return wa > wb ? wa : wb;
}

static size_t workspace_bytes_one(const vc_conv3d_desc* d, int32_t op) {
    int To, Ho, Wo;
    if (conv_check(d, To, Ho, Wo)) return 0;
    if ((op == 0 || op == 1) && halo_ok(d)) {
        VcHaloPlan plan;
        if (halo_plan(d, op == 1, &plan)) return 0;
        const size_t out_e = (size_t)d->N * d->Ti * d->Hi * d->Wi * (op == 1 ? d->Ci : d->Co);
        return plan.splits > 1 ? (size_t)plan.splits * out_e * 4 : 0;
    }
    if (op == 0) {
        const int s = vc_igemm_splits(d->N * To * Ho * Wo, d->Co, fwd_nchunks(d));
        return s > 1 ? (size_t)s * d->N * To * Ho * Wo * d->Co * 4 : 0;
    }
    if (op == 1) {
        const int s = dgrad_splits(d);
        return s > 1 ? (size_t)s * d->N * d->Ti * d->Hi * d->Wi * d->Ci * 4 : 0;
    }
    if (op == 2 && wgrad9_ok(d)) {
        int rps;
        const int s = vc_wgrad9_plan((long)d->N * d->Ti * d->Hi * d->Wi, d->Wi, d->Ci, d->Co, d->Ti > 1 ? 3 : 1, &rps);
        if (s > 0) return s > 1 ? (size_t)s * d->Co * 27 * d->Ci * 4 : 0;
    }
    if (op == 2) {
        int rps;
        const int taps = d->kt * d->kh * d->kw;
        const int s = vc_wgrad_splits(d->N * To * Ho * Wo, d->Co, live_taps(d) * (d->Ci / 64), &rps);
        return s > 1 ? (size_t)s * d->Co * taps * d->Ci * 4 : 0;
    }
    return 0;
}

static int conv3d_fwd_impl(const vc_conv3d_desc* d, const void* x, const void* w, void* y, float* bn_stats,
                           const VcBnFin* fin, const float* ep_params, int ep_relu, const void* ep_res, void* workspace,
                           size_t ws_bytes, void* stream) {
    int To, Ho, Wo;
    int rc = conv_check(d, To, Ho, Wo);
    if (rc) return rc;
    VC_REQUIRE(x && w && y, "conv3d_fwd: null pointer");
    const int nb = conv_max_batch(d, To, Ho, Wo);
    if (nb < d->N) {  // sub-batches (see conv_max_batch)
        const size_t in_s = (size_t)d->Ti * d->Hi * d->Wi * d->Ci, out_s = (size_t)To * Ho * Wo * d->Co;
        const VcBnFin sub_fin = shift_only(fin);
        for (int n0 = 0; n0 < d->N; n0 += nb) {
            vc_conv3d_desc sub = *d;
            sub.N = d->N - n0 < nb ? d->N - n0 : nb;
            rc = conv3d_fwd_impl(&sub, (const bf16*)x + n0 * in_s, w, (bf16*)y + n0 * out_s, bn_stats,
                                 fin != nullptr && bn_stats != nullptr ? &sub_fin : nullptr, ep_params, ep_relu,
                                 ep_res ? (const bf16*)ep_res + n0 * out_s : nullptr, workspace, ws_bytes, stream);
            if (rc) return rc;
        }
        if (fin == nullptr || bn_stats == nullptr) return 0;
        return vc_ew_bn_finalize(bn_stats, d->Co, fin->count, fin->gamma, fin->beta, fin->running_mean, fin->running_var,
                                 fin->momentum, fin->eps, fin->params, fin->shift, fin->shift_out, S(stream));
    }
    if (halo_ok(d)) return halo_run(d, false, x, w, y, bn_stats, fin, ep_params, ep_relu, ep_res, 0, workspace, ws_bytes, stream);
    VcIgemmParams p;
    memset(&p, 0, sizeof(p));
    p.src = (const bf16*)x;
    p.wgt = (const bf16*)w;
    p.out = (bf16*)y;
    p.stats = bn_stats;
    p.M = d->N * To * Ho * Wo;
    p.Ta = To, p.Ha = Ho, p.Wa = Wo;
    p.dWa = vc_make_fastdiv(Wo), p.dHa = vc_make_fastdiv(Ho), p.dTa = vc_make_fastdiv(To);
    p.Ts = d->Ti, p.Hs = d->Hi, p.Ws = d->Wi, p.Cs = d->Ci;
    p.st = d->st, p.sh = d->sh, p.sw = d->sw;
    p.Kw = d->kt * d->kh * d->kw * d->Ci;
    p.Cn = d->Co;
    p.out_w = d->Co, p.out_h = Wo * d->Co, p.out_t = Ho * Wo * d->Co, p.out_n = To * Ho * Wo * d->Co, p.out_base = 0;
    p.accumulate = 0;
    int i = 0, nlive = 0;
    for (int a = 0; a < d->kt; ++a)
        for (int b = 0; b < d->kh; ++b)
            for (int c = 0; c < d->kw; ++c, ++i) {
                if (!tap_live(d, To, Ho, Wo, a, b, c)) continue;
                VcTap& t = p.taps[nlive++];
                t.ot = (short)(a - d->pt), t.oh = (short)(b - d->ph), t.ow = (short)(c - d->pw);
                t.src_off = ((t.ot * d->Hi + t.oh) * d->Wi + t.ow) * d->Ci;
                t.wk_off = i * d->Ci;
            }
    p.ntaps = nlive;
    const int splits = vc_igemm_splits(p.M, p.Cn, fwd_nchunks(d));
    const size_t out_e = (size_t)p.M * d->Co;
    if (splits > 1 && workspace != nullptr && ws_bytes >= (size_t)splits * out_e * 4) {
        p.splits = splits;
        p.cps = (fwd_nchunks(d) + splits - 1) / splits;
        p.partial = (float*)workspace;
        p.partial_stride = (long long)out_e;
        p.stats = nullptr;
        rc = vc_launch_igemm(p, 0, S(stream));
        if (rc) return rc;
        rc = vc_ew_splitk_finalize(p.partial, splits, p.partial_stride, y, 0, bn_stats, p.M, d->Co, fin, S(stream));
        if (rc || ep_params == nullptr) return rc;
        return vc_ew_bn_apply(y, ep_params, ep_res, y, (long)out_e, d->Co, ep_relu, nullptr, S(stream));  // in place
    }
    if (fin != nullptr && bn_stats != nullptr) p.fin = *fin, p.stat_shift = fin->shift;
    p.ep_params = ep_params, p.ep_relu = ep_relu, p.ep_res = (const bf16*)ep_res;
    return vc_launch_igemm(p, 0, S(stream));
}

// public vc_bn_fin -> internal descriptor (validated)
static int make_fin(const vc_bn_fin* f, const float* stats, VcBnFin& o) {
    VC_REQUIRE(f && stats, "fused BatchNorm finalize needs bn_stats and a vc_bn_fin");
    VC_REQUIRE(f->ticket && f->gamma && f->beta && f->params && f->count > 0.f, "vc_bn_fin: bad arguments");
    VC_REQUIRE((f->running_mean == nullptr) == (f->running_var == nullptr), "vc_bn_fin: running statistics come in pairs");
    o.ticket = f->ticket, o.gamma = f->gamma, o.beta = f->beta;
    o.running_mean = f->running_mean, o.running_var = f->running_var, o.params = f->params;
    o.count = f->count, o.momentum = f->momentum, o.eps = f->eps;
    o.shift = f->shift, o.shift_out = f->shift_next;
    return 0;
}

int vc_conv3d_fwd(const vc_conv3d_desc* d, const void* x, const void* w, void* y, float* bn_stats, void* workspace,
                  size_t ws_bytes, void* stream) {
    return conv3d_fwd_impl(d, x, w, y, bn_stats, nullptr, nullptr, 0, nullptr, workspace, ws_bytes, stream);
}
int vc_conv3d_fwd_act(const vc_conv3d_desc* d, const void* x, const void* w, void* y, const float* bn_params,
                      const void* residual, int32_t relu, void* workspace, size_t ws_bytes, void* stream) {
    VC_REQUIRE(bn_params != nullptr, "conv3d_fwd_act: bn_params is required");
    VC_REQUIRE(residual != y || residual == nullptr, "conv3d_fwd_act: the residual must not alias the output");
    return conv3d_fwd_impl(d, x, w, y, nullptr, nullptr, bn_params, relu ? 1 : 0, residual, workspace, ws_bytes, stream);
}
int vc_conv3d_fwd_bn(const vc_conv3d_desc* d, const void* x, const void* w, void* y, float* bn_stats, const vc_bn_fin* fin,
                     void* workspace, size_t ws_bytes, void* stream) {
    VcBnFin f;
    int rc = make_fin(fin, bn_stats, f);
    if (rc) return rc;
    return conv3d_fwd_impl(d, x, w, y, bn_stats, &f, nullptr, 0, nullptr, workspace, ws_bytes, stream);
}

}  // extern "C"

/* ---- the entry point that allocates nothing (no counterpart in the reference; see include/vc3d.h) ----
 * The split-K partial slabs of vc_conv3d_fwd live in a file-scope device array instead of a caller workspace.
 * `vc_static_scratch` below is the HOST-side stand-in of that array: it is never handed to a kernel or a memset;
 * every device holds its own copy at its own address, looked up once per device with hipGetSymbolAddress (that
 * device current) and kept in g_scratch_ptr[]. */
static __device__ __align__(256) unsigned char vc_static_scratch[VC_STATIC_SCRATCH_BYTES];
static void* g_scratch_ptr[64];
static hipError_t g_scratch_err[64];
static std::once_flag g_scratch_once[64];

static int static_scratch(void** out) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
        vc_set_error("static scratch: no current device (or device index %d >= 64)", dev);
        return -1;
    }
    std::call_once(g_scratch_once[dev], [&]() {
        void* ptr = nullptr;
        const hipError_t err = hipGetSymbolAddress(&ptr, HIP_SYMBOL(vc_static_scratch));
        g_scratch_err[dev] = err;  // kept: later calls on this device report the lookup's own error
        g_scratch_ptr[dev] = err == hipSuccess ? ptr : nullptr;
    });
    if (g_scratch_ptr[dev] == nullptr) {
        vc_set_error("static scratch: hipGetSymbolAddress failed on device %d: %s", dev, hipGetErrorString(g_scratch_err[dev]));
        return -1;
    }
    *out = g_scratch_ptr[dev];
    return 0;
}

extern "C" {

// Also the warm-up: with a device current, resolves that device's scratch address now, so that the first
// vc_conv3d_fwd_noalloc may already sit inside a stream capture (where a symbol lookup is not welcome).
size_t vc_static_scratch_bytes(void) {
    int dev = -1;
    void* unused = nullptr;
    if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) (void)static_scratch(&unused);
    return VC_STATIC_SCRATCH_BYTES;
}

int vc_conv3d_fwd_noalloc(const vc_conv3d_desc* d, const void* x, const void* w, void* y, float* bn_stats, void* stream) {
    int To, Ho, Wo;
    int rc = conv_check(d, To, Ho, Wo);
    if (rc) return rc;
    const size_t need = vc_conv3d_workspace_bytes(d, 0);
    if (need == 0)  // this geometry is not split: no scratch involved at all
        return conv3d_fwd_impl(d, x, w, y, bn_stats, nullptr, nullptr, 0, nullptr, nullptr, 0, stream);
    if (need > (size_t)VC_STATIC_SCRATCH_BYTES) {
        vc_set_error("conv3d_fwd_noalloc: this geometry needs %zu bytes of split-K scratch, the static scratch holds %zu; "
                     "use vc_conv3d_fwd with a caller-owned workspace (or a smaller batch)",
                     need, (size_t)VC_STATIC_SCRATCH_BYTES);
        return -3;
    }
    void* scratch = nullptr;
    rc = static_scratch(&scratch);
    if (rc) return rc;
    // nothing in the scratch has to start at zero: every slab element that the finalize kernel reads is written by
    // the conv kernel of the same call first (stream order)
    return conv3d_fwd_impl(d, x, w, y, bn_stats, nullptr, nullptr, 0, nullptr, scratch, (size_t)VC_STATIC_SCRATCH_BYTES, stream);
}

}  // extern "C"

// the split-K finalize of a data gradient takes the BatchNorm backward reduction over the finished gradient along
// (vc_ew_splitk_finalize_red) when the caller asked for one: *folded reports it.  VC_NO_FINRED=1: separate launches.
static bool finred_on() {
    static const bool off = getenv("VC_NO_FINRED") != nullptr;
    return !off;
}
static int finalize_dgrad(const float* slabs, int splits, long long stride, void* dx, int accumulate, long rows, int C,
                          const vc_bn_bwd_red* red, bool* folded, void* stream) {
    if (red != nullptr && folded != nullptr && finred_on()) {
        *folded = true;
        return vc_ew_splitk_finalize_red(slabs, splits, stride, dx, accumulate, rows, C, red->y, red->params, red->relu_bits,
                                         red->sums, VcBnBwdFin{red->ticket, red->coef, red->dgamma, red->dbeta, (float)rows},
                                         S(stream));
    }
    return vc_ew_splitk_finalize(slabs, splits, stride, dx, accumulate, nullptr, rows, C, nullptr, S(stream));
}

static int halo_run(const vc_conv3d_desc* d, bool dgrad, const void* src, const void* w, void* out, float* stats,
                    const VcBnFin* fin, const float* ep_params, int ep_relu, const void* ep_res,
                    int accumulate, void* workspace, size_t ws_bytes, void* stream,
                    const vc_bn_bwd_red* red, bool* folded) {
    VcHaloPlan plan;
    if (halo_plan(d, dgrad, &plan)) {
        vc_set_error("conv3d: no halo tiling fits (H=%d W=%d)", d->Hi, d->Wi);
        return -2;
    }
    VcHaloParams p;
    halo_fill(p, d, dgrad);
    p.src = (const bf16*)src, p.wgt = (const bf16*)w, p.out = (bf16*)out;
    p.stats = stats;
    p.stat_shift = (stats != nullptr && fin != nullptr) ? fin->shift : nullptr;
    p.accumulate = accumulate;
    const size_t out_e = (size_t)p.NP * p.H * p.W * p.Cn;
    const int nchunks = p.Cs / plan.CK;
    if (plan.splits > 1 && (workspace == nullptr || ws_bytes < (size_t)plan.splits * out_e * 4)) plan.splits = 1;
    if (plan.splits > 1) {
        p.splits = plan.splits;
        p.cps = (nchunks + plan.splits - 1) / plan.splits;
        p.partial = (float*)workspace;
        p.partial_stride = (long long)out_e;
        p.stats = nullptr;
        int rc = vc_launch_halo(p, plan, S(stream));
        if (rc) return rc;
        if (dgrad && stats == nullptr && ep_params == nullptr)
            return finalize_dgrad(p.partial, plan.splits, p.partial_stride, out, accumulate, (long)(out_e / p.Cn), p.Cn, red,
                                  folded, stream);
        rc = vc_ew_splitk_finalize(p.partial, plan.splits, p.partial_stride, out, accumulate, stats,
                                   (long)(out_e / p.Cn), p.Cn, stats ? fin : nullptr, S(stream));
        if (rc || ep_params == nullptr) return rc;
        return vc_ew_bn_apply(out, ep_params, ep_res, out, (long)out_e, p.Cn, ep_relu, nullptr, S(stream));  // in place
    }
    p.ep_params = ep_params, p.ep_relu = ep_relu, p.ep_res = (const bf16*)ep_res;
    // The halo kernel runs several rounds of one workgroup per CU: a ticket round trip at the end of every workgroup
    // costs more there than the separate finalize launch (measured), so the finalize stays a launch of its own.
    int rc = vc_launch_halo(p, plan, S(stream));
    if (rc || fin == nullptr || stats == nullptr || fin->params == nullptr) return rc;  // params null: shift_only()
    return vc_ew_bn_finalize(stats, p.Cn, fin->count, fin->gamma, fin->beta, fin->running_mean, fin->running_var,
                             fin->momentum, fin->eps, fin->params, fin->shift, fin->shift_out, S(stream));
}

extern "C" {

static inline int floordiv(int a, int b) { return (a >= 0) ? a / b : -((-a + b - 1) / b); }
static inline int posmod(int a, int b) { return ((a % b) + b) % b; }

// The data gradient goes out as one dense GEMM launch of pw_conv.hip (1x1x1, unit stride, unsplit, one batch): the
// route on which vc_conv3d_dgrad_fused's extras ride along.
static bool dgrad_dense_route(const vc_conv3d_desc* d, const void* workspace, size_t ws_bytes) {
    int To, Ho, Wo;
    if (conv_dims(d, To, Ho, Wo)) return false;
    if (conv_max_batch(d, To, Ho, Wo) < d->N) return false;
    if (d->kt * d->kh * d->kw != 1 || d->st != 1 || d->sh != 1 || d->sw != 1 || d->pt != 0 || d->ph != 0 || d->pw != 0)
        return false;
    const size_t dx_e = (size_t)d->N * d->Ti * d->Hi * d->Wi * d->Ci;
    const int splits = dgrad_splits(d);
    if (splits > 1 && workspace != nullptr && ws_bytes >= (size_t)splits * dx_e * 4) return false;
    return getenv("VC_NO_PW_KERNEL") == nullptr && getenv("VC_NO_POINTWISE") == nullptr && getenv("VC_NO_BNRED_FOLD") == nullptr;
}

// vc_conv3d_dgrad_fused's extras -- red (nullable): the BatchNorm backward reduction over dx that the caller wants with
// it; acc_bits (nullable, with accumulate): ReLU bits the old dx is masked by before the sum.  *folded tells whether the
// conv's own launch has taken them (both or neither).
static int conv3d_dgrad_impl(const vc_conv3d_desc* d, const void* dy, const void* w_dgrad, void* dx, int accumulate,
                             const uint8_t* acc_bits, const vc_bn_bwd_red* red, bool* folded, void* workspace,
                             size_t ws_bytes, void* stream) {
    int To, Ho, Wo;
    int rc = conv_check(d, To, Ho, Wo);
    if (rc) return rc;
    VC_REQUIRE(dy && w_dgrad && dx, "conv3d_dgrad: null pointer");
    const int nb = conv_max_batch(d, To, Ho, Wo);
    if (nb < d->N) {  // sub-batches (see conv_max_batch)
        const size_t in_s = (size_t)d->Ti * d->Hi * d->Wi * d->Ci, out_s = (size_t)To * Ho * Wo * d->Co;
        for (int n0 = 0; n0 < d->N; n0 += nb) {
            vc_conv3d_desc sub = *d;
            sub.N = d->N - n0 < nb ? d->N - n0 : nb;
            rc = conv3d_dgrad_impl(&sub, (const bf16*)dy + n0 * out_s, w_dgrad, (bf16*)dx + n0 * in_s, accumulate, nullptr,
                                   nullptr, nullptr, workspace, ws_bytes, stream);
            if (rc) return rc;
        }
        return 0;
    }
    // (3x3x3 stride-1: the halo kernel.  Folding the reduction into its copy-out as well measured no gain -- its
    // epilogue is exposed, and the extra y reads cost it what the separate pass costs -- so `red` runs as a launch there.)
    // (with split-K slabs the reduction rides in the finalize launch instead: halo_run / finalize_dgrad)
    if (halo_ok(d))
        return halo_run(d, true, dy, w_dgrad, dx, nullptr, nullptr, nullptr, 0, nullptr, accumulate, workspace, ws_bytes, stream,
                        acc_bits == nullptr ? red : nullptr, folded);
    const int ntaps_all = d->kt * d->kh * d->kw;
    const size_t dx_e = (size_t)d->N * d->Ti * d->Hi * d->Wi * d->Ci;
    int splits = dgrad_splits(d);
    if (splits > 1 && (workspace == nullptr || ws_bytes < (size_t)splits * dx_e * 4)) splits = 1;
    if ((red != nullptr || acc_bits != nullptr) && dgrad_dense_route(d, workspace, ws_bytes)) {
        // a dense GEMM whose epilogue holds the finished dx tile: the reduction rides along (pw_conv.hip)
        VcPwParams w = {};
        w.src = (const bf16*)dy, w.wgt = (const bf16*)w_dgrad, w.out = (bf16*)dx;
        w.M = (int)(dx_e / d->Ci), w.K = d->Co, w.Cn = d->Ci, w.accumulate = accumulate;
        w.acc_bits = accumulate ? acc_bits : nullptr;
        if (red != nullptr) {
            w.red_y = (const bf16*)red->y, w.red_params = red->params, w.red_bits = red->relu_bits, w.red_sums = red->sums;
            w.red_fin = VcBnBwdFin{red->ticket, red->coef, red->dgamma, red->dbeta, (float)w.M};
        }
        *folded = true;
        return vc_launch_pw(w, S(stream));
    }
    // a parity class with no contributing tap gets zeros
    bool need_zero = false;
    if (!accumulate) {
        for (int rt = 0; rt < d->st && !need_zero; ++rt)
            for (int rh = 0; rh < d->sh && !need_zero; ++rh)
                for (int rw = 0; rw < d->sw && !need_zero; ++rw) {
                    int cnt = 0;
                    for (int a = 0; a < d->kt; ++a)
                        for (int b = 0; b < d->kh; ++b)
                            for (int c = 0; c < d->kw; ++c)
                                if (posmod(rt + d->pt - a, d->st) == 0 && posmod(rh + d->ph - b, d->sh) == 0 &&
                                    posmod(rw + d->pw - c, d->sw) == 0 && tap_live(d, To, Ho, Wo, a, b, c))
                                    ++cnt;
                    if (cnt == 0) need_zero = true;
                }
        if (need_zero) {
            if (hipMemsetAsync(dx, 0, (size_t)d->N * d->Ti * d->Hi * d->Wi * d->Ci * 2, S(stream)) != hipSuccess) {
                vc_set_error("conv3d_dgrad: memset failed");
                return -1;
            }
        }
    }
    // every stride-parity class of input positions is its own small GEMM (own taps, own output lattice): they go
    // out as ONE launch (blockIdx.z = class) when their tap tables fit together, else one launch per class
    VcIgemmParams p;
    memset(&p, 0, sizeof(p));
    p.src = (const bf16*)dy;
    p.wgt = (const bf16*)w_dgrad;
    p.out = (bf16*)dx;
    p.stats = nullptr;
    p.Ts = To, p.Hs = Ho, p.Ws = Wo, p.Cs = d->Co;
    p.st = p.sh = p.sw = 1;
    p.Kw = ntaps_all * d->Co;
    p.Cn = d->Ci;
    p.out_w = d->sw * d->Ci;
    p.out_h = d->sh * d->Wi * d->Ci;
    p.out_t = d->st * d->Hi * d->Wi * d->Ci;
    p.out_n = d->Ti * d->Hi * d->Wi * d->Ci;
    p.accumulate = accumulate;
    if (splits > 1) {
        p.splits = splits;
        p.partial = (float*)workspace;
        p.partial_stride = (long long)dx_e;
    }
    int ncls = 0, ntap = 0, maxM = 0;
    const bool batch = ntaps_all <= 32 && d->st * d->sh * d->sw <= 8 && getenv("VC_NO_DGRAD_BATCH") == nullptr;
    for (int rt = 0; rt < d->st; ++rt)
        for (int rh = 0; rh < d->sh; ++rh)
            for (int rw = 0; rw < d->sw; ++rw) {
                VcIgemmParams::Class c;
                memset(&c, 0, sizeof(c));
                c.Ta = (d->Ti - rt + d->st - 1) / d->st;
                c.Ha = (d->Hi - rh + d->sh - 1) / d->sh;
                c.Wa = (d->Wi - rw + d->sw - 1) / d->sw;
                if (c.Ta <= 0 || c.Ha <= 0 || c.Wa <= 0) continue;
                c.M = d->N * c.Ta * c.Ha * c.Wa;
                c.dWa = vc_make_fastdiv(c.Wa), c.dHa = vc_make_fastdiv(c.Ha), c.dTa = vc_make_fastdiv(c.Ta);
                c.out_base = ((rt * d->Hi + rh) * d->Wi + rw) * d->Ci;
                c.tap0 = batch ? ntap : 0;
                int n = 0, i = 0;
                for (int a = 0; a < d->kt; ++a)
                    for (int b = 0; b < d->kh; ++b)
                        for (int cc = 0; cc < d->kw; ++cc, ++i) {
                            const int nt = rt + d->pt - a, nh = rh + d->ph - b, nw = rw + d->pw - cc;
                            if (posmod(nt, d->st) || posmod(nh, d->sh) || posmod(nw, d->sw)) continue;
                            if (!tap_live(d, To, Ho, Wo, a, b, cc)) continue;  // pairs no output with a real input
                            VcTap& t = p.taps[c.tap0 + n++];
                            t.ot = (short)floordiv(nt, d->st), t.oh = (short)floordiv(nh, d->sh),
                            t.ow = (short)floordiv(nw, d->sw);
                            t.src_off = ((t.ot * Ho + t.oh) * Wo + t.ow) * d->Co;
                            t.wk_off = i * d->Co;
                        }
                c.ntaps = n;
                if (n == 0) continue;
                if (batch) {
                    p.cls[ncls++] = c;
                    ntap += n;
                    if (c.M > maxM) maxM = c.M;
                    continue;
                }
                p.ncls = 0;
                p.M = c.M, p.Ta = c.Ta, p.Ha = c.Ha, p.Wa = c.Wa, p.dWa = c.dWa, p.dHa = c.dHa, p.dTa = c.dTa;
                p.out_base = c.out_base, p.ntaps = n;
                if (splits > 1) p.cps = (n * (d->Co / 64) + splits - 1) / splits;
                rc = vc_launch_igemm(p, 0, S(stream));
                if (rc) return rc;
            }
    if (batch && ncls > 0) {
        if (ncls == 1) {  // a single class: the plain form
            const VcIgemmParams::Class& c = p.cls[0];
            p.ncls = 0;
            p.M = c.M, p.Ta = c.Ta, p.Ha = c.Ha, p.Wa = c.Wa, p.dWa = c.dWa, p.dHa = c.dHa, p.dTa = c.dTa;
            p.out_base = c.out_base, p.ntaps = c.ntaps;
            if (splits > 1) p.cps = (c.ntaps * (d->Co / 64) + splits - 1) / splits;
        } else {
            p.ncls = ncls;
            p.M = maxM;
            p.ntaps = ntap;
        }
        rc = vc_launch_igemm(p, 0, S(stream));
        if (rc) return rc;
    }
    if (splits > 1)
        return finalize_dgrad((const float*)workspace, splits, (long long)dx_e, dx, accumulate, (long)(dx_e / d->Ci), d->Ci,
                              acc_bits == nullptr ? red : nullptr, folded, stream);
    return 0;
}

int vc_conv3d_dgrad(const vc_conv3d_desc* d, const void* dy, const void* w_dgrad, void* dx, int accumulate,
                    void* workspace, size_t ws_bytes, void* stream) {
    return conv3d_dgrad_impl(d, dy, w_dgrad, dx, accumulate, nullptr, nullptr, nullptr, workspace, ws_bytes, stream);
}

int vc_conv3d_dgrad_fused(const vc_conv3d_desc* d, const void* dy, const void* w_dgrad, void* dx, int accumulate,
                          const uint8_t* acc_relu_bits, const vc_bn_bwd_red* red, void* workspace, size_t ws_bytes,
                          void* stream) {
    VC_REQUIRE(d, "conv3d_dgrad_fused: null descriptor");
    VC_REQUIRE(acc_relu_bits == nullptr || accumulate, "conv3d_dgrad_fused: acc_relu_bits needs accumulate");
    if (red != nullptr) {
        VC_REQUIRE(red->y && red->params && red->sums && red->ticket && red->coef,
                   "conv3d_dgrad_fused: y, params, sums, ticket and coef are required");
        VC_REQUIRE(pow2_channels(d->Ci), "bn: C=%d must be a power of two in [64, 2048]", d->Ci);
    }
    const int64_t rows = (int64_t)d->N * d->Ti * d->Hi * d->Wi;
    bool folded = false;
    int rc;
    if (acc_relu_bits != nullptr && !dgrad_dense_route(d, workspace, ws_bytes)) {
        // the general route: dx is masked in a pass of its own, then the plain accumulate (and the reduction after it)
        VC_REQUIRE(dx, "conv3d_dgrad_fused: null dx");
        rc = vc_ew_mask_bits(dx, acc_relu_bits, (long)(rows * d->Ci), S(stream));
        if (rc) return rc;
        acc_relu_bits = nullptr;
    }
    rc = conv3d_dgrad_impl(d, dy, w_dgrad, dx, accumulate, acc_relu_bits, red, &folded, workspace, ws_bytes, stream);
    if (rc || folded || red == nullptr) return rc;
    return vc_bn_bwd_reduce_fin(red->y, dx, red->relu_bits, red->params, red->sums, rows, d->Ci, red->relu_bits ? 3 : 2,
                                red->ticket, red->coef, red->dgamma, red->dbeta, stream);
}

int vc_conv3d_wgrad(const vc_conv3d_desc* d, const void* x, const void* dy, float* dw, void* workspace,
                    size_t ws_bytes, void* stream) {
    int To, Ho, Wo;
    int rc = conv_check(d, To, Ho, Wo);
    if (rc) return rc;
    VC_REQUIRE(x && dy && dw, "conv3d_wgrad: null pointer");
    const int nb = conv_max_batch(d, To, Ho, Wo);
    if (nb < d->N) {  // sub-batches (see conv_max_batch): dw accumulates
        const size_t in_s = (size_t)d->Ti * d->Hi * d->Wi * d->Ci, out_s = (size_t)To * Ho * Wo * d->Co;
        for (int n0 = 0; n0 < d->N; n0 += nb) {
            vc_conv3d_desc sub = *d;
            sub.N = d->N - n0 < nb ? d->N - n0 : nb;
            rc = vc_conv3d_wgrad(&sub, (const bf16*)x + n0 * in_s, (const bf16*)dy + n0 * out_s, dw, workspace, ws_bytes, stream);
            if (rc) return rc;
        }
        return 0;
    }
    if (wgrad9_ok(d)) {
        int rps9 = 0;
        const int s9 = vc_wgrad9_plan((long)d->N * d->Ti * d->Hi * d->Wi, d->Wi, d->Ci, d->Co, d->Ti > 1 ? 3 : 1, &rps9);
        if (s9 > 0) {
            VcWgrad9Params q;
            memset(&q, 0, sizeof(q));
            q.x = (const bf16*)x, q.dy = (const bf16*)dy, q.dw = dw;
            q.NP = d->N * d->Ti, q.T = d->Ti, q.H = d->Hi, q.W = d->Wi;
            q.Cs = d->Ci, q.Co = d->Co, q.Kw = 27 * d->Ci;
            q.rows_per_split = rps9, q.splits = s9;
            q.dt0 = d->Ti > 1 ? 0 : 1, q.ndt = d->Ti > 1 ? 3 : 1;  // T == 1: the dt = -1 / +1 taps only see padding
            const size_t dw_e9 = (size_t)d->Co * q.Kw;
            if (s9 > 1 && workspace != nullptr && ws_bytes >= (size_t)s9 * dw_e9 * 4) {
                q.partial = (float*)workspace;
                q.partial_stride = (long long)dw_e9;
                rc = vc_launch_wgrad9(q, S(stream));
                if (rc) return rc;
                // T == 1: only the nine dt = 0 taps are computed; the other slab entries hold whatever the workspace held
                return vc_ew_splitk_reduce_wgrad(q.partial, s9, q.partial_stride, dw, (long)dw_e9, d->Ci, 27,
                                                 q.ndt == 3 ? 0x7ffffffu : (0x1ffu << 9), S(stream));
            }
            return vc_launch_wgrad9(q, S(stream));  // no workspace: float atomics when split
        }
    }
    VcWgradParams p;
    memset(&p, 0, sizeof(p));
    p.x = (const bf16*)x;
    p.dy = (const bf16*)dy;
    p.dw = dw;
    p.M = d->N * To * Ho * Wo;
    p.Ta = To, p.Ha = Ho, p.Wa = Wo;
    p.dWa = vc_make_fastdiv(Wo), p.dHa = vc_make_fastdiv(Ho), p.dTa = vc_make_fastdiv(To);
    p.Ts = d->Ti, p.Hs = d->Hi, p.Ws = d->Wi, p.Cs = d->Ci;
    p.st = d->st, p.sh = d->sh, p.sw = d->sw;
    p.Kw = d->kt * d->kh * d->kw * d->Ci;
    p.Co = d->Co;
    int i = 0, nlive = 0;
    for (int a = 0; a < d->kt; ++a)
        for (int b = 0; b < d->kh; ++b)
            for (int c = 0; c < d->kw; ++c, ++i) {
                if (!tap_live(d, To, Ho, Wo, a, b, c)) continue;  // its weight gradient is exactly 0: dw is left alone
                VcTap& t = p.taps[nlive++];
                t.ot = (short)(a - d->pt), t.oh = (short)(b - d->ph), t.ow = (short)(c - d->pw);
                t.src_off = ((t.ot * d->Hi + t.oh) * d->Wi + t.ow) * d->Ci;
                t.wk_off = i * d->Ci;
            }
    p.ntaps = nlive;
    unsigned live_mask = 0;
    for (int t = 0; t < nlive; ++t) live_mask |= 1u << (p.taps[t].wk_off / d->Ci);
    int rps;
    const int splits = vc_wgrad_splits(p.M, p.Co, live_taps(d) * (d->Ci / 64), &rps);
    const size_t dw_e = (size_t)d->Co * p.Kw;
    if (splits > 1 && workspace != nullptr && ws_bytes >= (size_t)splits * dw_e * 4) {
        p.partial = (float*)workspace;
        p.partial_stride = (long long)dw_e;
        rc = vc_launch_wgrad(p, 0, S(stream));
        if (rc) return rc;
        return vc_ew_splitk_reduce_wgrad(p.partial, splits, p.partial_stride, dw, (long)dw_e, d->Ci, d->kt * d->kh * d->kw,
                                         live_mask, S(stream));
    }
    return vc_launch_wgrad(p, 0, S(stream));
}

int vc_pack_weight(const float* w_master, void* w_fwd, void* w_dgrad, int32_t Co, int32_t taps, int32_t Ci,
                   void* stream) {
    VC_REQUIRE(w_master && w_fwd && Co > 0 && taps > 0 && Ci > 0, "pack_weight: bad arguments");
    return vc_ew_pack_weight(w_master, w_fwd, w_dgrad, Co, taps, Ci, S(stream));
}

/* ------------------------------------------------------------------ flat-buffer optimizer and weight repack */
int vc_sgd_step(float* params, const float* grads, float* momentum_buf, void* shadow_bf16, int64_t n, float lr,
                float momentum, float weight_decay, int32_t nesterov, void* stream) {
    VC_REQUIRE(params && grads && n > 0, "sgd_step: bad arguments");
    VC_REQUIRE(((uintptr_t)params | (uintptr_t)grads | (uintptr_t)momentum_buf) % 16 == 0 &&
                   (uintptr_t)shadow_bf16 % 8 == 0,
               "sgd_step: buffers must be 16-byte aligned");
    VC_REQUIRE(!nesterov || (momentum_buf && momentum > 0.f), "sgd_step: nesterov needs momentum");
    return vc_ew_sgd_step(params, grads, momentum_buf, shadow_bf16, (long)n, lr, momentum, weight_decay, nesterov,
                          S(stream));
}
int vc_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, void* shadow_bf16, int64_t n,
                 float lr, float beta1, float beta2, float eps, float weight_decay, int64_t step, void* stream) {
    VC_REQUIRE(params && grads && exp_avg && exp_avg_sq && n > 0 && step >= 1, "adam_step: bad arguments");
    VC_REQUIRE(((uintptr_t)params | (uintptr_t)grads | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) % 16 == 0 &&
                   (uintptr_t)shadow_bf16 % 8 == 0,
               "adam_step: buffers must be 16-byte aligned");
    VC_REQUIRE(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f && eps >= 0.f, "adam_step: bad betas / eps");
    const double bias1 = 1.0 - pow((double)beta1, (double)step), bias2 = 1.0 - pow((double)beta2, (double)step);
    return vc_ew_adam_step(params, grads, exp_avg, exp_avg_sq, shadow_bf16, (long)n, (float)((double)lr / bias1), beta1, beta2,
                           eps, weight_decay, (float)sqrt(bias2), S(stream));
}
int vc_shadow_bf16(const float* params, void* shadow_bf16, int64_t n, void* stream) {
    VC_REQUIRE(params && shadow_bf16 && n > 0, "shadow_bf16: bad arguments");
    return vc_ew_shadow(params, shadow_bf16, (long)n, S(stream));
}
int vc_pack_dgrad_batched(const void* shadow_bf16, void* w_dgrad_flat, const int32_t* table_dev, const int32_t* table_host,
                          int32_t nconv, void* stream) {
    VC_REQUIRE(shadow_bf16 && w_dgrad_flat && table_dev && table_host && nconv > 0, "pack_dgrad_batched: bad arguments");
    long tiles = 0;
    for (int i = 0; i < nconv; ++i) {
        const int32_t* t = table_host + i * 6;
        VC_REQUIRE(t[0] >= 0 && t[1] >= 0 && t[2] > 0 && t[3] > 0 && t[4] > 0 && t[2] % 64 == 0 && t[4] % 64 == 0 &&
                       t[0] % 8 == 0 && t[1] % 8 == 0,
                   "pack_dgrad_batched: entry %d: channels must be multiples of 64, offsets of 8", i);
        VC_REQUIRE(t[5] == tiles, "pack_dgrad_batched: entry %d: first tile %d, expected %ld", i, t[5], tiles);
        tiles += (long)t[3] * (t[2] / 64) * (t[4] / 64);
    }
    VC_REQUIRE(tiles < (1l << 31), "pack_dgrad_batched: too many tiles");
    return vc_ew_pack_dgrad_batched(shadow_bf16, w_dgrad_flat, table_dev, nconv, (int)tiles, S(stream));
}

/* ------------------------------------------------------------------ stem */
static int stem_check(int N, int T, int H, int W, int Co) {
    VC_REQUIRE(N > 0 && T > 0 && H > 0 && W > 0, "stem: bad dims");
    VC_REQUIRE(H % 2 == 0 && W % 2 == 0, "stem: H=%d and W=%d must be even", H, W);
    VC_REQUIRE(Co % 64 == 0 && Co > 0, "stem: Co=%d must be a multiple of 64", Co);
    const long long in_s = (long long)(T + 6) * (H / 2 + 3) * ((W / 2 + 4) & ~1) * 12;
    const long long out_s = (long long)T * (H / 2) * (W / 2) * Co;
    VC_REQUIRE(in_s < (1ll << 31) && out_s < (1ll << 31), "stem: one clip exceeds 2^31 elements");
    return 0;
}
// largest sub-batch whose packed source and output stay below 2^31 elements (the stem kernels index with 32 bits)
static int stem_max_batch(int N, int T, int H, int W, int Co) {
    const long long in_s = (long long)(T + 6) * (H / 2 + 3) * ((W / 2 + 4) & ~1) * 12;
    const long long out_s = (long long)T * (H / 2) * (W / 2) * Co;
    const long long big = in_s > out_s ? in_s : out_s;
    long long nmax = max_tensor_elems() / big;
    if (nmax < 1) nmax = 1;
    return nmax < N ? (int)nmax : N;
}

size_t vc_stem_packed_input_elems(int32_t N, int32_t T, int32_t H, int32_t W) {
    return (size_t)N * (T + 6) * (H / 2 + 3) * ((W / 2 + 4) & ~1) * 12;
}

int vc_stem_pack_input(const void* x, int32_t x_is_bf16, void* xs, int32_t N, int32_t T, int32_t H, int32_t W,
                       void* stream) {
    int rc = stem_check(N, T, H, W, 64);
    if (rc) return rc;
    VC_REQUIRE(x && xs, "stem_pack_input: null pointer");
    return vc_ew_stem_pack_input(x, x_is_bf16, xs, N, T, H, W, S(stream));
}

int vc_clip_pack_u8(const uint8_t* frames, const int32_t* crops_dev, const int32_t* crops_host, void* xs, int32_t N,
                    int32_t T, int32_t H, int32_t W, int32_t Ts, int32_t Hs, int32_t Ws, int32_t tstep,
                    const float* mean, const float* std, void* stream) {
    int rc = stem_check(N, T, H, W, 64);
    if (rc) return rc;
    VC_REQUIRE(frames && crops_dev && crops_host && xs && mean && std, "clip_pack_u8: null pointer");
    VC_REQUIRE(Ts > 0 && Hs >= H && Ws >= W && tstep >= 1, "clip_pack_u8: source %dx%dx%d smaller than the crop", Ts, Hs,
               Ws);
    VC_REQUIRE((long long)N * Ts * Hs * Ws * 3 < (1ll << 40), "clip_pack_u8: source too large");
    // every read is checked here, on the host copy of the table: the kernel does no bounds tests
    for (int n = 0; n < N; ++n) {
        const int32_t* c = crops_host + 4 * n;
        VC_REQUIRE(c[0] >= 0 && (long long)c[0] + (long long)(T - 1) * tstep < Ts, "clip_pack_u8: clip %d: frames %d + %d*%d leave the video (%d frames)", n, c[0], T - 1, tstep, Ts);
        VC_REQUIRE(c[1] >= 0 && c[1] + H <= Hs && c[2] >= 0 && c[2] + W <= Ws, "clip_pack_u8: clip %d: crop at (%d, %d) leaves the %d x %d frame", n, c[1], c[2], Hs, Ws);
        VC_REQUIRE(c[3] == 0 || c[3] == 1, "clip_pack_u8: clip %d: flip must be 0 or 1", n);
    }
    float a[3], b[3];
    for (int c = 0; c < 3; ++c) {
        VC_REQUIRE(std[c] > 0.f, "clip_pack_u8: std must be positive");
        a[c] = 1.0f / (255.0f * std[c]);
        b[c] = -mean[c] / std[c];
    }
    return vc_ew_clip_pack_u8(frames, crops_dev, xs, N, T, H, W, Ts, Hs, Ws, tstep, a, b, S(stream));
}

int vc_stem_pack_weight(const float* w_master, void* ws, int32_t Co, void* stream) {
    VC_REQUIRE(w_master && ws && Co > 0, "stem_pack_weight: bad arguments");
    return vc_ew_stem_pack_weight(w_master, ws, Co, S(stream));
}

static void stem_geometry(int N, int T, int H, int W, int& M, int& Ho, int& Wo, int& Tp, int& Hp, int& Wp) {
    Ho = H / 2, Wo = W / 2;
    Tp = T + 6, Hp = Ho + 3, Wp = (Wo + 4) & ~1;  // row pitch rounded to an even count: 16-byte rows
    M = N * T * Ho * Wo;
}

static int stem_fwd_impl(const void* xs, const void* ws, void* y, float* bn_stats, const VcBnFin* fin, int32_t N,
                         int32_t T, int32_t H, int32_t W, int32_t Co, void* stream) {
    int rc = stem_check(N, T, H, W, Co);
    if (rc) return rc;
    VC_REQUIRE(xs && ws && y, "stem_fwd: null pointer");
    const int nb = stem_max_batch(N, T, H, W, Co);
    if (nb < N) {  // sub-batches: BatchNorm sums accumulate across the launches, finalize after the last one
        const size_t in_s = vc_stem_packed_input_elems(1, T, H, W), out_s = (size_t)T * (H / 2) * (W / 2) * Co;
        const VcBnFin sub_fin = shift_only(fin);
        for (int n0 = 0; n0 < N; n0 += nb) {
            rc = stem_fwd_impl((const bf16*)xs + n0 * in_s, ws, (bf16*)y + n0 * out_s, bn_stats,
                               fin != nullptr && bn_stats != nullptr ? &sub_fin : nullptr, N - n0 < nb ? N - n0 : nb, T,
                               H, W, Co, stream);
            if (rc) return rc;
        }
        if (fin == nullptr || bn_stats == nullptr) return 0;
        return vc_ew_bn_finalize(bn_stats, Co, fin->count, fin->gamma, fin->beta, fin->running_mean, fin->running_var,
                                 fin->momentum, fin->eps, fin->params, fin->shift, fin->shift_out, S(stream));
    }
    int M, Ho, Wo, Tp, Hp, Wp;
    stem_geometry(N, T, H, W, M, Ho, Wo, Tp, Hp, Wp);
    VcStemFwdParams sp;
    if (vc_stem_fwd_plan(N, T, Ho, Wo, Hp, Wp, Co, &sp)) {  // image-resident form
        sp.xs = (const bf16*)xs;
        sp.ws = (const bf16*)ws;
        sp.y = (bf16*)y;
        sp.stats = bn_stats;
        sp.stat_shift = nullptr;
        if (fin != nullptr && bn_stats != nullptr) sp.fin = *fin, sp.stat_shift = fin->shift;
        return vc_launch_stem_fwd(sp, S(stream));
    }
    VcIgemmParams p;
    memset(&p, 0, sizeof(p));
    p.src = (const bf16*)xs;
    p.wgt = (const bf16*)ws;
    p.out = (bf16*)y;
    p.stats = bn_stats;
    p.M = M;
    p.Ta = T, p.Ha = Ho, p.Wa = Wo;
    p.dWa = vc_make_fastdiv(Wo), p.dHa = vc_make_fastdiv(Ho), p.dTa = vc_make_fastdiv(T);
    p.Ts = Tp, p.Hs = Hp, p.Ws = Wp, p.Cs = 12;
    p.st = p.sh = p.sw = 1;
    p.ntaps = 0;
    p.Kw = 1344;
    p.Cn = Co;
    p.out_w = Co, p.out_h = Wo * Co, p.out_t = Ho * Wo * Co, p.out_n = T * Ho * Wo * Co, p.out_base = 0;
    if (fin != nullptr && bn_stats != nullptr) p.fin = *fin, p.stat_shift = fin->shift;
    return vc_launch_igemm(p, 1, S(stream));
}
int vc_stem_fwd(const void* xs, const void* ws, void* y, float* bn_stats, int32_t N, int32_t T, int32_t H, int32_t W,
                int32_t Co, void* stream) {
    return stem_fwd_impl(xs, ws, y, bn_stats, nullptr, N, T, H, W, Co, stream);
}
int vc_stem_fwd_bn(const void* xs, const void* ws, void* y, float* bn_stats, const vc_bn_fin* fin, int32_t N, int32_t T,
                   int32_t H, int32_t W, int32_t Co, void* stream) {
    VcBnFin f;
    int rc = make_fin(fin, bn_stats, f);
    if (rc) return rc;
    return stem_fwd_impl(xs, ws, y, bn_stats, &f, N, T, H, W, Co, stream);
}

size_t vc_stem_wgrad_workspace_bytes(int32_t N, int32_t T, int32_t H, int32_t W, int32_t Co) {
    if (stem_check(N, T, H, W, Co)) return 0;
    const int nb = stem_max_batch(N, T, H, W, Co);
    if (nb < N) {
        const size_t wa = vc_stem_wgrad_workspace_bytes(nb, T, H, W, Co);
        const size_t wb = N % nb ? vc_stem_wgrad_workspace_bytes(N % nb, T, H, W, Co) : 0;
        return wa > wb ? wa : wb;
    }
    int M, Ho, Wo, Tp, Hp, Wp;
    stem_geometry(N, T, H, W, M, Ho, Wo, Tp, Hp, Wp);
    VcStemWgradParams sp;
    if (!vc_stem_wgrad_plan(N, T, Ho, Wo, Hp, Wp, Co, &sp)) return 0;
    return vc_stem_wgrad_slab_bytes(sp);
}

int vc_stem_wgrad(const void* xs, const void* dy, float* dws, int32_t N, int32_t T, int32_t H, int32_t W, int32_t Co,
                  void* workspace, size_t ws_bytes, void* stream) {
    int rc = stem_check(N, T, H, W, Co);
    if (rc) return rc;
    VC_REQUIRE(xs && dy && dws, "stem_wgrad: null pointer");
    const int nb = stem_max_batch(N, T, H, W, Co);
    if (nb < N) {  // sub-batches: dws accumulates
        const size_t in_s = vc_stem_packed_input_elems(1, T, H, W), out_s = (size_t)T * (H / 2) * (W / 2) * Co;
        for (int n0 = 0; n0 < N; n0 += nb) {
            rc = vc_stem_wgrad((const bf16*)xs + n0 * in_s, (const bf16*)dy + n0 * out_s, dws, N - n0 < nb ? N - n0 : nb, T, H, W, Co,
                               workspace, ws_bytes, stream);
            if (rc) return rc;
        }
        return 0;
    }
    int M, Ho, Wo, Tp, Hp, Wp;
    stem_geometry(N, T, H, W, M, Ho, Wo, Tp, Hp, Wp);
    // image-resident form (persistent workgroups, per-workgroup fp32 slabs in the workspace)
    VcStemWgradParams sp;
    if (workspace != nullptr && vc_stem_wgrad_plan(N, T, Ho, Wo, Hp, Wp, Co, &sp) &&
        ws_bytes >= vc_stem_wgrad_slab_bytes(sp)) {
        sp.xs = (const bf16*)xs;
        sp.dy = (const bf16*)dy;
        sp.slabs = (float*)workspace;
        return vc_launch_stem_wgrad(sp, dws, S(stream));
    }
    VcWgradParams p;
    memset(&p, 0, sizeof(p));
    p.x = (const bf16*)xs;
    p.dy = (const bf16*)dy;
    p.dw = dws;
    p.M = M;
    p.Ta = T, p.Ha = Ho, p.Wa = Wo;
    p.dWa = vc_make_fastdiv(Wo), p.dHa = vc_make_fastdiv(Ho), p.dTa = vc_make_fastdiv(T);
    p.Ts = Tp, p.Hs = Hp, p.Ws = Wp, p.Cs = 12;
    p.st = p.sh = p.sw = 1;
    p.ntaps = 0;
    p.Kw = 1344;
    p.Co = Co;
    return vc_launch_wgrad(p, 1, S(stream));
}

int vc_stem_unpack_wgrad(const float* dws, float* dw_master, int32_t Co, void* stream) {
    VC_REQUIRE(dws && dw_master && Co > 0, "stem_unpack_wgrad: bad arguments");
    return vc_ew_stem_unpack_wgrad(dws, dw_master, Co, S(stream));
}

/* ------------------------------------------------------------------ batch norm */
int vc_bn_stats(const void* y, float* sums, float* shift_out, int64_t rows, int32_t C, void* stream) {
    VC_REQUIRE(y && sums && rows > 0, "bn_stats: bad arguments");
    VC_REQUIRE(pow2_channels(C), "bn: C=%d must be a power of two in [64, 2048]", C);
    return vc_ew_bn_stats(y, sums, rows, C, shift_out, S(stream));
}
int vc_bn_finalize(const float* stats, int32_t C, float count, const float* gamma, const float* beta,
                   float* running_mean, float* running_var, float momentum, float eps, float* params,
                   const float* shift, float* shift_next, void* stream) {
    VC_REQUIRE(stats && gamma && beta && params && C > 0 && count > 0, "bn_finalize: bad arguments");
    VC_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bn_finalize: running stats must come in pairs");
    return vc_ew_bn_finalize(stats, C, count, gamma, beta, running_mean, running_var, momentum, eps, params, shift, shift_next,
                             S(stream));
}
int vc_bn_eval_params(int32_t C, const float* gamma, const float* beta, const float* running_mean,
                      const float* running_var, float eps, float* params, void* stream) {
    VC_REQUIRE(gamma && beta && running_mean && running_var && params && C > 0, "bn_eval_params: bad arguments");
    return vc_ew_bn_eval_params(C, gamma, beta, running_mean, running_var, eps, params, S(stream));
}
int vc_bn_apply(const void* y, const float* params, const void* residual, void* out, int64_t elems, int32_t C,
                int32_t relu, void* stream) {
    VC_REQUIRE(y && params && out && elems > 0, "bn_apply: bad arguments");
    VC_REQUIRE(C % 8 == 0 && elems % C == 0, "bn_apply: C=%d must divide elems and be a multiple of 8", C);
    return vc_ew_bn_apply(y, params, residual, out, elems, C, relu, nullptr, S(stream));
}
int vc_bn_apply_mask(const void* y, const float* params, const void* residual, void* out, uint8_t* relu_bits, int64_t elems,
                     int32_t C, void* stream) {
    VC_REQUIRE(y && params && out && relu_bits && elems > 0, "bn_apply_mask: bad arguments");
    VC_REQUIRE(C % 8 == 0 && elems % C == 0, "bn_apply_mask: C=%d must divide elems and be a multiple of 8", C);
    return vc_ew_bn_apply(y, params, residual, out, elems, C, 1, relu_bits, S(stream));
}
int vc_bn_bwd_reduce(const void* y, const void* dz, const void* aux, const float* params, float* sums, int64_t rows,
                     int32_t C, int32_t mask, void* stream) {
    VC_REQUIRE(y && dz && params && sums && rows > 0, "bn_bwd_reduce: bad arguments");
    VC_REQUIRE(pow2_channels(C), "bn: C=%d must be a power of two in [64, 2048]", C);
    VC_REQUIRE(mask >= 0 && mask <= 3 && ((mask != 1 && mask != 3) || aux), "bn_bwd_reduce: bad mask mode");
    return vc_ew_bn_bwd_reduce(y, dz, aux, params, sums, rows, C, mask, nullptr, S(stream));
}
int vc_bn_bwd_reduce_fin(const void* y, const void* dz, const void* aux, const float* params, float* sums, int64_t rows,
                         int32_t C, int32_t mask, uint32_t* ticket, float* coef, float* dgamma, float* dbeta,
                         void* stream) {
    VC_REQUIRE(y && dz && params && sums && rows > 0, "bn_bwd_reduce_fin: bad arguments");
    VC_REQUIRE(pow2_channels(C), "bn: C=%d must be a power of two in [64, 2048]", C);
    VC_REQUIRE(mask >= 0 && mask <= 3 && ((mask != 1 && mask != 3) || aux), "bn_bwd_reduce_fin: bad mask mode");
    VC_REQUIRE(ticket && coef, "bn_bwd_reduce_fin: ticket and coef are required");
    const VcBnBwdFin f = {ticket, coef, dgamma, dbeta, (float)rows};
    return vc_ew_bn_bwd_reduce(y, dz, aux, params, sums, rows, C, mask, &f, S(stream));
}
int vc_bn_bwd_finalize(const float* sums, int32_t C, float count, float* coef, float* dgamma, float* dbeta,
                       void* stream) {
    VC_REQUIRE(sums && coef && C > 0 && count > 0, "bn_bwd_finalize: bad arguments");
    return vc_ew_bn_bwd_finalize(sums, C, count, coef, dgamma, dbeta, S(stream));
}
int vc_bn_bwd_apply(const void* y, const void* dz, const void* aux, const float* params, const float* coef, void* dy,
                    void* g_out, int64_t elems, int32_t C, int32_t mask, void* stream) {
    VC_REQUIRE(y && dz && params && coef && dy && elems > 0, "bn_bwd_apply: bad arguments");
    VC_REQUIRE(C % 8 == 0 && elems % C == 0, "bn_bwd_apply: C=%d must divide elems and be a multiple of 8", C);
    VC_REQUIRE(mask >= 0 && mask <= 3 && ((mask != 1 && mask != 3) || aux), "bn_bwd_apply: bad mask mode");
    return vc_ew_bn_bwd_apply(y, dz, aux, params, coef, dy, g_out, elems, C, mask, S(stream));
}
int vc_add(const void* a, const void* b, void* out, int64_t elems, void* stream) {
    VC_REQUIRE(a && b && out && elems > 0 && elems % 8 == 0, "add: bad arguments");
    return vc_ew_add(a, b, out, elems, S(stream));
}

/* ------------------------------------------------------------------ pooling */
static int pool_geom(const vc_pool3d_desc* d, VcPoolGeom& g) {
    VC_REQUIRE(d != nullptr, "null descriptor");
    VC_REQUIRE(d->N > 0 && d->C > 0 && d->C % 8 == 0 && d->T > 0 && d->H > 0 && d->W > 0, "pool3d: bad dims");
    VC_REQUIRE(d->kt > 0 && d->kh > 0 && d->kw > 0 && d->st > 0 && d->sh > 0 && d->sw > 0, "pool3d: bad kernel");
    VC_REQUIRE(d->kt * d->kh * d->kw <= 255, "pool3d: window too large");
    VC_REQUIRE(2 * d->pt <= d->kt && 2 * d->ph <= d->kh && 2 * d->pw <= d->kw, "pool3d: pad must be <= kernel/2");
    g.N = d->N, g.C = d->C, g.T = d->T, g.H = d->H, g.W = d->W;
    g.kt = d->kt, g.kh = d->kh, g.kw = d->kw, g.st = d->st, g.sh = d->sh, g.sw = d->sw;
    g.pt = d->pt, g.ph = d->ph, g.pw = d->pw;
    g.To = (d->T + 2 * d->pt - d->kt) / d->st + 1;
    g.Ho = (d->H + 2 * d->ph - d->kh) / d->sh + 1;
    g.Wo = (d->W + 2 * d->pw - d->kw) / d->sw + 1;
    VC_REQUIRE(g.To > 0 && g.Ho > 0 && g.Wo > 0 && d->T + 2 * d->pt >= d->kt && d->H + 2 * d->ph >= d->kh &&
                   d->W + 2 * d->pw >= d->kw,
               "pool3d: empty output");
    return 0;
}
int vc_pool3d_out_dims(const vc_pool3d_desc* d, int32_t* To, int32_t* Ho, int32_t* Wo) {
    VcPoolGeom g;
    int rc = pool_geom(d, g);
    if (rc) return rc;
    *To = g.To, *Ho = g.Ho, *Wo = g.Wo;
    return 0;
}
int vc_maxpool3d_fwd(const vc_pool3d_desc* d, const void* y, const float* bn_params, void* out, uint8_t* argmax,
                     void* stream) {
    VcPoolGeom g;
    int rc = pool_geom(d, g);
    if (rc) return rc;
    VC_REQUIRE(y && out && argmax, "maxpool3d_fwd: null pointer");
    return vc_ew_maxpool_fwd(y, bn_params, out, argmax, g, S(stream));
}
int vc_maxpool3d_bwd(const vc_pool3d_desc* d, const void* dout, const uint8_t* argmax, void* din, void* stream) {
    VcPoolGeom g;
    int rc = pool_geom(d, g);
    if (rc) return rc;
    VC_REQUIRE(dout && argmax && din, "maxpool3d_bwd: null pointer");
    return vc_ew_maxpool_bwd(dout, argmax, din, g, S(stream));
}
int vc_bn_bwd_reduce_pooled(const vc_pool3d_desc* d, const void* y, const void* dpooled, const uint8_t* argmax,
                            const void* pooled, const float* params, float* sums, void* stream) {
    VcPoolGeom g;
    int rc = pool_geom(d, g);
    if (rc) return rc;
    VC_REQUIRE(y && dpooled && argmax && params && sums, "bn_bwd_reduce_pooled: null pointer");
    VC_REQUIRE((long long)g.N * g.T * g.H < (1ll << 31), "bn_bwd_reduce_pooled: too many rows");
    VC_REQUIRE(pow2_channels(g.C), "bn: C=%d must be a power of two in [64, 2048]", g.C);
    return vc_ew_bn_bwd_reduce_pooled(y, dpooled, argmax, pooled, params, sums, g, nullptr, S(stream));
}
int vc_bn_bwd_reduce_pooled_fin(const vc_pool3d_desc* d, const void* y, const void* dpooled, const uint8_t* argmax,
                                const void* pooled, const float* params, float* sums, uint32_t* ticket, float* coef,
                                float* dgamma, float* dbeta, void* stream) {
    VcPoolGeom g;
    int rc = pool_geom(d, g);
    if (rc) return rc;
    VC_REQUIRE(y && dpooled && argmax && params && sums && ticket && coef, "bn_bwd_reduce_pooled_fin: null pointer");
    VC_REQUIRE((long long)g.N * g.T * g.H < (1ll << 31), "bn_bwd_reduce_pooled_fin: too many rows");
    VC_REQUIRE(pow2_channels(g.C), "bn: C=%d must be a power of two in [64, 2048]", g.C);
    const VcBnBwdFin f = {ticket, coef, dgamma, dbeta, (float)((long long)g.N * g.T * g.H * g.W)};
    return vc_ew_bn_bwd_reduce_pooled(y, dpooled, argmax, pooled, params, sums, g, &f, S(stream));
}
int vc_bn_bwd_apply_pooled(const vc_pool3d_desc* d, const void* y, const void* dpooled, const uint8_t* argmax,
                           const float* params, const float* coef, void* dy, void* stream) {
    VcPoolGeom g;
    int rc = pool_geom(d, g);
    if (rc) return rc;
    VC_REQUIRE(y && dpooled && argmax && params && coef && dy, "bn_bwd_apply_pooled: null pointer");
    VC_REQUIRE(g.kt <= 2 * g.st && g.kh <= 2 * g.sh && g.kw <= 2 * g.sw,
               "bn_bwd_apply_pooled: kernel must be <= 2*stride (at most 2 windows per axis)");
    VC_REQUIRE(pow2_channels(g.C), "bn: C=%d must be a power of two in [64, 2048]", g.C);
    return vc_ew_bn_bwd_apply_pooled(y, dpooled, argmax, params, coef, dy, g, S(stream));
}
int vc_gavgpool_fwd(const void* in, float* out, int32_t N, int32_t Sp, int32_t C, void* stream) {
    VC_REQUIRE(in && out && N > 0 && Sp > 0 && C > 0, "gavgpool_fwd: bad arguments");
    return vc_ew_avgpool_fwd(in, out, N, Sp, C, S(stream));
}
int vc_gavgpool_bwd(const float* dfeat, void* din, int32_t N, int32_t Sp, int32_t C, void* stream) {
    VC_REQUIRE(dfeat && din && N > 0 && Sp > 0 && C > 0, "gavgpool_bwd: bad arguments");
    return vc_ew_avgpool_bwd(dfeat, din, N, Sp, C, S(stream));
}

/* ------------------------------------------------------------------ head */
int vc_linear_fwd(const float* feat, const float* W, const float* bias, float* out, int32_t N, int32_t F, int32_t K,
                  void* stream) {
    VC_REQUIRE(feat && W && out && N > 0 && F > 0 && K > 0, "linear_fwd: bad arguments");
    return vc_ew_linear_fwd(feat, W, bias, out, N, F, K, S(stream));
}
int vc_linear_bwd(const float* dlogits, const float* feat, const float* W, float* dfeat, float* dW, float* db,
                  int32_t N, int32_t F, int32_t K, void* stream) {
    VC_REQUIRE(dlogits && N > 0 && F > 0 && K > 0, "linear_bwd: bad arguments");
    VC_REQUIRE((!dfeat || W) && (!dW || feat), "linear_bwd: missing operand");
    return vc_ew_linear_bwd(dlogits, feat, W, dfeat, dW, db, N, F, K, S(stream));
}

int vc_softmax_ce(const float* logits, const int64_t* labels, float* loss, float* dlogits, int32_t N, int32_t K, void* stream) {
    VC_REQUIRE(logits && labels && loss && N > 0 && K > 0, "softmax_ce: bad arguments");
    return vc_ew_softmax_ce(logits, (const long long*)labels, loss, dlogits, N, K, S(stream));
}

/* ------------------------------------------------------------------ LSTM cell (element-wise half) */
int vc_lstm_pointwise_fwd(const void* gates, const float* b_ih, const float* b_hh, const float* c_prev, float* h, float* c,
                          void* h_bf16, float* saved, int32_t N, int32_t H, void* stream) {
    VC_REQUIRE(gates && b_ih && b_hh && c_prev && h && c && N > 0 && H > 0, "lstm_pointwise_fwd: bad arguments");
    return vc_ew_lstm_pointwise_fwd(gates, b_ih, b_hh, c_prev, h, c, h_bf16, saved, N, H, S(stream));
}
int vc_lstm_pointwise_bwd(const float* dh, const float* dc_next, const float* saved, const float* c_prev, void* dgates,
                          float* dc_prev, float* db_ih, float* db_hh, int32_t N, int32_t H, void* stream) {
    VC_REQUIRE(saved && c_prev && dgates && (dh || dc_next) && N > 0 && H > 0, "lstm_pointwise_bwd: bad arguments");
    return vc_ew_lstm_pointwise_bwd(dh, dc_next, saved, c_prev, dgates, dc_prev, db_ih, db_hh, N, H, S(stream));
}

/* ------------------------------------------------------------------ layout */
int vc_ncthw_to_nthwc(const void* in, int32_t in_is_bf16, void* out, int32_t N, int32_t C, int64_t Sp, void* stream) {
    VC_REQUIRE(in && out && N > 0 && C > 0 && Sp > 0, "ncthw_to_nthwc: bad arguments");
    return vc_ew_ncs_to_nsc(in, in_is_bf16, out, N, C, Sp, S(stream));
}
int vc_nthwc_to_ncthw(const void* in, void* out, int32_t out_is_bf16, int32_t N, int32_t C, int64_t Sp, void* stream) {
    VC_REQUIRE(in && out && N > 0 && C > 0 && Sp > 0, "nthwc_to_ncthw: bad arguments");
    return vc_ew_nsc_to_ncs(in, out, out_is_bf16, N, C, Sp, S(stream));
}

}  // extern "C"
