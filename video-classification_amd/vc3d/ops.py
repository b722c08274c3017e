"""Tensor-level wrappers over the C ABI (include/vc3d.h).

Every wrapper checks device, dtype, contiguity and element counts on the host before the
launch: a kernel that reads past a buffer can take the whole GPU down.  torch is used for
device memory and the current stream only.
"""
import torch

from . import capi
from .capi import ConvDesc, PoolDesc, VC_STAT_REPLICAS, STEM_K, check, lib, ptr, cur_stream

BF16 = torch.bfloat16
F32 = torch.float32


class KernelTimer:
    """HIP-event timing of individual launches on the current stream (bench.py / profiling only).
    ``select`` is a predicate on the launch key (kind, shape...); None times every conv launch."""

    def __init__(self, select=None, key=None):
        self.select = select
        self.key = key  # the one launch a REPLAYED step times (engine command lists); select still decides how often
        self.records = {}  # key -> [flops_per_launch, [(start, end), ...]]

    def begin(self, key, flops):
        if self.select is not None and not self.select(key):
            return None
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        self.records.setdefault(key, [flops, []])[1].append((e0, e1))
        return e1

    def replay_pair(self, key, flops):
        """for a REPLAYED step (engine command lists): raw HIP events (start, stop) that vc_replay records around the
        command of launch `key`, or None when the key is not selected."""
        if self.select is None or not self.select(key):
            return None
        import ctypes
        evs = (ctypes.c_void_p * 2)()
        check(lib.vc_replay_events_create(2, 1, evs), "vc_replay_events_create")
        self.records.setdefault(key, [flops, []])[1].append((evs, None))
        return evs

    def summary(self):
        """{key: (launches, avg_ms, flops_per_launch)}; call after a device synchronize."""
        import ctypes
        out = {}
        for key, (flops, evs) in self.records.items():
            ms = []
            for a, b in evs:
                if b is None:  # a replay pair
                    v = ctypes.c_float()
                    check(lib.vc_replay_event_elapsed_ms(a[0], a[1], ctypes.byref(v)), "vc_replay_event_elapsed_ms")
                    lib.vc_replay_events_destroy(2, a)
                    ms.append(v.value)
                else:
                    ms.append(a.elapsed_time(b))
            out[key] = (len(ms), sum(ms) / max(len(ms), 1), flops)
        return out


TIMER = None  # set to a KernelTimer to time conv launches

# While the engine records a step (vc3d/replay.py) every tensor allocated here is appended to TAPE and so stays alive
# for as long as the recording does: the recorded commands name these buffers by address.
TAPE = None
WS_EPOCH = 0  # bumped whenever a split-K workspace is (re)allocated: recorded commands name it by address too


def _empty(*shape, dtype, device):
    t = torch.empty(*shape, dtype=dtype, device=device)
    if TAPE is not None:
        TAPE.append(t)
    return t


def _empty_like(ref):
    t = torch.empty_like(ref)
    if TAPE is not None:
        TAPE.append(t)
    return t


def zeros(n, dtype, device):
    """n zeroed elements by an explicit (recordable) memset on the current stream"""
    t = _empty(n, dtype=dtype, device=device)
    check(lib.vc_memset_async(t.data_ptr(), 0, t.numel() * t.element_size(), cur_stream()), "vc_memset_async")
    return t


def incr_i64(table, n):
    """*ptr += 1 for the n device pointers in `table` (int64 tensor on the GPU): BatchNorm's num_batches_tracked"""
    _chk(table, torch.int64, n, "pointer table")
    check(lib.vc_incr_i64(table.data_ptr(), n, cur_stream()), "vc_incr_i64")




def _conv_key(kind, d):
    return (kind, d.N, d.Ci, d.Ti, d.Hi, d.Wi, d.Co, d.kt, d.kh, d.kw, d.st, d.sh, d.sw)


def _conv_flops(d):
    To, Ho, Wo = d.out_dims()
    return 2.0 * d.N * To * Ho * Wo * d.Co * d.Ci * d.kt * d.kh * d.kw


def _chk(t, dtype, numel=None, name="tensor"):
    if t is None:
        return
    if not t.is_cuda:
        raise capi.Vc3dError("%s must be a CUDA/HIP tensor: the vc3d path has no CPU fallback" % name)
    if t.dtype != dtype:
        raise capi.Vc3dError("%s: expected %s, got %s" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise capi.Vc3dError("%s must be contiguous" % name)
    if numel is not None and t.numel() != numel:
        raise capi.Vc3dError("%s: expected %d elements, got %d" % (name, numel, t.numel()))


_WS = {}  # device -> split-K workspace (grown on demand; reuse is safe: launches are stream-ordered)
USE_WORKSPACE = True


def _workspace(d, op, device, need=None):
    """(ptr, bytes) of a workspace large enough for conv `d`, op 0/1/2; (None, 0) if none is needed."""
    if not USE_WORKSPACE:
        return None, 0
    if need is None:
        need = int(lib.vc_conv3d_workspace_bytes(d, op))
    if need == 0:
        return None, 0
    key = (device.type, device.index, torch.cuda.current_stream().cuda_stream)
    buf = _WS.get(key)
    if buf is None or buf.numel() < need:
        global WS_EPOCH
        WS_EPOCH += 1
        if capi.RECORDER is not None:  # commands recorded so far name the old buffer: this step cannot be kept
            capi.RECORDER.failed = "the split-K workspace grew while recording"
        buf = torch.empty(max(need, 1 << 20), dtype=torch.uint8, device=device)
        _WS[key] = buf
    return buf.data_ptr(), buf.numel()


def conv_desc(N, Ci, T, H, W, Co, k, s, p):
    k = (k,) * 3 if isinstance(k, int) else tuple(k)
    s = (s,) * 3 if isinstance(s, int) else tuple(s)
    p = (p,) * 3 if isinstance(p, int) else tuple(p)
    return ConvDesc(N, Ci, T, H, W, Co, *k, *s, *p)


def new_stats(C, device):
    return zeros(VC_STAT_REPLICAS * 2 * C, F32, device).view(VC_STAT_REPLICAS, 2, C)


# ------------------------------------------------------------------ conv
def pack_weight(w_master, want_dgrad=True):
    """w_master: fp32 parameter [Co][Ci][kt][kh][kw] stored channels_last_3d."""
    Co, Ci, kt, kh, kw = w_master.shape
    wm = w_master.detach().permute(0, 2, 3, 4, 1)  # physical [Co][kt][kh][kw][Ci]
    _chk(wm, F32, name="w_master (channels_last_3d)")
    taps = kt * kh * kw
    wf = _empty(Co, taps * Ci, dtype=BF16, device=wm.device)
    wd = _empty(Ci, taps * Co, dtype=BF16, device=wm.device) if want_dgrad else None
    check(lib.vc_pack_weight(ptr(wm), ptr(wf), ptr(wd), Co, taps, Ci, cur_stream()), "vc_pack_weight")
    return wf, wd


def bn_fin(ticket, count, gamma, beta, running_mean, running_var, momentum, eps, params=None, shift=None):
    """descriptor of a BatchNorm finalize folded into the conv launch that produces the sums (the last workgroup
    does bn_finalize's work).  `ticket`: capi.VC_TICKET_WORDS int32 words, zero before their first use (left zero by the launch).
    Returns (descriptor, params [4][C])."""
    Cn = gamma.numel()
    _chk(ticket, torch.int32, capi.VC_TICKET_WORDS, "ticket")
    for t, nm in ((gamma, "gamma"), (beta, "beta"), (running_mean, "running_mean"), (running_var, "running_var")):
        _chk(t, F32, Cn, nm)
    if params is None:
        params = _empty(4, Cn, dtype=F32, device=gamma.device)
    _chk(params, F32, 4 * Cn, "bn params")
    _chk(shift, F32, Cn, "bn shift")  # sums relative to it; receives this batch's mean (the next step's shift)
    f = capi.BnFin(ptr(ticket), ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var), ptr(params), float(count),
                   float(momentum), float(eps), ptr(shift), ptr(shift))
    f._keep = (ticket, gamma, beta, running_mean, running_var, params, shift)
    return f, params


def conv3d_fwd(d, x, wf, stats=None, out=None, fin=None, act_params=None, relu=True, residual=None):
    """act_params ([4][Co] BatchNorm params with fixed statistics, inference): y = relu(scale * conv + shift) in the
    conv's own epilogue (no separate bn_apply pass); not combinable with stats / fin."""
    To, Ho, Wo = d.out_dims()
    taps = d.kt * d.kh * d.kw
    _chk(x, BF16, d.N * d.Ti * d.Hi * d.Wi * d.Ci, "x")
    _chk(wf, BF16, d.Co * taps * d.Ci, "w")
    if out is None:
        out = _empty(d.N, To, Ho, Wo, d.Co, dtype=BF16, device=x.device)
    _chk(out, BF16, d.N * To * Ho * Wo * d.Co, "y")
    _chk(stats, F32, VC_STAT_REPLICAS * 2 * d.Co, "bn_stats")
    ev = TIMER.begin(_conv_key("fwd", d), _conv_flops(d)) if TIMER is not None else None
    if capi.RECORDER is not None:
        capi.RECORDER.tag = (_conv_key("fwd", d), _conv_flops(d))
    ws, wsb = _workspace(d, 0, x.device)
    if act_params is not None:
        if stats is not None or fin is not None:
            raise capi.Vc3dError("conv3d_fwd: act_params (inference) excludes batch statistics")
        _chk(act_params, F32, 4 * d.Co, "bn params")
        _chk(residual, BF16, out.numel(), "residual")
        check(lib.vc_conv3d_fwd_act(d, ptr(x), ptr(wf), ptr(out), ptr(act_params), ptr(residual), 1 if relu else 0, ws, wsb,
                                    cur_stream()), "vc_conv3d_fwd_act")
    elif fin is not None:
        check(lib.vc_conv3d_fwd_bn(d, ptr(x), ptr(wf), ptr(out), ptr(stats), fin, ws, wsb, cur_stream()),
              "vc_conv3d_fwd_bn")
    else:
        check(lib.vc_conv3d_fwd(d, ptr(x), ptr(wf), ptr(out), ptr(stats), ws, wsb, cur_stream()), "vc_conv3d_fwd")
    if ev is not None:
        ev.record()
    return out


def conv3d_dgrad(d, dy, wd, out=None, accumulate=False, bnred=None, acc_bits=None):
    """vc_conv3d_dgrad_fused's extras.  bnred = (y, params, relu_bits, sums, ticket, coef, dgamma, dbeta): the reduction
    of the BatchNorm backward that consumes dx (mask mode 3, or 2 when relu_bits is None) comes with the conv (dense
    geometries: from its epilogue); follow it with bn_bwd(..., coef=coef, reduced=True).  acc_bits (with accumulate):
    the existing dx counts only where its ReLU bit is set."""
    To, Ho, Wo = d.out_dims()
    taps = d.kt * d.kh * d.kw
    _chk(dy, BF16, d.N * To * Ho * Wo * d.Co, "dy")
    _chk(wd, BF16, d.Ci * taps * d.Co, "w_dgrad")
    if out is None:
        if accumulate:
            raise capi.Vc3dError("conv3d_dgrad: accumulate needs an existing dx")
        out = _empty(d.N, d.Ti, d.Hi, d.Wi, d.Ci, dtype=BF16, device=dy.device)
    _chk(out, BF16, d.N * d.Ti * d.Hi * d.Wi * d.Ci, "dx")
    ev = TIMER.begin(_conv_key("dgrad", d), _conv_flops(d)) if TIMER is not None else None
    if capi.RECORDER is not None:
        capi.RECORDER.tag = (_conv_key("dgrad", d), _conv_flops(d))
    ws, wsb = _workspace(d, 1, dy.device)
    if acc_bits is not None and not accumulate:
        raise capi.Vc3dError("conv3d_dgrad: acc_bits needs accumulate")
    _chk(acc_bits, torch.uint8, d.N * d.Ti * d.Hi * d.Wi * d.Ci // 8, "acc_bits")
    if bnred is not None or acc_bits is not None:
        ry, rp, rb, rs, rt, rc, rdg, rdb = bnred if bnred is not None else (None,) * 8
        n = d.N * d.Ti * d.Hi * d.Wi * d.Ci
        _chk(ry, BF16, n, "bnred y")
        _chk(rp, F32, 4 * d.Ci, "bnred params")
        _chk(rb, torch.uint8, n // 8, "bnred relu bits")
        _chk(rs, F32, VC_STAT_REPLICAS * 2 * d.Ci, "bnred sums")
        _chk(rt, torch.int32, capi.VC_TICKET_WORDS, "bnred ticket")
        _chk(rc, F32, 2 * d.Ci, "bnred coef")
        _chk(rdg, F32, d.Ci, "bnred dgamma")
        _chk(rdb, F32, d.Ci, "bnred dbeta")
        red = None if bnred is None else capi.BnBwdRed(ptr(ry), ptr(rp), ptr(rb), ptr(rs), ptr(rt), ptr(rc), ptr(rdg),
                                                       ptr(rdb))
        check(lib.vc_conv3d_dgrad_fused(d, ptr(dy), ptr(wd), ptr(out), 1 if accumulate else 0, ptr(acc_bits), red, ws, wsb,
                                        cur_stream()), "vc_conv3d_dgrad_fused")
    else:
        check(lib.vc_conv3d_dgrad(d, ptr(dy), ptr(wd), ptr(out), 1 if accumulate else 0, ws, wsb, cur_stream()),
              "vc_conv3d_dgrad")
    if ev is not None:
        ev.record()
    return out


def conv3d_wgrad(d, x, dy, dw):
    """dw: fp32 [Co][kt][kh][kw][Ci] (physical), accumulated into."""
    To, Ho, Wo = d.out_dims()
    taps = d.kt * d.kh * d.kw
    _chk(x, BF16, d.N * d.Ti * d.Hi * d.Wi * d.Ci, "x")
    _chk(dy, BF16, d.N * To * Ho * Wo * d.Co, "dy")
    _chk(dw, F32, d.Co * taps * d.Ci, "dw")
    ev = TIMER.begin(_conv_key("wgrad", d), _conv_flops(d)) if TIMER is not None else None
    if capi.RECORDER is not None:
        capi.RECORDER.tag = (_conv_key("wgrad", d), _conv_flops(d))
    ws, wsb = _workspace(d, 2, x.device)
    check(lib.vc_conv3d_wgrad(d, ptr(x), ptr(dy), ptr(dw), ws, wsb, cur_stream()), "vc_conv3d_wgrad")
    if ev is not None:
        ev.record()
    return dw


def sgd_step(params, grads, mom, shadow, lr, momentum, wd, nesterov):
    """fused torch.optim.SGD step over flat fp32 buffers; also writes the bf16 shadow."""
    n = params.numel()
    _chk(params, F32, n, "flat params")
    _chk(grads, F32, n, "flat grads")
    _chk(mom, F32, n, "momentum buffer")
    _chk(shadow, BF16, n, "bf16 shadow")
    check(lib.vc_sgd_step(ptr(params), ptr(grads), ptr(mom), ptr(shadow), n, float(lr), float(momentum), float(wd),
                          1 if nesterov else 0, cur_stream()), "vc_sgd_step")


def adam_step(params, grads, exp_avg, exp_avg_sq, shadow, lr, beta1, beta2, eps, wd, step):
    """fused torch.optim.Adam step over flat fp32 buffers (step counts from 1); also writes the bf16 shadow."""
    n = params.numel()
    for t, nm in ((params, "params"), (grads, "grads"), (exp_avg, "exp_avg"), (exp_avg_sq, "exp_avg_sq")):
        _chk(t, F32, n, nm)
    _chk(shadow, BF16, n, "shadow")
    check(lib.vc_adam_step(ptr(params), ptr(grads), ptr(exp_avg), ptr(exp_avg_sq), ptr(shadow), n, float(lr), float(beta1),
                           float(beta2), float(eps), float(wd), int(step), cur_stream()), "vc_adam_step")


def shadow_bf16(params, shadow):
    _chk(params, F32, name="flat params")
    _chk(shadow, BF16, params.numel(), "bf16 shadow")
    check(lib.vc_shadow_bf16(ptr(params), ptr(shadow), params.numel(), cur_stream()), "vc_shadow_bf16")


def pack_dgrad_batched(shadow, wd_flat, table_dev, table_host):
    _chk(shadow, BF16, name="bf16 shadow")
    _chk(wd_flat, BF16, name="dgrad weights")
    _chk(table_dev, torch.int32, table_host.numel(), "table")
    if table_host.is_cuda or table_host.dtype != torch.int32 or not table_host.is_contiguous():
        raise capi.Vc3dError("pack_dgrad_batched: host table must be a contiguous CPU int32 tensor")
    th = table_host.view(-1, 6)
    for src, dst, Co, taps, Ci, _ in th.tolist():  # the kernel trusts these offsets
        if src + Co * taps * Ci > shadow.numel() or dst + Co * taps * Ci > wd_flat.numel():
            raise capi.Vc3dError("pack_dgrad_batched: table entry exceeds its buffer")
    check(lib.vc_pack_dgrad_batched(ptr(shadow), ptr(wd_flat), ptr(table_dev), table_host.data_ptr(), th.shape[0],
                                    cur_stream()), "vc_pack_dgrad_batched")


# ------------------------------------------------------------------ stem
def stem_wp(W):
    """row pitch (positions) of the packed stem source: W/2 + 3 rounded up to even."""
    return (W // 2 + 4) & ~1


def stem_pack_input(x):
    """x: [N][3][T][H][W] fp32 or bf16 (contiguous) -> padded space-to-depth bf16 source."""
    N, Cc, T, H, W = x.shape
    if Cc != 3:
        raise capi.Vc3dError("stem expects 3 input channels, got %d" % Cc)
    if x.dtype not in (F32, BF16):
        raise capi.Vc3dError("stem input must be fp32 or bf16")
    _chk(x, x.dtype, name="clip")
    xs = _empty(N, T + 6, H // 2 + 3, stem_wp(W), 12, dtype=BF16, device=x.device)
    check(lib.vc_stem_pack_input(ptr(x), 1 if x.dtype == BF16 else 0, ptr(xs), N, T, H, W, cur_stream()),
          "vc_stem_pack_input")
    return xs


class PackedClips:
    """a clip batch already in the stem's packed space-to-depth form (what stem_pack_input makes): the engine takes
    it in place of the N x 3 x T x H x W tensor and skips its own packing pass."""

    def __init__(self, xs, N, T, H, W):
        self.xs, self.N, self.T, self.H, self.W = xs, N, T, H, W
        self.shape = (N, 3, T, H, W)
        self.device = xs.device
        self.is_cuda = xs.is_cuda
        self.requires_grad = False

    def dim(self):
        return 5


def clip_pack_u8(frames, crops, T, H, W, tstep=1, mean=(0.0, 0.0, 0.0), std=(1.0, 1.0, 1.0)):
    """frames: uint8 [N][Ts][Hs][Ws][3] on the GPU; crops: int32 [N][4] = (t0, y0, x0, flip) on the HOST.
    Temporal sampling + crop + flip + normalisation + bf16 + stem packing in one pass -> PackedClips."""
    if frames.dim() != 5 or frames.shape[-1] != 3:
        raise capi.Vc3dError("clip_pack_u8: expected uint8 frames N x Ts x Hs x Ws x 3, got %s" % (tuple(frames.shape),))
    _chk(frames, torch.uint8, name="frames")
    N, Ts, Hs, Ws, _ = frames.shape
    if crops.is_cuda or crops.dtype != torch.int32 or tuple(crops.shape) != (N, 4) or not crops.is_contiguous():
        raise capi.Vc3dError("clip_pack_u8: crops must be a contiguous int32 host tensor of shape N x 4")
    crops_dev = crops.to(frames.device, non_blocking=True)
    xs = _empty(N, T + 6, H // 2 + 3, stem_wp(W), 12, dtype=BF16, device=frames.device)
    m = (capi.C.c_float * 3)(*[float(v) for v in mean])
    s = (capi.C.c_float * 3)(*[float(v) for v in std])
    check(lib.vc_clip_pack_u8(ptr(frames), ptr(crops_dev), crops.data_ptr(), ptr(xs), N, T, H, W, Ts, Hs, Ws, int(tstep),
                              m, s, cur_stream()), "vc_clip_pack_u8")
    return PackedClips(xs, N, T, H, W)


def stem_pack_weight(w_master):
    Co = w_master.shape[0]
    if tuple(w_master.shape[1:]) != (3, 7, 7, 7):
        raise capi.Vc3dError("stem weight must be [Co][3][7][7][7]")
    wm = w_master.detach().permute(0, 2, 3, 4, 1)
    _chk(wm, F32, name="stem w_master (channels_last_3d)")
    ws = _empty(Co, STEM_K, dtype=BF16, device=wm.device)
    check(lib.vc_stem_pack_weight(ptr(wm), ptr(ws), Co, cur_stream()), "vc_stem_pack_weight")
    return ws


def stem_fwd(xs, ws, N, T, H, W, Co, stats=None, out=None, fin=None):
    _chk(xs, BF16, N * (T + 6) * (H // 2 + 3) * stem_wp(W) * 12, "xs")
    _chk(ws, BF16, Co * STEM_K, "ws")
    if out is None:
        out = _empty(N, T, H // 2, W // 2, Co, dtype=BF16, device=xs.device)
    _chk(out, BF16, N * T * (H // 2) * (W // 2) * Co, "y")
    _chk(stats, F32, VC_STAT_REPLICAS * 2 * Co, "bn_stats")
    flops = 2.0 * N * T * (H // 2) * (W // 2) * Co * 1029
    ev = TIMER.begin(("stem_fwd", N, T, H, W, Co), flops) if TIMER is not None else None
    if capi.RECORDER is not None:
        capi.RECORDER.tag = (("stem_fwd", N, T, H, W, Co), flops)
    if fin is not None:
        check(lib.vc_stem_fwd_bn(ptr(xs), ptr(ws), ptr(out), ptr(stats), fin, N, T, H, W, Co, cur_stream()),
              "vc_stem_fwd_bn")
    else:
        check(lib.vc_stem_fwd(ptr(xs), ptr(ws), ptr(out), ptr(stats), N, T, H, W, Co, cur_stream()), "vc_stem_fwd")
    if ev is not None:
        ev.record()
    return out


def stem_wgrad(xs, dy, N, T, H, W, Co, dw_master_phys, scratch=None):
    """dw_master_phys: fp32 [Co][7][7][7][3] physical view of the stem gradient, accumulated into."""
    _chk(xs, BF16, N * (T + 6) * (H // 2 + 3) * stem_wp(W) * 12, "xs")
    _chk(dy, BF16, N * T * (H // 2) * (W // 2) * Co, "dy")
    _chk(dw_master_phys, F32, Co * 1029, "dw")
    if scratch is None:
        scratch = torch.zeros(Co, STEM_K, dtype=F32, device=xs.device)
    else:
        _chk(scratch, F32, Co * STEM_K, "stem wgrad scratch")
        # (an explicit memset call, not scratch.zero_(): a recorded step replays entry points of the library only)
        check(lib.vc_memset_async(scratch.data_ptr(), 0, scratch.numel() * 4, cur_stream()), "vc_memset_async")
    flops = 2.0 * N * T * (H // 2) * (W // 2) * Co * 1029
    ev = TIMER.begin(("stem_wgrad", N, T, H, W, Co), flops) if TIMER is not None else None
    if capi.RECORDER is not None:
        capi.RECORDER.tag = (("stem_wgrad", N, T, H, W, Co), flops)
    ws, wsb = _workspace(None, 2, xs.device, need=int(lib.vc_stem_wgrad_workspace_bytes(N, T, H, W, Co)))
    check(lib.vc_stem_wgrad(ptr(xs), ptr(dy), ptr(scratch), N, T, H, W, Co, ws, wsb, cur_stream()), "vc_stem_wgrad")
    if ev is not None:
        ev.record()
    check(lib.vc_stem_unpack_wgrad(ptr(scratch), ptr(dw_master_phys), Co, cur_stream()), "vc_stem_unpack_wgrad")
    return dw_master_phys


# ------------------------------------------------------------------ batch norm
def bn_stats(y, C, sums=None, shift=None):
    """shift ([C] fp32, optional): the kernel takes the sums relative to row 0 of y and writes that row there; hand the
    same tensor to bn_finalize (robust against |mean| >> sigma)."""
    _chk(y, BF16, name="y")
    rows = y.numel() // C
    if sums is None:
        sums = new_stats(C, y.device)
    _chk(sums, F32, VC_STAT_REPLICAS * 2 * C, "sums")
    _chk(shift, F32, C, "bn shift")
    check(lib.vc_bn_stats(ptr(y), ptr(sums), ptr(shift), rows, C, cur_stream()), "vc_bn_stats")
    return sums


def bn_finalize(stats, count, gamma, beta, running_mean, running_var, momentum, eps, params=None, shift=None,
                shift_next=None):
    Cn = gamma.numel()
    _chk(stats, F32, VC_STAT_REPLICAS * 2 * Cn, "stats")
    for t, nm in ((gamma, "gamma"), (beta, "beta"), (running_mean, "running_mean"), (running_var, "running_var")):
        _chk(t, F32, Cn, nm)
    if params is None:
        params = _empty(4, Cn, dtype=F32, device=gamma.device)
    _chk(params, F32, 4 * Cn, "bn params")
    _chk(shift, F32, Cn, "bn shift")
    _chk(shift_next, F32, Cn, "bn shift_next")
    check(lib.vc_bn_finalize(ptr(stats), Cn, float(count), ptr(gamma), ptr(beta), ptr(running_mean),
                             ptr(running_var), float(momentum), float(eps), ptr(params), ptr(shift), ptr(shift_next),
                             cur_stream()), "vc_bn_finalize")
    return params


def bn_eval_params(gamma, beta, running_mean, running_var, eps, params=None):
    Cn = gamma.numel()
    for t, nm in ((gamma, "gamma"), (beta, "beta"), (running_mean, "running_mean"), (running_var, "running_var")):
        _chk(t, F32, Cn, nm)
    if params is None:
        params = _empty(4, Cn, dtype=F32, device=gamma.device)
    _chk(params, F32, 4 * Cn, "bn params")
    check(lib.vc_bn_eval_params(Cn, ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var), float(eps),
                                ptr(params), cur_stream()), "vc_bn_eval_params")
    return params


def bn_apply(y, params, C, residual=None, relu=True, out=None, relu_bits=None):
    """relu_bits (uint8, y.numel() / 8 bytes; needs relu): also receives the ReLU mask of `out`, one bit per element --
    what bn_bwd(mask=3, aux=relu_bits) reads in place of the activation."""
    _chk(y, BF16, name="y")
    _chk(params, F32, 4 * C, "bn params")
    _chk(residual, BF16, y.numel(), "residual")
    if out is None:
        out = _empty_like(y)
    _chk(out, BF16, y.numel(), "out")
    if relu_bits is not None:
        if not relu:
            raise capi.Vc3dError("bn_apply: relu_bits needs relu=True")
        _chk(relu_bits, torch.uint8, y.numel() // 8, "relu_bits")
        check(lib.vc_bn_apply_mask(ptr(y), ptr(params), ptr(residual), ptr(out), ptr(relu_bits), y.numel(), C,
                                   cur_stream()), "vc_bn_apply_mask")
        return out
    check(lib.vc_bn_apply(ptr(y), ptr(params), ptr(residual), ptr(out), y.numel(), C, 1 if relu else 0,
                          cur_stream()), "vc_bn_apply")
    return out


def bn_bwd(y, dz, params, C, gamma_grad, beta_grad, mask=0, aux=None, want_g=False, sums=None, coef=None,
           dy=None, g_out=None, ticket=None, reduced=False, frozen=False):
    """BatchNorm backward (batch statistics).  Returns (dy, g) with g = masked dz or None.
    `frozen`: the forward normalised with the running statistics (eval mode), which do not depend on the batch: the
    mean terms of the backward vanish, dy = scale * g.  The reduction still makes dgamma / dbeta; its coef is zeroed
    before the apply pass (one small fill: this is the fine-tuning-with-frozen-statistics path, not the hot one).
    `ticket` (capi.VC_TICKET_WORDS zero int32 words): the reduction's last workgroup does the finalize (one launch less).
    `reduced`: coef, dgamma and dbeta are already there (conv3d_dgrad's bnred made them with dz); only the apply pass runs."""
    n = y.numel()
    _chk(y, BF16, name="y")
    _chk(dz, BF16, n, "dz")
    if mask == 3:
        _chk(aux, torch.uint8, n // 8, "aux (relu bits)")
    else:
        _chk(aux, BF16, n, "aux")
    _chk(params, F32, 4 * C, "bn params")
    _chk(gamma_grad, F32, C, "dgamma")
    _chk(beta_grad, F32, C, "dbeta")
    rows = n // C
    if reduced:
        if coef is None:
            raise capi.Vc3dError("bn_bwd: reduced=True needs the coef the reduction produced")
    elif sums is None:
        sums = new_stats(C, y.device)
    _chk(sums, F32, VC_STAT_REPLICAS * 2 * C, "sums")
    if coef is None:
        coef = _empty(2, C, dtype=F32, device=y.device)
    _chk(coef, F32, 2 * C, "coef")
    if dy is None:
        dy = _empty_like(y)
    _chk(dy, BF16, n, "dy")
    if want_g and g_out is None:
        g_out = _empty_like(y)
    _chk(g_out, BF16, n, "g_out")
    s = cur_stream()
    if reduced:
        pass
    elif ticket is not None:
        _chk(ticket, torch.int32, capi.VC_TICKET_WORDS, "ticket")
        check(lib.vc_bn_bwd_reduce_fin(ptr(y), ptr(dz), ptr(aux), ptr(params), ptr(sums), rows, C, mask, ptr(ticket),
                                       ptr(coef), ptr(gamma_grad), ptr(beta_grad), s), "vc_bn_bwd_reduce_fin")
    else:
        check(lib.vc_bn_bwd_reduce(ptr(y), ptr(dz), ptr(aux), ptr(params), ptr(sums), rows, C, mask, s),
              "vc_bn_bwd_reduce")
        check(lib.vc_bn_bwd_finalize(ptr(sums), C, float(rows), ptr(coef), ptr(gamma_grad), ptr(beta_grad), s),
              "vc_bn_bwd_finalize")
    if frozen:
        if reduced:
            raise capi.Vc3dError("bn_bwd: frozen statistics and a folded reduction do not go together")
        coef.zero_()
    check(lib.vc_bn_bwd_apply(ptr(y), ptr(dz), ptr(aux), ptr(params), ptr(coef), ptr(dy), ptr(g_out), n, C, mask, s),
          "vc_bn_bwd_apply")
    return dy, g_out


def bn_bwd_pooled(pd, y, dpooled, idx, params, C, gamma_grad, beta_grad, sums=None, pooled=None, ticket=None,
                  frozen=False):
    """BN+ReLU backward of the stem straight from the max-pool output gradient (no materialised
    pool-input gradient).  Returns dy (gradient of the conv output).  `pooled` = the pool's output
    (maxpool3d_fwd with the same params): lets the reduction skip its element-wise gather of y."""
    To, Ho, Wo = pd.out_dims()
    n = pd.N * pd.T * pd.H * pd.W * pd.C
    _chk(y, BF16, n, "y")
    _chk(dpooled, BF16, pd.N * To * Ho * Wo * pd.C, "dpooled")
    _chk(idx, torch.uint8, pd.N * To * Ho * Wo * pd.C, "argmax")
    _chk(pooled, BF16, pd.N * To * Ho * Wo * pd.C, "pooled")
    _chk(params, F32, 4 * C, "bn params")
    _chk(gamma_grad, F32, C, "dgamma")
    _chk(beta_grad, F32, C, "dbeta")
    if C != pd.C:
        raise capi.Vc3dError("bn_bwd_pooled: channel mismatch")
    if sums is None:
        sums = new_stats(C, y.device)
    _chk(sums, F32, VC_STAT_REPLICAS * 2 * C, "sums")
    coef = _empty(2, C, dtype=F32, device=y.device)
    dy = _empty_like(y)
    s = cur_stream()
    if ticket is not None:
        _chk(ticket, torch.int32, capi.VC_TICKET_WORDS, "ticket")
        check(lib.vc_bn_bwd_reduce_pooled_fin(pd, ptr(y), ptr(dpooled), ptr(idx), ptr(pooled), ptr(params), ptr(sums),
                                              ptr(ticket), ptr(coef), ptr(gamma_grad), ptr(beta_grad), s),
              "vc_bn_bwd_reduce_pooled_fin")
    else:
        check(lib.vc_bn_bwd_reduce_pooled(pd, ptr(y), ptr(dpooled), ptr(idx), ptr(pooled), ptr(params), ptr(sums), s),
              "vc_bn_bwd_reduce_pooled")
        check(lib.vc_bn_bwd_finalize(ptr(sums), C, float(n // C), ptr(coef), ptr(gamma_grad), ptr(beta_grad), s),
              "vc_bn_bwd_finalize")
    if frozen:  # see bn_bwd
        coef.zero_()
    check(lib.vc_bn_bwd_apply_pooled(pd, ptr(y), ptr(dpooled), ptr(idx), ptr(params), ptr(coef), ptr(dy), s),
          "vc_bn_bwd_apply_pooled")
    return dy


def add(a, b, out=None):
    _chk(a, BF16, name="a")
    _chk(b, BF16, a.numel(), "b")
    if out is None:
        out = _empty_like(a)
    _chk(out, BF16, a.numel(), "out")
    check(lib.vc_add(ptr(a), ptr(b), ptr(out), a.numel(), cur_stream()), "vc_add")
    return out


# ------------------------------------------------------------------ pooling / head / layout
def pool_desc(N, C, T, H, W, k=3, s=2, p=1):
    k = (k,) * 3 if isinstance(k, int) else tuple(k)
    s = (s,) * 3 if isinstance(s, int) else tuple(s)
    p = (p,) * 3 if isinstance(p, int) else tuple(p)
    return PoolDesc(N, C, T, H, W, *k, *s, *p)


def maxpool3d_fwd(d, y, bn_params=None):
    To, Ho, Wo = d.out_dims()
    _chk(y, BF16, d.N * d.T * d.H * d.W * d.C, "y")
    _chk(bn_params, F32, 4 * d.C, "bn params")
    out = _empty(d.N, To, Ho, Wo, d.C, dtype=BF16, device=y.device)
    idx = _empty(d.N, To, Ho, Wo, d.C, dtype=torch.uint8, device=y.device)
    check(lib.vc_maxpool3d_fwd(d, ptr(y), ptr(bn_params), ptr(out), ptr(idx), cur_stream()), "vc_maxpool3d_fwd")
    return out, idx


def maxpool3d_bwd(d, dout, idx):
    To, Ho, Wo = d.out_dims()
    _chk(dout, BF16, d.N * To * Ho * Wo * d.C, "dout")
    _chk(idx, torch.uint8, d.N * To * Ho * Wo * d.C, "argmax")
    din = _empty(d.N, d.T, d.H, d.W, d.C, dtype=BF16, device=dout.device)
    check(lib.vc_maxpool3d_bwd(d, ptr(dout), ptr(idx), ptr(din), cur_stream()), "vc_maxpool3d_bwd")
    return din


def gavgpool_fwd(x, N, S, Cn):
    _chk(x, BF16, N * S * Cn, "x")
    out = _empty(N, Cn, dtype=F32, device=x.device)
    check(lib.vc_gavgpool_fwd(ptr(x), ptr(out), N, S, Cn, cur_stream()), "vc_gavgpool_fwd")
    return out


def gavgpool_bwd(dfeat, N, S, Cn, shape):
    _chk(dfeat, F32, N * Cn, "dfeat")
    din = _empty(shape, dtype=BF16, device=dfeat.device)
    _chk(din, BF16, N * S * Cn, "din")
    check(lib.vc_gavgpool_bwd(ptr(dfeat), ptr(din), N, S, Cn, cur_stream()), "vc_gavgpool_bwd")
    return din


def softmax_ce(logits, labels, want_grad=True):
    """mean cross-entropy of fp32 logits [N][K] against int64 labels [N] and its gradient (softmax - onehot) / N, one
    launch (vc_softmax_ce) -> (loss [1], dlogits [N][K] or None)"""
    N, K = logits.shape
    _chk(logits, F32, N * K, "logits")
    _chk(labels, torch.int64, N, "labels")
    loss = _empty(1, dtype=F32, device=logits.device)
    dlog = _empty(N, K, dtype=F32, device=logits.device) if want_grad else None
    check(lib.vc_softmax_ce(ptr(logits), ptr(labels), ptr(loss), ptr(dlog), N, K, cur_stream()), "vc_softmax_ce")
    return loss, dlog


def linear_fwd(feat, W, bias):
    N, Fd = feat.shape
    K = W.shape[0]
    _chk(feat, F32, name="feat")
    _chk(W, F32, K * Fd, "W")
    _chk(bias, F32, K, "bias")
    out = _empty(N, K, dtype=F32, device=feat.device)
    check(lib.vc_linear_fwd(ptr(feat), ptr(W), ptr(bias), ptr(out), N, Fd, K, cur_stream()), "vc_linear_fwd")
    return out


def linear_bwd(dlogits, feat, W, dW=None, db=None, want_dfeat=True):
    N, K = dlogits.shape
    Fd = W.shape[1]
    _chk(dlogits, F32, name="dlogits")
    _chk(feat, F32, N * Fd, "feat")
    _chk(W, F32, K * Fd, "W")
    _chk(dW, F32, K * Fd, "dW")
    _chk(db, F32, K, "db")
    dfeat = _empty(N, Fd, dtype=F32, device=dlogits.device) if want_dfeat else None
    check(lib.vc_linear_bwd(ptr(dlogits), ptr(feat), ptr(W), ptr(dfeat), ptr(dW), ptr(db), N, Fd, K, cur_stream()),
          "vc_linear_bwd")
    return dfeat


def ncthw_to_nthwc(x):
    N, Cc = x.shape[0], x.shape[1]
    S = x.numel() // (N * Cc)
    if x.dtype not in (F32, BF16):
        raise capi.Vc3dError("layout: fp32 or bf16 only")
    _chk(x, x.dtype, name="x")
    out = _empty((N,) + tuple(x.shape[2:]) + (Cc,), dtype=BF16, device=x.device)
    check(lib.vc_ncthw_to_nthwc(ptr(x), 1 if x.dtype == BF16 else 0, ptr(out), N, Cc, S, cur_stream()),
          "vc_ncthw_to_nthwc")
    return out


def nthwc_to_ncthw(x, dtype=F32):
    N, Cc = x.shape[0], x.shape[-1]
    S = x.numel() // (N * Cc)
    _chk(x, BF16, name="x")
    out = _empty((N, Cc) + tuple(x.shape[1:-1]), dtype=dtype, device=x.device)
    check(lib.vc_nthwc_to_ncthw(ptr(x), ptr(out), 1 if dtype == BF16 else 0, N, Cc, S, cur_stream()),
          "vc_nthwc_to_ncthw")
    return out


def set_deterministic(on):
    """vc_set_deterministic: BatchNorm sums accumulated as exact 64-bit fixed-point integers instead of float atomics
    (include/vc3d.h): a step's results no longer depend on the order in which workgroups finish.  Process-wide; also
    VC_DETERMINISTIC=1 in the environment.  Returns the previous setting."""
    prev = bool(lib.vc_get_deterministic())
    check(lib.vc_set_deterministic(1 if on else 0), "vc_set_deterministic")
    return prev
