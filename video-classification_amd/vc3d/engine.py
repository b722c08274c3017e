"""Forward/backward executor of the 3D-ResNet backbone on the vc3d HIP kernels.

The nn.Module tree (lib/models/backbone.py) only holds parameters and buffers under the
reference's names.  This executor walks that tree and issues the kernel sequence itself --
forward, and a hand-written backward in reverse order -- instead of recording one autograd
node per op.  That keeps every activation in channels-last bf16, lets the BatchNorm
statistics ride in the conv epilogue, and tells the data-parallel layer exactly when each
gradient is final so its bucket can be all-reduced while the rest of backward runs.

Gradients are written into one flat fp32 buffer (reverse layer order); ``param.grad`` are
views of it with the parameter's own strides.
"""
import os
import weakref

import torch

from . import capi, ops
from .capi import VC_STAT_REPLICAS, VC_TICKET_WORDS, Vc3dError
from .replay import Recorder

BF16 = torch.bfloat16
F32 = torch.float32
FUSED_EVAL_ACT = os.environ.get("VC_NO_EVAL_ACT") is None  # inference: BatchNorm + ReLU in the conv epilogue
FUSED_BN_FIN = os.environ.get("VC_NO_BN_FIN") is None  # BatchNorm finalize done by the last workgroup of the producer
SIDE_WGRAD = os.environ.get("VC_WGRAD_SIDE_STREAM", "1") != "0"  # weight gradients on a side stream (engine._wgrad)
# the BatchNorm backward reduction of a block's last BN rides along with the data-gradient launch that completes the
# gradient it reduces (vc_conv3d_dgrad_fused; VC_NO_BNRED_FOLD=1 in the library runs the two launches instead)
FOLD_BN_RED = os.environ.get("VC_NO_BNRED") is None
RELU_BITS = os.environ.get("VC_NO_RELU_BITS") is None  # residual blocks save their ReLU mask as bits (VC_NO_RELU_BITS=1: read the activation instead)
SIDE_DS = os.environ.get("VC_NO_SIDE_DS") is None  # a block's downsample branch runs beside its main branch (forward)
DS_LAST = os.environ.get("VC_NO_DS_LAST") is None  # strided downsample's data gradient after the block's first conv's
# a training step's launch sequence is recorded once and replayed by one C call per forward and per backward
# (vc3d/replay.py, csrc/replay.hip); VC_NO_REPLAY=1 keeps every step on the eager path
REPLAY = os.environ.get("VC_NO_REPLAY") is None
FIRST_STATS = os.environ.get("VC_FIRST_STATS") is not None  # stand-alone row-0-shifted statistics on a layer's first step


def _triple(v):
    return tuple(v) if isinstance(v, (tuple, list)) else (v, v, v)


class _TapeStep:
    """what a recorded forward hands to autograd in place of the saved activations: which step of which recording"""
    __slots__ = ("tape", "serial", "__weakref__")

    def __init__(self, tape, serial):
        self.tape, self.serial = tape, serial


class _ConvBN:
    """one conv + its BatchNorm: geometry cache, packed weights, saved tensors."""

    def __init__(self, conv, bn, name):
        self.conv, self.bn, self.name = conv, bn, name
        self.k, self.s, self.p = _triple(conv.kernel_size), _triple(conv.stride), _triple(conv.padding)
        self.Ci, self.Co = conv.in_channels, conv.out_channels
        self.wf = self.wd = None
        self.wver = -1
        self.eval_params = None
        self.eval_ver = None
        self.shift = None         # [Co] fp32: what the BatchNorm sums are taken relative to = the previous batch mean
        self.shift_valid = False  # False until a first robust estimate exists (engine._first_stats)

    def desc(self, x):
        N, T, H, W, C = x.shape
        if C != self.Ci:
            raise Vc3dError("%s: expected %d input channels, got %d" % (self.name, self.Ci, C))
        return ops.conv_desc(N, C, T, H, W, self.Co, self.k, self.s, self.p)

    def weights(self):
        """(forward / wgrad operand, dgrad operand), repacked when the master changed.  A fused
        optimizer step (vc3d/optim.py) refreshes both itself and records the version it packed."""
        w = self.conv.weight
        if self.wver != w._version or self.wf is None or self.wf.device != w.device:
            self.wf, self.wd = ops.pack_weight(w, want_dgrad=True)
            self.wver = w._version
        return self.wf, self.wd


class R3DEngine:
    def __init__(self, backbone):
        self.bb = backbone
        self.stem = _ConvBN(backbone.conv1, backbone.bn1, "conv1")
        if (self.stem.k, self.stem.s, self.stem.p, self.stem.Ci) != ((7, 7, 7), (1, 2, 2), (3, 3, 3), 3):
            raise Vc3dError("stem must be 7x7x7 stride (1,2,2) pad 3 on 3 channels")
        self.stem_ws = None
        self.stem_wver = -1
        self.pool = (_triple(backbone.maxpool.kernel_size), _triple(backbone.maxpool.stride),
                     _triple(backbone.maxpool.padding))
        self.blocks = []
        for li in range(1, 5):
            layer = getattr(backbone, "layer%d" % li)
            for bi, blk in enumerate(layer):
                pre = "layer%d.%d" % (li, bi)
                cbs = [_ConvBN(blk.conv1, blk.bn1, pre + ".conv1"), _ConvBN(blk.conv2, blk.bn2, pre + ".conv2")]
                if hasattr(blk, "conv3"):
                    cbs.append(_ConvBN(blk.conv3, blk.bn3, pre + ".conv3"))
                ds = None
                if blk.downsample is not None:
                    ds = _ConvBN(blk.downsample[0], blk.downsample[1], pre + ".downsample")
                self.blocks.append((cbs, ds))
        self.flat_grad = None
        self.flat_param = None   # fp32 master of every parameter, same order / offsets as flat_grad
        self.flat_shadow = None  # bf16 copy of flat_param (conv slices = forward / wgrad operands)
        self.flat_wd = None      # bf16 dgrad operands of every conv, packed by one launch
        self.param_epoch = 0     # bumped by a fused optimizer step (memory changed behind torch's back)
        self.stat_epoch = 0      # bumped by every training forward: running statistics are updated by raw pointer
        self.grad_views = None
        self.grad_ready_cb = None  # set by the data-parallel wrapper: cb(param)
        self._side = None          # side stream of the weight-gradient launches (see _wgrad)
        self._main = None          # the stream backward was issued on (what a bucket all-reduce must also wait for)
        self.backward_done_cb = None
        # set by a fused optimizer for ONE backward (optim.arm_backward): hook(params, stream or None) -> True when it
        # launched a slice of its step on `stream`; not used together with grad_ready_cb.  opt_owner: that optimizer
        self.opt_hook = None
        self.opt_owner = None
        self._zero_ev = None  # recorded behind a clear of flat_grad on the side stream (optim.zero_grad)
        self._bwd_serial = 0  # backward passes run so far
        self._stem_scratch = None
        self._tickets = None       # zero int32 words of the fused BatchNorm finalize (see _ticket)
        self._ticket_ix = None
        self.trace = None  # tests set this to {} to record every intermediate tensor per layer
        self._tape = None  # the recorded training step (command lists + the tensors they name), see _forward_recorded
        # BatchNorm sums are taken relative to the previous batch mean (cb.shift).  On a layer's FIRST training step no
        # such estimate exists: by default the fused sums run with a zero shift (the textbook one-pass form on the fp32
        # accumulators: exact to 1e-5 for |mean| <= 10 sigma, and a bf16 tensor with |mean| = 100 sigma has lost its
        # content to storage rounding anyway); True (VC_FIRST_STATS=1): that one step takes a stand-alone pass over the
        # stored bf16 tensor relative to its row 0 instead (robust at any ratio, but it sees the rounded values)
        self.first_stats = FIRST_STATS

    def _t(self, name, **kw):
        if self.trace is not None:
            self.trace.setdefault(name, {}).update(kw)

    # ------------------------------------------------------------------ parameters
    def all_convbn(self):
        yield self.stem
        for cbs, ds in self.blocks:
            for cb in cbs:
                yield cb
            if ds is not None:
                yield ds

    def params_backward_order(self):
        """parameters in the order their gradients become final during backward."""
        cached = getattr(self, "_pbo", None)
        if cached is not None:  # the module tree (and so the order) is fixed once the engine exists
            return cached
        out = []
        for cbs, ds in reversed(self.blocks):
            seq = list(reversed(cbs))
            # block backward: last bn/conv first, downsample, then first conv
            for cb in seq[:-1]:
                out += [cb.bn.weight, cb.bn.bias, cb.conv.weight]
            if ds is not None:
                out += [ds.bn.weight, ds.bn.bias, ds.conv.weight]
            out += [seq[-1].bn.weight, seq[-1].bn.bias, seq[-1].conv.weight]
        out += [self.stem.bn.weight, self.stem.bn.bias, self.stem.conv.weight]
        self._pbo = out
        return out

    def setup_grads(self, extra_first=()):
        """(re)create the flat gradient buffer; ``extra_first`` are parameters (the head's) whose
        gradients come before the backbone's."""
        plist = list(extra_first) + self.params_backward_order()
        dev = plist[0].device
        pad = lambda n: (n + 63) // 64 * 64  # every parameter starts on a 256-byte (fp32) / 128-byte (bf16) line
        total = sum(pad(p.numel()) for p in plist)
        self.flat_grad = torch.zeros(total, dtype=F32, device=dev)
        self.flat_param = torch.zeros(total, dtype=F32, device=dev)
        self.grad_views = {}
        self.grad_ranges = []
        self.param_off = {}
        off = 0
        for p in plist:
            n = p.numel()
            seg = self.flat_grad[off:off + n]
            pseg = self.flat_param[off:off + n]
            if p.dim() == 5:
                Co, Ci, kt, kh, kw = p.shape
                v = seg.view(Co, kt, kh, kw, Ci).permute(0, 4, 1, 2, 3)
                pv = pseg.view(Co, kt, kh, kw, Ci).permute(0, 4, 1, 2, 3)
            else:
                v = seg.view(p.shape)
                pv = pseg.view(p.shape)
            # the parameter becomes a view of the flat master (conv weights: channels_last_3d strides)
            with torch.no_grad():
                pv.copy_(p.data)
                p.data = pv
            self.grad_views[id(p)] = v
            self.grad_ranges.append((p, off, n))
            self.param_off[id(p)] = off
            off += pad(n)
        self.plist = plist
        self.flat_shadow = None
        self.flat_wd = None
        return self.flat_grad

    def setup_flat_weights(self):
        """bf16 shadow of the flat master + the dgrad operands of every conv (one buffer, one table)."""
        dev = self.flat_param.device
        self.flat_shadow = torch.empty(self.flat_param.numel(), dtype=BF16, device=dev)
        rows, off_d, tiles = [], 0, 0
        self.wd_ranges = []
        for cb in self.all_convbn():
            if cb is self.stem:
                continue
            w = cb.conv.weight
            taps = cb.k[0] * cb.k[1] * cb.k[2]
            rows.append([self.param_off[id(w)], off_d, cb.Co, taps, cb.Ci, tiles])
            self.wd_ranges.append((cb, self.param_off[id(w)], off_d, w.numel()))
            off_d += w.numel()
            tiles += taps * (cb.Co // 64) * (cb.Ci // 64)
        self.flat_wd = torch.empty(off_d, dtype=BF16, device=dev)
        self.wd_table_host = torch.tensor(rows, dtype=torch.int32).contiguous()
        self.wd_table_dev = self.wd_table_host.to(dev)

    def refresh_packed_weights(self):
        """after a fused optimizer step wrote flat_param AND flat_shadow: one launch makes every dgrad
        operand, and every conv's cached operands become views of the flat buffers."""
        ops.pack_dgrad_batched(self.flat_shadow, self.flat_wd, self.wd_table_dev, self.wd_table_host)
        views = getattr(self, "_wviews", None)
        if views is None or views[0] is not self.flat_shadow or views[1] is not self.flat_wd:
            made = []
            for cb, off_s, off_d, n in self.wd_ranges:  # the views never change: made once per pair of flat buffers
                taps = cb.k[0] * cb.k[1] * cb.k[2]
                made.append((cb, self.flat_shadow[off_s:off_s + n].view(cb.Co, taps * cb.Ci),
                             self.flat_wd[off_d:off_d + n].view(cb.Ci, taps * cb.Co)))
            views = self._wviews = (self.flat_shadow, self.flat_wd, made)
        for cb, wf, wd in views[2]:
            cb.wf, cb.wd, cb.wver = wf, wd, cb.conv.weight._version
        self.param_epoch += 1

    def invalidate_weights(self):
        """forget every cached operand made from the parameters (packed bf16 conv weights, eval-mode BatchNorm
        scale / shift).  The caches follow the parameters' tensor versions, which writes through ``param.data`` do not
        move (``param.data`` carries a version counter of its own): code that edits weights that way -- an EMA copy, a
        clamp -- calls this afterwards.  ``with torch.no_grad(): param.mul_(...)``, ``load_state_dict`` and the
        optimizers need nothing."""
        for cb in self.all_convbn():
            cb.wver = -1
            cb.eval_params = None
        self.stem_wver = -1
        self.param_epoch += 1
        if self.flat_shadow is not None:  # the fused optimizers' shadow and dgrad operands: rebuilt from the master
            ops.shadow_bf16(self.flat_param, self.flat_shadow)
            self.refresh_packed_weights()

    def _bind_grads(self):
        """point every param.grad at its flat view (keeping accumulated values)."""
        if self.flat_grad is None or self.flat_grad.device != self.plist[0].device:
            raise Vc3dError("engine.setup_grads() has not been called for this device")
        fresh = True
        for p in self.plist:
            v = self.grad_views[id(p)]
            if p.grad is None:
                continue
            if p.grad.data_ptr() == v.data_ptr() and p.grad.stride() == v.stride():
                fresh = False
        if fresh:
            if self._zero_ev is not None:  # cleared on the side stream by the optimizer's zero_grad(): wait for that
                torch.cuda.current_stream().wait_event(self._zero_ev)
            else:
                self.flat_grad.zero_()
        self._zero_ev = None
        for p in self.plist:
            v = self.grad_views[id(p)]
            if not p.requires_grad:
                # a frozen parameter keeps grad None, as under torch autograd: the torch optimizers (what create_optimizer
                # returns for a partly frozen model) then skip it.  Its slice of the flat buffer is still written.
                continue
            if p.grad is None:
                if not fresh:
                    v.zero_()
                p.grad = v
            elif p.grad.data_ptr() != v.data_ptr():
                v.copy_(p.grad)
                p.grad = v

    def gview(self, p):
        return self.grad_views[id(p)]

    def _gphys(self, p):
        """physical [Co][kt][kh][kw][Ci] contiguous view of a conv weight gradient."""
        return self.grad_views[id(p)].permute(0, 2, 3, 4, 1)

    def _ready(self, *params, side=None):
        """the gradients of `params` are final in stream order from here on (the data-parallel wrapper launches a
        bucket's all-reduce when its last gradient is).  While a backward is being recorded the point is marked in the
        command list as well: its replay stops there and makes the same calls (CommandList.replay_segments)."""
        if self.grad_ready_cb is not None:
            # what the wrapper enqueues is its own to enqueue again when the step is replayed: it is not recorded
            rec, capi.RECORDER = capi.RECORDER, None
            try:
                cut = False
                for p in params:
                    r = self.grad_ready_cb(p)  # True: it enqueued work (a bucket's all-reduce); None: unknown, assume so
                    cut = cut or r is None or bool(r)
            finally:
                capi.RECORDER = rec
            if rec is not None:
                rec.mark(params, cut)
        elif self.opt_hook is not None:
            # side = "side": this point is in the side stream's order (a weight gradient was just issued there), and
            # everything made final so far, on either stream, precedes it there; "main": the step runs on one stream;
            # None: a main-stream point while weight gradients are in flight on the side stream.
            # The optimizer's launches ARE recorded, as commands like any other: their scalars (learning rate, Adam's
            # step count) are rewritten before each replay (optim._patch), so a replayed backward stays one C call.
            stream = self._side if side == "side" else torch.cuda.current_stream() if side == "main" else None
            rec = capi.RECORDER
            n0 = len(rec.cmds) if rec is not None else 0
            self.opt_hook(params, stream)
            if rec is not None:
                rec.opt_cmds.extend(range(n0, len(rec.cmds)))

    def _wgrad(self, d, h, dy, conv):
        """weight gradient of one conv.  Nothing downstream in backward consumes it, so it is issued on a side
        stream behind an event: the dependent chain of the main stream (dgrad -> BN backward reduce / finalize /
        apply of the next layer, mostly short bandwidth- or latency-bound kernels) overlaps with the MFMA-bound
        weight-gradient kernels instead of waiting behind them.  Joined at the end of backward."""
        if not SIDE_WGRAD or self.trace is not None:
            ops.conv3d_wgrad(d, h, dy, self._gphys(conv.weight))
            self._ready(conv.weight, side="main")
            return
        if self._side is None or self._side.device != dy.device:
            self._side = torch.cuda.Stream(device=dy.device)
        ev = torch.cuda.Event()
        ev.record()
        rec = capi.RECORDER
        rev = rec.event_record(capi.cur_stream()) if rec is not None else None
        with torch.cuda.stream(self._side):
            self._side.wait_event(ev)
            if rec is not None:
                rec.stream_wait(capi.cur_stream(), rev)
            ops.conv3d_wgrad(d, h, dy, self._gphys(conv.weight))
            self._ready(conv.weight, side="side")  # a bucket all-reduce must wait for THIS stream: the event is recorded here
        dy.record_stream(self._side)  # dy is dropped right after the dgrad on the main stream
        h.record_stream(self._side)

    # ------------------------------------------------------------------ forward
    def _ticket(self, cb, bwd):
        """one block of zero int32 ticket words per BatchNorm layer and direction (the fused finalize leaves it zero)."""
        dev = cb.bn.weight.device
        if self._tickets is None or self._tickets.device != dev:
            self._ticket_ix = {id(c): i for i, c in enumerate(self.all_convbn())}
            self._tickets = torch.zeros(2 * len(self._ticket_ix) * VC_TICKET_WORDS, dtype=torch.int32, device=dev)
            self._tape = None  # a recorded step names the old words
        i = (2 * self._ticket_ix[id(cb)] + (1 if bwd else 0)) * VC_TICKET_WORDS
        return self._tickets[i:i + VC_TICKET_WORDS]

    def _bticket(self, cb):
        return self._ticket(cb, True) if FUSED_BN_FIN else None

    def _bn_fin(self, cb, count):
        """descriptor of the training-mode finalize folded into the conv launch, or None."""
        bn = cb.bn
        if not FUSED_BN_FIN:
            return None, None
        if bn.momentum is None:
            raise Vc3dError("cumulative-average BatchNorm (momentum=None) is not supported")
        fin, p = ops.bn_fin(self._ticket(cb, False), count, bn.weight.detach(), bn.bias.detach(),
                            bn.running_mean if bn.track_running_stats else None,
                            bn.running_var if bn.track_running_stats else None, bn.momentum, bn.eps,
                            shift=self._shift(cb))
        if bn.track_running_stats and bn.num_batches_tracked is not None:
            self._nbt.append(bn.num_batches_tracked)
        return fin, p

    def _shift(self, cb):
        dev = cb.bn.weight.device
        if cb.shift is None or cb.shift.device != dev:
            cb.shift = torch.zeros(cb.Co, dtype=F32, device=dev)
            cb.shift_valid = False
            self._tape = None  # a recorded step names the old buffer
        return cb.shift

    def _first_stats(self, cb, y, stats, count):
        """BatchNorm statistics of a layer that has no shift yet (first training step, or a new device): one extra pass
        over y whose sums are taken relative to row 0 of y (vc_bn_stats), so that the very first batch is as robust
        against |mean| >> sigma as the later ones, whose fused sums are relative to the previous batch mean."""
        bn = cb.bn
        if bn.momentum is None:
            raise Vc3dError("cumulative-average BatchNorm (momentum=None) is not supported")
        shift = self._shift(cb)
        ops.bn_stats(y, cb.Co, sums=stats, shift=shift)
        p = ops.bn_finalize(stats, count, bn.weight.detach(), bn.bias.detach(),
                            bn.running_mean if bn.track_running_stats else None,
                            bn.running_var if bn.track_running_stats else None, bn.momentum, bn.eps, shift=shift,
                            shift_next=shift)
        if bn.track_running_stats and bn.num_batches_tracked is not None:
            self._nbt.append(bn.num_batches_tracked)
        cb.shift_valid = True
        return p

    def _bn_params(self, cb, stats, count, training):
        bn = cb.bn
        if training:
            if bn.momentum is None:
                raise Vc3dError("cumulative-average BatchNorm (momentum=None) is not supported")
            p = ops.bn_finalize(stats, count, bn.weight.detach(), bn.bias.detach(),
                                bn.running_mean if bn.track_running_stats else None,
                                bn.running_var if bn.track_running_stats else None, bn.momentum, bn.eps)
            if bn.track_running_stats and bn.num_batches_tracked is not None:
                self._nbt.append(bn.num_batches_tracked)
            return p
        # gamma/beta are rewritten by the fused optimizer step and the running statistics by the finalize kernels, both
        # through raw pointers (no tensor version moves): the two epoch counters stand for those writes
        ver = (bn.weight._version, bn.bias._version, bn.running_mean._version, bn.running_var._version,
               bn.weight.data_ptr(), bn.running_mean.data_ptr(), self.param_epoch, self.stat_epoch)
        if cb.eval_params is None or cb.eval_ver != ver or cb.eval_params.device != bn.weight.device:
            cb.eval_params = ops.bn_eval_params(bn.weight.detach(), bn.bias.detach(), bn.running_mean,
                                                bn.running_var, bn.eps)
            cb.eval_ver = ver
        return cb.eval_params

    def _conv_bn(self, cb, x, training, stats_pool):
        d = cb.desc(x)
        wf, _ = cb.weights()
        stats = stats_pool.take(cb.Co) if training else None
        To, Ho, Wo = d.out_dims()
        count = d.N * To * Ho * Wo
        if training and count <= 1:
            # nn.BatchNorm3d's own refusal (torch.nn.functional.batch_norm), same type and text: the reference raises it
            raise ValueError("Expected more than 1 value per channel when training, got input size %s"
                             % (torch.Size([d.N, cb.Co, To, Ho, Wo]),))
        if training and FUSED_BN_FIN and self.first_stats and not (cb.shift_valid and cb.shift is not None
                                                                     and cb.shift.device == x.device):
            y = ops.conv3d_fwd(d, x, wf)
            return d, y, self._first_stats(cb, y, stats, count)
        fin, params = self._bn_fin(cb, count) if training else (None, None)
        y = ops.conv3d_fwd(d, x, wf, stats=stats, fin=fin)
        if fin is None:
            params = self._bn_params(cb, stats, count, training)
        return d, y, params

    def _ds_forward_on_side_stream(self, ds, xin, training, stats_pool):
        """conv + BatchNorm + apply of a downsample branch on the side stream, behind an event on the current one.
        Returns (desc, y, params, residual, torch event, recorder event) -- the caller waits for the event before it
        reads the residual.  Tensors made here belong to the side stream's allocator pool and are used on the main
        stream afterwards (record_stream); the split-K workspace is per stream (ops._workspace)."""
        main = torch.cuda.current_stream()
        if self._side is None or self._side.device != xin.device:
            self._side = torch.cuda.Stream(device=xin.device)
        ev0 = torch.cuda.Event()
        ev0.record()
        rec = capi.RECORDER
        rev0 = rec.event_record(capi.cur_stream()) if rec is not None else None
        with torch.cuda.stream(self._side):
            self._side.wait_event(ev0)
            if rec is not None:
                rec.stream_wait(capi.cur_stream(), rev0)
            dd, yd, prmd = self._conv_bn(ds, xin, training, stats_pool)
            r = ops.bn_apply(yd, prmd, ds.Co, relu=False)
            ev = torch.cuda.Event()
            ev.record()
            rev = rec.event_record(capi.cur_stream()) if rec is not None else None
        xin.record_stream(self._side)
        for t in (yd, prmd, r):
            if t is not None:
                t.record_stream(main)
        return dd, yd, prmd, r, ev, rev

    # ------------------------------------------------------------------ recorded steps
    def _bn_signature(self):
        return tuple((cb.bn.momentum, cb.bn.eps, cb.bn.track_running_stats) for cb in self.all_convbn())

    def _replay_key(self, x, training, save):
        """the conditions under which a recorded training step stays valid, or None when this call is not a candidate.
        Everything a recorded command names by address must be persistent: the flat parameter / gradient buffers and
        the packed weights the fused optimizers maintain, BatchNorm buffers, shifts and tickets, the split-K workspace,
        and the step's own intermediates (kept alive by the tape)."""
        if not (REPLAY and training and save) or self.trace is not None or self.first_stats:
            return None
        if ops.TIMER is not None and ops.TIMER.select is None:
            return None  # a step that times every launch runs eagerly
        if isinstance(x, ops.PackedClips) or not x.is_cuda or x.dim() != 5 or x.shape[1] != 3 or x.dtype not in (F32, BF16):
            return None
        if x.shape[3] % 2 or x.shape[4] % 2 or not x.is_contiguous() or x.device.index != torch.cuda.current_device():
            return None
        if self.flat_shadow is None or self.flat_grad is None or x.device != self.flat_shadow.device:
            return None  # torch optimizers repack the weights into fresh tensors every step
        for cb in self.all_convbn():
            if cb.bn.training != training:
                return None
            if cb is not self.stem and (cb.wf is None or cb.wver != cb.conv.weight._version):
                return None
        bn0, bn1 = self.stem.bn, self.blocks[-1][0][-1].bn  # a module moved or re-created takes all its buffers along
        return (tuple(x.shape), x.dtype, x.device, capi.cur_stream(), self.flat_grad.data_ptr(), self.flat_shadow.data_ptr(),
                ops.WS_EPOCH, bn0.running_mean.data_ptr() if bn0.running_mean is not None else 0,
                bn1.running_var.data_ptr() if bn1.running_var is not None else 0, bool(capi.lib.vc_get_deterministic()), self._bn_signature(), FUSED_BN_FIN, FOLD_BN_RED,
                SIDE_WGRAD, SIDE_DS, DS_LAST)

    def _forward_recorded(self, x, key):
        """(features, step token) from the recorded step, or None when this call has to run eagerly."""
        tape = self._tape
        if tape is not None and tape["key"] == key:
            live = tape["live"]() if tape["live"] is not None else None
            if live is not None:
                return None  # the recorded step's activations still wait for their backward: this call runs eagerly
            fwd = tape["fwd"]
            fwd.patch("x", x.data_ptr())
            self.stat_epoch += 1
            pair = tcmd = None
            tk = getattr(ops.TIMER, "key", None) if ops.TIMER is not None else None
            if tk is not None and tk in tape["tags"]:
                tcmd, flops = tape["tags"][tk]
                pair = ops.TIMER.replay_pair(tk, flops)
            if pair is not None:
                fwd.replay(tcmd, pair[0], pair[1])
            else:
                fwd.replay()
            tape["serial"] += 1
            step = _TapeStep(tape, tape["serial"])
            tape["live"] = weakref.ref(step)
            return tape["feat"].detach(), step
        # record: the step runs as usual, with every launch noted and every intermediate tensor kept alive
        self._tape = None
        self.stem_wver = -1  # the stem's weight pack must be part of the recording, whatever was cached
        tensors, rec = [], Recorder()
        ops.TAPE, capi.RECORDER = tensors, rec
        try:
            feat, saved = self._forward(x, True, True)
        finally:
            ops.TAPE, capi.RECORDER = None, None
        if rec.failed is not None or ops.WS_EPOCH != key[6]:
            return feat, saved  # (e.g. the split-K workspace grew: the next step records again)
        fwd = rec.freeze({"x": x.data_ptr()})
        tape = {"key": key, "fwd": fwd, "bwd": None, "tensors": tensors, "feat": feat, "saved": saved, "serial": 0,
                "live": None, "tags": {tag[0]: (idx, tag[1]) for tag, idx in fwd.tags.items()}}
        step = _TapeStep(tape, 0)
        tape["live"] = weakref.ref(step)
        self._tape = tape
        return feat, step

    def _backward_recorded(self, step, dfeat):
        tape = step.tape
        if tape["live"] is None or tape["live"]() is not step or step.serial != tape["serial"]:
            raise Vc3dError("backward of a recorded step whose activations have been overwritten by a later step")
        tape["live"] = None
        df = dfeat.contiguous()
        eager = (tape is not self._tape or self.trace is not None or (ops.TIMER is not None and ops.TIMER.select is None) or
                 capi.cur_stream() != tape["key"][3] or self.flat_grad.data_ptr() != tape["key"][4])
        if eager:
            return self._backward(tape["saved"], df)
        bwd = tape["bwd"]
        opt = self.opt_owner if (self.opt_hook is not None and self.grad_ready_cb is None) else None
        if bwd is not None and bool(bwd.marks) != (self.grad_ready_cb is not None):
            bwd = tape["bwd"] = None  # recorded with / without a data-parallel wrapper, now the other way round: again
        if bwd is not None:
            # ... or with / without an optimizer stepping inside it, or with another optimizer state (its buffers are
            # named in the commands)
            ropt = tape.get("opt")
            if (ropt is None) != (opt is None) or (opt is not None and ropt["key"] != opt._state_key()):
                bwd = tape["bwd"] = None
        if bwd is not None:
            bwd.patch("dfeat", df.data_ptr())
            self._main = torch.cuda.current_stream()
            if opt is not None:
                opt._patch(bwd, ropt["cmds"])
            if bwd.marks:
                # data parallel: the wrapper's bucket launches come from Python between the kernels -- the list is
                # replayed in segments that end where a gradient became final while it was recorded
                bwd.replay_segments(lambda params: self._ready(*params))
            else:
                bwd.replay()
            if opt is not None:
                opt._done = ropt["done"]  # (what the recorded slices cover; step() does the rest)
                self.opt_hook = None
            if self.backward_done_cb is not None:
                self.backward_done_cb()
            return None
        rec = Recorder()
        ops.TAPE, capi.RECORDER = tape["tensors"], rec
        try:
            self._backward(tape["saved"], df)
        finally:
            ops.TAPE, capi.RECORDER = None, None
        if rec.failed is None and ops.WS_EPOCH == tape["key"][6]:
            tape["bwd"] = rec.freeze({"dfeat": df.data_ptr()})
            tape["opt"] = None if opt is None else {"cmds": list(rec.opt_cmds), "done": opt._done, "key": opt._state_key()}
        else:
            self._tape = None
        return None

    def forward(self, x, training, save):
        key = self._replay_key(x, training, save)
        if key is not None:
            try:
                out = self._forward_recorded(x, key)
            except Vc3dError:
                self._tape = None
                self._reset_tickets()
                raise
            except ValueError:  # BatchNorm's one-value refusal while recording (see below)
                self._tape = None
                if self._nbt:
                    torch._foreach_add_(self._nbt, 1)
                    self._nbt = []
                raise
            if out is not None:
                return out
        try:
            if x.is_cuda and x.device.index != torch.cuda.current_device():
                with torch.cuda.device(x.device):  # launches go to the current device's stream: make it the clips' device
                    return self._forward(x, training, save)
            return self._forward(x, training, save)
        except Vc3dError:
            self._reset_tickets()
            raise
        except ValueError:  # BatchNorm's one-value refusal: the layers before it have run, as in torch
            if self._nbt:
                torch._foreach_add_(self._nbt, 1)
                self._nbt = []
            raise

    def backward(self, saved, dfeat):
        self._bwd_serial += 1
        try:
            if isinstance(saved, _TapeStep):
                return self._backward_recorded(saved, dfeat)
            return self._backward(saved, dfeat)
        except Vc3dError:
            self._reset_tickets()
            raise

    def state_dict(self):
        """engine state that is not a module parameter or buffer: the BatchNorm shifts.  Saved by train.py so that a
        resumed run takes its sums relative to the same values as the uninterrupted one (bit-for-bit continuation)."""
        return {"bn_shift": [None if cb.shift is None else cb.shift.detach().cpu().clone() for cb in self.all_convbn()]}

    def load_state_dict(self, sd):
        dev = self.stem.bn.weight.device
        for cb, t in zip(self.all_convbn(), sd.get("bn_shift", [])):
            if t is not None:
                cb.shift = t.to(device=dev, dtype=F32).clone()
                cb.shift_valid = True
        self._tape = None  # recorded steps name the buffers just replaced

    def _count_batches(self):
        """num_batches_tracked += 1 for the BatchNorm layers of this forward: one launch over a table of their device
        addresses (a recordable command; torch._foreach_add_ for a list that is not the usual one)."""
        # (sorted: the table does not depend on the order in which this forward met its BatchNorm layers -- a step with
        # the downsample branches on the side stream lists them earlier than one without)
        key = tuple(sorted(t.data_ptr() for t in self._nbt))
        if getattr(self, "_nbt_key", None) != key:
            if any(t.dtype != torch.int64 or not t.is_cuda for t in self._nbt):
                if capi.RECORDER is not None:
                    capi.RECORDER.failed = "BatchNorm step counters that are not int64 device tensors"
                torch._foreach_add_(self._nbt, 1)
                return
            self._nbt_table = torch.tensor(key, dtype=torch.int64).to(self._nbt[0].device)
            self._nbt_key = key
            if capi.RECORDER is None:
                self._tape = None  # a recorded step names the old table (while recording, the new one is the one named)
        ops.incr_i64(self._nbt_table, len(key))

    def _reset_tickets(self):
        """a launch that failed midway may leave ticket words of the fused BatchNorm finalize non-zero (they are only
        reset by the workgroup that draws the last ticket): zero them so the next step starts clean."""
        if self._tickets is not None:
            self._tickets.zero_()

    def _forward(self, x, training, save):
        """x: [N][3][T][H][W] fp32/bf16 on the GPU.  Returns (features fp32 [N][F], saved state)."""
        packed = isinstance(x, ops.PackedClips)  # from ops.clip_pack_u8: already the stem's packed source
        if x.dim() != 5 or x.shape[1] != 3:
            raise Vc3dError("expected a clip batch N x 3 x T x H x W, got %s" % (tuple(x.shape),))
        if not x.is_cuda:
            raise Vc3dError("the vc3d backbone runs on the GPU only (no CPU fallback); move the clips to cuda")
        if not packed:
            if x.dtype not in (F32, BF16):
                x = x.float()
            if x.shape[3] % 2 or x.shape[4] % 2:
                # the stem's space-to-depth source wants even extents; a zero row / column at the far edge is what the
                # reference's padding-3 conv sees there anyway, and the output extents are the same (one extra copy, for
                # odd crops only)
                x = torch.nn.functional.pad(x, (0, x.shape[4] % 2, 0, x.shape[3] % 2))
            x = x.contiguous()
        N, _, T, H, W = x.shape
        dev = x.device
        if dev != self.stem.conv.weight.device:
            # what a torch.nn.DataParallel replica would do: its parameters are copies on another GPU, the engine belongs
            # to the original module.  One process per GPU (vc3d.ddp.DataParallel3D) is the supported way to use several.
            raise Vc3dError("clips on %s but the backbone's parameters on %s: move both to one GPU (for several GPUs run "
                            "one process per GPU with DataParallel3D; torch.nn.DataParallel replicas are not supported)"
                            % (dev, self.stem.conv.weight.device))
        if any(cb.bn.training != training for cb in self.all_convbn()):
            # e.g. bn.eval() on some layers of a model in train(): one mode per step is what the kernels implement
            raise Vc3dError("BatchNorm layers in a different train / eval mode than the backbone are not supported "
                            "(the engine runs every BatchNorm in the backbone's mode)")
        self._nbt = []
        if training:
            self.stat_epoch += 1
        stats_pool = _StatsPool(sum(cb.Co for cb in self.all_convbn()), dev) if training else None
        st = self.stem
        w = st.conv.weight
        if self.stem_wver != (w._version, self.param_epoch) or self.stem_ws is None or self.stem_ws.device != dev:
            self.stem_ws = ops.stem_pack_weight(w)
            self.stem_wver = (w._version, self.param_epoch)
        xs = x.xs if packed else ops.stem_pack_input(x)
        stats0 = stats_pool.take(st.Co) if training else None
        if training and FUSED_BN_FIN and self.first_stats and not (st.shift_valid and st.shift is not None
                                                                     and st.shift.device == dev):
            y0 = ops.stem_fwd(xs, self.stem_ws, N, T, H, W, st.Co)
            p0 = self._first_stats(st, y0, stats0, y0.numel() // st.Co)
        else:
            fin0, p0 = self._bn_fin(st, N * T * (H // 2) * (W // 2)) if training else (None, None)
            y0 = ops.stem_fwd(xs, self.stem_ws, N, T, H, W, st.Co, stats=stats0, fin=fin0)
            if fin0 is None:
                p0 = self._bn_params(st, stats0, y0.numel() // st.Co, training)
        pk, ps, pp = self.pool
        pd = ops.pool_desc(N, st.Co, T, H // 2, W // 2, pk, ps, pp)
        a0, idx = ops.maxpool3d_fwd(pd, y0, p0)
        self._t("conv1", clip=x, y=y0, prm=p0, pooled=a0)
        saved = {"stem": (xs, y0, p0, pd, idx, a0, (N, T, H, W)), "blocks": [], "training": training} if save else None
        if not save:
            del xs, y0, idx
        cur = a0
        fused_eval = not training and not save and self.trace is None and FUSED_EVAL_ACT
        for cbs, ds in self.blocks:
            xin = cur
            recs = []
            h = xin
            if fused_eval:
                # inference: BatchNorm (running statistics), the residual add and the ReLU are applied in the conv's
                # epilogue (vc_conv3d_fwd_act); no conv output is materialised before them
                def act(cb, x, relu, residual=None):
                    wf, _ = cb.weights()
                    return ops.conv3d_fwd(cb.desc(x), x, wf, act_params=self._bn_params(cb, None, 0, False), relu=relu,
                                          residual=residual)
                for cb in cbs[:-1]:
                    h = act(cb, h, True)
                r = act(ds, xin, False) if ds is not None else xin
                cur = act(cbs[-1], h, True, residual=r)
                continue
            # The downsample branch (1x1x1 conv + BatchNorm of the block input) depends on nothing the main branch
            # computes: it goes to the side stream when the block starts and is joined where the residual is added.
            # Both branches are short launches that leave most of the GPU idle in layers 2-4 (strided convs on tiny
            # planes), so they overlap almost for free.
            side_ds = ds is not None and SIDE_DS and SIDE_WGRAD and training and self.trace is None
            if side_ds:
                dsrec_side = self._ds_forward_on_side_stream(ds, xin, training, stats_pool)
            for i, cb in enumerate(cbs):
                last = (i == len(cbs) - 1)
                d, y, prm = self._conv_bn(cb, h, training, stats_pool)
                if not last:
                    z = ops.bn_apply(y, prm, cb.Co, relu=True)
                    recs.append((d, h, y, prm, z))
                    self._t(cb.name, x=h, y=y, prm=prm, out=z, res=None, relu=True)
                    h = z
                else:
                    if side_ds:
                        dd, yd, prmd, r, ev, rev = dsrec_side
                        torch.cuda.current_stream().wait_event(ev)
                        if rev is not None:
                            capi.RECORDER.stream_wait(capi.cur_stream(), rev)
                    elif ds is not None:
                        dd, yd, prmd = self._conv_bn(ds, xin, training, stats_pool)
                        r = ops.bn_apply(yd, prmd, ds.Co, relu=False)
                        self._t(ds.name, x=xin, y=yd, prm=prmd, out=r, res=None, relu=False)
                    else:
                        dd = yd = prmd = None
                        r = xin
                    # the block output's ReLU mask as bits for the backward (it reads 1 bit, not 16, per element)
                    bits = ops._empty(y.numel() // 8, dtype=torch.uint8, device=y.device) if save and RELU_BITS else None
                    out = ops.bn_apply(y, prm, cb.Co, residual=r, relu=True, relu_bits=bits)
                    recs.append((d, h, y, prm, out if bits is None else bits))
                    dsrec = (dd, yd, prmd)
                    self._t(cb.name, x=h, y=y, prm=prm, out=out, res=r, relu=True)
            if save:
                saved["blocks"].append((xin, recs, dsrec))
            cur = out
        if self._nbt:
            self._count_batches()  # one launch for all BatchNorm step counters
        Nn, To, Ho, Wo, Cf = cur.shape
        feat = ops.gavgpool_fwd(cur, Nn, To * Ho * Wo, Cf)
        self._t("final", x=cur, feat=feat)
        if save:
            saved["final"] = (tuple(cur.shape),)
        return feat, saved

    # ------------------------------------------------------------------ backward
    def _backward(self, saved, dfeat):
        """dfeat: fp32 [N][F].  Writes every backbone parameter gradient (accumulating)."""
        dev = dfeat.device
        self._main = torch.cuda.current_stream()
        sums_pool = _StatsPool(sum(cb.Co for cb in self.all_convbn()), dev)
        shape = saved["final"][0]
        N, To, Ho, Wo, Cf = shape
        g = ops.gavgpool_bwd(dfeat.contiguous(), N, To * Ho * Wo, Cf, shape)
        self._t("final", dfeat=dfeat, dx=g)
        order = list(zip(reversed(self.blocks), reversed(saved["blocks"])))
        # eval-mode forward (running statistics: fine-tuning with frozen BatchNorm statistics): plain reductions, no folds
        frozen = not saved.get("training", True)
        coef = None  # of this block's last BN, when its reduction came with the dgrad that completed g
        for bi, ((cbs, ds), (xin, recs, dsrec)) in enumerate(order):
            # last conv of the block: BN + residual + ReLU
            cb = cbs[-1]
            d, h, y, prm, out = recs[-1]  # `out`: the block output, or its ReLU mask as bits
            # identity shortcut into a pointwise first conv: the masked copy of g (gres) is not made -- that conv's data
            # gradient accumulates onto g itself and applies the ReLU bits to it on the way (vc_conv3d_dgrad_fused)
            d0 = recs[0][0]
            masked_acc = (FOLD_BN_RED and not frozen and self.trace is None and ds is None and out.dtype == torch.uint8 and
                          d0.kt * d0.kh * d0.kw == 1 and d0.st * d0.sh * d0.sw == 1)
            dy, gres = ops.bn_bwd(y, g, prm, cb.Co, self.gview(cb.bn.weight), self.gview(cb.bn.bias),
                                  mask=3 if out.dtype == torch.uint8 else 1, aux=out, want_g=not masked_acc,
                                  sums=sums_pool.take(cb.Co) if coef is None else None, ticket=self._bticket(cb),
                                  coef=coef, reduced=coef is not None, frozen=frozen)
            coef = None
            self._ready(cb.bn.weight, cb.bn.bias)
            self._wgrad(d, h, dy, cb.conv)
            _, wd = cb.weights()
            fold = FOLD_BN_RED and FUSED_BN_FIN and self.trace is None and not frozen

            def red_of(pcb, py, pprm, bits):
                """the reduction of pcb's BatchNorm backward, to ride along with the dgrad that makes its gradient"""
                c = ops._empty(2, pcb.Co, dtype=F32, device=dev)
                return c, (py, pprm, bits, sums_pool.take(pcb.Co), self._ticket(pcb, True), c,
                           self.gview(pcb.bn.weight), self.gview(pcb.bn.bias))

            icoef, red = red_of(cbs[-2], recs[-2][2], recs[-2][3], None) if fold else (None, None)
            dz = ops.conv3d_dgrad(d, dy, wd, bnred=red)
            self._t(cb.name, dz=g, dy=dy, gres=gres, dx=dz, dx_base=None)
            del dy
            # middle / first convs: BN + ReLU
            for i in range(len(cbs) - 2, -1, -1):
                cb = cbs[i]
                d, h, y, prm, z = recs[i]
                # BN -> ReLU with nothing in between: the ReLU mask is recomputed from y (scale * y + shift > 0,
                # the forward's own expression) instead of reading the activation z a third tensor per pass
                dy, _ = ops.bn_bwd(y, dz, prm, cb.Co, self.gview(cb.bn.weight), self.gview(cb.bn.bias), mask=2,
                                   sums=sums_pool.take(cb.Co) if icoef is None else None, ticket=self._bticket(cb),
                                   coef=icoef, reduced=icoef is not None, frozen=frozen)
                icoef = None
                self._ready(cb.bn.weight, cb.bn.bias)
                # A block with a downsample branch adds two data gradients into the block input's gradient.  Which one
                # goes first is free; the second accumulates (reads what the first wrote).  A STRIDED 1x1x1 downsample
                # only touches one input position in st*sh*sw: as the first it costs a memset of the whole gradient and
                # makes the block's first conv read all of it back (BasicBlock, layer2.0: 9 + 26 us of a 120 us pair);
                # as the second it reads and rewrites an eighth.  So the first conv goes first unless its own launch is
                # the dense GEMM that carries the folded reduction (Bottleneck: 1x1x1, unit stride).
                ds_last = (i == 0 and ds is not None and self.trace is None and DS_LAST and
                           not (d.kt * d.kh * d.kw == 1 and d.st * d.sh * d.sw == 1))
                if i == 0 and ds is not None:
                    dd, yd, prmd = dsrec
                    dyd, _ = ops.bn_bwd(yd, gres, prmd, ds.Co, self.gview(ds.bn.weight), self.gview(ds.bn.bias),
                                        mask=0, sums=sums_pool.take(ds.Co), ticket=self._bticket(ds), frozen=frozen)
                    self._ready(ds.bn.weight, ds.bn.bias)
                    self._wgrad(dd, xin, dyd, ds.conv)
                    _, wdd = ds.weights()
                    if not ds_last:
                        # downsample branch first so the first conv's dgrad can accumulate onto it
                        dzd = gres
                        gres = ops.conv3d_dgrad(dd, dyd, wdd)
                        self._t(ds.name, dz=dzd, dy=dyd, dx=gres.clone() if self.trace is not None else None,
                                dx_base=None)
                        del dyd
                self._wgrad(d, h, dy, cb.conv)
                _, wd = cb.weights()
                dz_in = dz
                if i == 0:
                    base = gres.clone() if self.trace is not None else None
                    red = None
                    if fold and bi + 1 < len(order):
                        # dz is the gradient the previous block's last BN reduces next: take its sums here
                        (pcbs, _), (_, precs, _) = order[bi + 1]
                        _, _, py, pprm, pout = precs[-1]
                        if pout.dtype == torch.uint8:
                            coef, red = red_of(pcbs[-1], py, pprm, pout)
                    if ds_last:
                        dz = ops.conv3d_dgrad(d, dy, wd)
                        dz = ops.conv3d_dgrad(dd, dyd, wdd, out=dz, accumulate=True, bnred=red)
                        del dyd
                    elif masked_acc:
                        dz = ops.conv3d_dgrad(d, dy, wd, out=g, accumulate=True, bnred=red, acc_bits=out)
                    else:
                        dz = ops.conv3d_dgrad(d, dy, wd, out=gres, accumulate=True, bnred=red)
                else:
                    base = None
                    icoef, red = red_of(cbs[i - 1], recs[i - 1][2], recs[i - 1][3], None) if fold else (None, None)
                    dz = ops.conv3d_dgrad(d, dy, wd, bnred=red)
                self._t(cb.name, dz=dz_in, dy=dy, dx=dz, dx_base=base)
                del dy
            g = dz
        # stem: max pool -> BN+ReLU (mask recomputed) -> conv wgrad
        xs, y0, p0, pd, idx, a0, (N, T, H, W) = saved["stem"]
        st = self.stem
        if self.trace is not None:  # tests look at the materialised pool-input gradient as well
            din = ops.maxpool3d_bwd(pd, g, idx)
            self._t("conv1", dpooled=g, dz=din)
        if all(k <= 2 * s for k, s in zip(self.pool[0], self.pool[1])):
            dy0 = ops.bn_bwd_pooled(pd, y0, g, idx, p0, st.Co, self.gview(st.bn.weight), self.gview(st.bn.bias),
                                    sums=sums_pool.take(st.Co), pooled=a0, ticket=self._bticket(st), frozen=frozen)
        else:  # windows overlapping more than twice per axis: materialise the pool-input gradient
            dy0, _ = ops.bn_bwd(y0, ops.maxpool3d_bwd(pd, g, idx), p0, st.Co, self.gview(st.bn.weight),
                                self.gview(st.bn.bias), mask=2, sums=sums_pool.take(st.Co), ticket=self._bticket(st),
                                frozen=frozen)
        self._ready(st.bn.weight, st.bn.bias)
        if self._stem_scratch is None or self._stem_scratch.device != dev or self._stem_scratch.shape[0] != st.Co:
            self._stem_scratch = torch.zeros(st.Co, ops.STEM_K, dtype=F32, device=dev)
            if capi.RECORDER is None:
                self._tape = None  # (while recording, the new buffer is the one the commands name)
        self._t("conv1", dy=dy0)
        ops.stem_wgrad(xs, dy0, N, T, H, W, st.Co, self._gphys(st.conv.weight), scratch=self._stem_scratch)
        self._ready(st.conv.weight)
        if self._side is not None:
            torch.cuda.current_stream().wait_stream(self._side)
            rec = capi.RECORDER
            if rec is not None:
                rec.stream_wait(capi.cur_stream(), rec.event_record(self._side.cuda_stream))
        if self.backward_done_cb is not None:
            self.backward_done_cb()


class _StatsPool:
    """one zero-filled allocation carved into per-layer [R][2][C] accumulators."""

    def __init__(self, total_channels, device):
        self.buf = ops.zeros(total_channels * 2 * VC_STAT_REPLICAS, F32, device)
        self.off = 0

    def take(self, C):
        n = VC_STAT_REPLICAS * 2 * C
        if self.off + n > self.buf.numel():
            raise Vc3dError("stats pool exhausted")
        v = self.buf[self.off:self.off + n]
        self.off += n
        return v
