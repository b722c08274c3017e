"""Fused SGD / Adam over the engine's flat buffers (what create_optimizer returns for 'sgd' / 'adam' on the GPU).

torch.optim.SGD arithmetic (momentum, weight decay, nesterov; dampening 0) in ONE launch for the
whole model (vc_sgd_step), which also writes the bf16 shadow of the new parameters; a second
launch (vc_pack_dgrad_batched) makes every conv's dgrad operand.  Replaces ~10 multi-tensor
launches plus one repack launch per conv per step.  Subclasses torch.optim.Optimizer, so
``zero_grad``, ``param_groups[0]['lr']`` and LR schedulers work as with the reference's optimizer
(/root/reference/lib/utils/utils.py:22-29, lib/core/function.py:80-82).  FlatAdam is the same for
torch.optim.Adam, the reference's default optimizer (config.py:43, utils.py:30-33): one vc_adam_step launch."""
import os

import torch

from . import ops
from .capi import Vc3dError

_IN_BACKWARD = os.environ.get("VC_OPT_IN_BACKWARD", "1") != "0"
_EARLY_ZERO = os.environ.get("VC_NO_EARLY_ZERO") is None


class _FlatOptimizer(torch.optim.Optimizer):
    """what the two fused optimizers share: the engine's flat parameter / gradient buffers and their checks"""

    def __init__(self, model, defaults):
        core = model.module if hasattr(model, "module") else model
        if not hasattr(core, "backbone") or not hasattr(core.backbone, "engine"):
            raise Vc3dError("%s needs a model built by create_model (backbone with a vc3d engine)" % type(self).__name__)
        self.core = core
        self._done = 0  # elements of the flat buffers that slices inside the backward have stepped already
        self._prepared = False  # arm_backward() has made this step's state (momentum buffers, Adam's step count)
        self._applied = []  # (data parallel) the buckets stepped behind their all-reduce: [(start, end)]
        self._consumed_ev = None  # recorded when step() has issued the last reader of the gradients
        self._consumed_serial = -1  # the engine's backward count at that moment
        super().__init__(list(core.parameters()), defaults)

    # ---- stepping inside the backward
    # The flat buffers are laid out in the order the backward finishes gradients (head, layer4 ... stem), so what is
    # final at any moment is a prefix.  Armed for one backward, the optimizer applies its step to that prefix in slices
    # while the rest of the backward still runs: the slice launches go on the engine's side stream right behind the
    # weight gradient that completed them (its order already has every gradient made so far), an HBM-bound pass beside
    # the main stream's latency-bound layers instead of 117 us (R3D-18) / 186 us (R3D-50) alone at the end of the step.
    # step() then finishes what is left (the stem's parameters, made last).  Same arithmetic, element for element.
    # Measured: R3D-18 step -0.65 % (30 us of the 117: the slices take bandwidth from the kernels they run beside, and the
    # side stream's chain is level with the main one at the end of the backward).  A stream of their own for the slices
    # measured +0.4 % instead.
    # elements: a slice launch costs what any launch costs; 3 M elements are ~10 us of step (VC_OPT_MIN_SLICE overrides)
    MIN_SLICE = int(os.environ.get("VC_OPT_MIN_SLICE", 3 << 20))

    def arm_backward(self, data_parallel=False):
        """The NEXT backward computes this step's complete gradients (no accumulation over several backwards, nothing
        edits the gradients before step()): parts of step() may run inside it.  Without this call step() does all the
        work, as before.  Under the data-parallel wrapper the call is ignored unless data_parallel=True: the slices are
        then its buckets, each stepped on the communication stream behind its all-reduce -- correct (tests/test_ddp_gpu.py)
        but not the default: on one rank with the buckets forced it measured 4.80 against 4.79 ms, and on several GPUs,
        where a slice delays the next bucket's all-reduce on that stream, it has not been measured at all.
        Ignored inside DataParallel3D.no_sync() and when VC_OPT_IN_BACKWARD=0."""
        eng = self.core.backbone.engine
        if not _IN_BACKWARD or eng.flat_grad is None or eng.flat_shadow is None:
            return False
        dp = None
        if eng.grad_ready_cb is not None:
            # under the data-parallel wrapper the ready points are its buckets': the slices are the buckets themselves
            # (contiguous pieces of the flat buffers, in order), stepped on its communication stream right behind each
            # bucket's all-reduce
            dp = getattr(eng.grad_ready_cb, "__self__", None)
            if not data_parallel or dp is None or not hasattr(dp, "after_reduce") or not getattr(dp, "_sync", True):
                return False
        if self._done or self._applied:
            raise Vc3dError("%s: arm_backward() again before step() finished the previous one" % type(self).__name__)
        if not self._prepared:  # (armed before and no backward came, or it failed before any slice: arm again)
            self._prepare(eng)
            self._prepared = True
        self._index = {id(p): i for i, (p, _, _) in enumerate(eng.grad_ranges)}
        self._flags = [False] * len(eng.grad_ranges)
        self._front = 0
        self._armed_eng, self._armed_dp = eng, dp
        if dp is not None:
            dp.after_reduce = self._after_reduce
            return True
        eng.opt_hook, eng.opt_owner = self._on_ready, self
        return True

    def _after_reduce(self, start, end):
        """(data parallel) bucket [start, end) of the flat gradient has been averaged, in the current stream's order"""
        self._apply(self._armed_eng, start, end)
        self._applied.append((start, end))  # (buckets need not finish in order: step() takes what is missing)

    def _on_ready(self, params, stream):
        eng = self._armed_eng
        for p in params:
            i = self._index.get(id(p))
            if i is not None:
                self._flags[i] = True
        n = len(self._flags)
        while self._front < n and self._flags[self._front]:
            self._front += 1
        end = eng.grad_ranges[self._front][1] if self._front < n else eng.flat_param.numel()
        if self._front == n:
            eng.opt_hook = None  # one backward only
        if stream is None or end - self._done < self.MIN_SLICE:
            return False
        with torch.cuda.stream(stream):
            self._apply(eng, self._done, end)
        self._done = end
        return True

    def _finish(self, eng):
        """step(): the part of the flat buffers that no slice has covered"""
        eng.opt_hook = None
        if getattr(self, "_armed_dp", None) is not None:
            self._armed_dp.after_reduce = None  # (its slices sit on the communication stream, joined when backward ended)
            self._armed_dp = None
        if not self._prepared:
            self._prepare(eng)
        pos, n = 0, eng.flat_param.numel()
        for a, b in sorted([(0, self._done)] + self._applied) + [(n, n)]:
            if a > pos:
                self._apply(eng, pos, a)
            pos = max(pos, b)
        self._done, self._prepared, self._applied = 0, False, []
        if eng.flat_grad.is_cuda:  # everything that reads this step's gradients has been issued: zero_grad() may clear them behind it
            self._consumed_ev = torch.cuda.Event()
            self._consumed_ev.record()
            self._consumed_serial = eng._bwd_serial

    def zero_grad(self, set_to_none=True):
        """torch's zero_grad; with set_to_none (the default) the flat gradient buffer is cleared right here on the
        engine's side stream -- behind the step that last read it -- instead of on the backward's own stream when it
        starts: 18 us (133 MB) that otherwise sit between the loss and the first backward launch."""
        super().zero_grad(set_to_none)
        eng = self.core.backbone.engine
        ev = self._consumed_ev
        if not (set_to_none and _EARLY_ZERO and ev is not None and eng.flat_grad is not None and eng.flat_grad.is_cuda):
            return
        if any(p.grad is not None for p in eng.plist):  # (a parameter outside this optimizer's groups kept its gradient)
            return
        if eng._bwd_serial != self._consumed_serial:  # a backward ran since step(): its writes are not behind `ev`
            return
        if eng._side is None or eng._side.device != eng.flat_grad.device:
            eng._side = torch.cuda.Stream(device=eng.flat_grad.device)
        with torch.cuda.stream(eng._side):
            eng._side.wait_event(ev)
            eng.flat_grad.zero_()
            eng._zero_ev = torch.cuda.Event()
            eng._zero_ev.record()

    def _flat(self):
        """the engine with flat gradients / weights set up and every parameter's gradient inside the flat buffer"""
        eng = self.core.backbone.engine
        head = (self.core.clf_haed.weight, self.core.clf_haed.bias)
        eng.head_params = head
        dev = head[0].device
        if eng.flat_grad is None or eng.flat_grad.device != dev:
            eng.setup_grads(extra_first=head)
        if getattr(self, "_known_for", None) is not eng.plist:  # checked once per flat-buffer layout
            known = {id(p) for p in eng.plist}
            for p in self.param_groups[0]["params"]:
                if id(p) not in known:
                    raise Vc3dError("%s: a parameter is not part of the engine's flat buffer" % type(self).__name__)
            self._known_for = eng.plist
        if eng.flat_shadow is None:
            eng.setup_flat_weights()
        for p in eng.plist:  # gradients that did not come from the engine's backward: move them in
            v = eng.grad_views[id(p)]
            if p.grad is None or not p.requires_grad:
                # the torch optimizers skip such a parameter altogether (no weight decay, no state update); the
                # one-launch step over the flat buffer cannot, so this is refused rather than silently trained
                raise Vc3dError("%s: every parameter needs a gradient (frozen or unused parameters are not supported "
                                "by the fused step; use the torch optimizer for them)" % type(self).__name__)
            if p.grad.data_ptr() != v.data_ptr():
                v.copy_(p.grad)
                p.grad = v
        return eng


class FlatAdam(_FlatOptimizer):
    """torch.optim.Adam(params, lr, betas, eps, weight_decay) arithmetic (amsgrad off) in one launch."""

    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(model, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        self.exp_avg = self.exp_avg_sq = None
        self.steps = 0

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if closure is not None and (self._done or self._applied):
            raise Vc3dError("FlatAdam: a closure cannot be combined with arm_backward()")
        eng = self._flat()
        self._finish(eng)
        eng.refresh_packed_weights()
        return loss

    def _prepare(self, eng):
        if self.exp_avg is None or self.exp_avg.numel() != eng.flat_param.numel() or self.exp_avg.device != eng.flat_param.device:
            self.exp_avg, self.exp_avg_sq = torch.zeros_like(eng.flat_param), torch.zeros_like(eng.flat_param)
        self.steps += 1

    def _state_key(self):
        return (self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr())

    def _patch(self, cl, idxs):
        """this step's scalars into the recorded vc_adam_step commands (ints: ..., n, step, stream; floats: lr, b1, b2, eps, wd)"""
        g = self.param_groups[0]
        for i in idxs:
            c = cl.array[i]
            c.iarg[6] = self.steps
            c.farg[0], c.farg[1], c.farg[2], c.farg[3], c.farg[4] = g["lr"], g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"]

    def _apply(self, eng, a, b):
        g = self.param_groups[0]
        ops.adam_step(eng.flat_param[a:b], eng.flat_grad[a:b], self.exp_avg[a:b], self.exp_avg_sq[a:b], eng.flat_shadow[a:b],
                      g["lr"], g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"], self.steps)

    def state_dict(self):
        sd = super().state_dict()
        sd["flat_exp_avg"] = None if self.exp_avg is None else self.exp_avg.clone()
        sd["flat_exp_avg_sq"] = None if self.exp_avg_sq is None else self.exp_avg_sq.clone()
        sd["flat_steps"] = self.steps
        return sd

    def load_state_dict(self, sd):
        sd = dict(sd)
        m, v, self.steps = sd.pop("flat_exp_avg", None), sd.pop("flat_exp_avg_sq", None), int(sd.pop("flat_steps", 0))
        super().load_state_dict(sd)
        dev = next(self.core.parameters()).device
        self.exp_avg = None if m is None else m.detach().to(device=dev, dtype=torch.float32).clone()
        self.exp_avg_sq = None if v is None else v.detach().to(device=dev, dtype=torch.float32).clone()


class FlatSGD(_FlatOptimizer):
    def __init__(self, model, lr, momentum=0.0, weight_decay=0.0, nesterov=False):
        if nesterov and momentum <= 0:
            raise ValueError("Nesterov momentum requires a momentum")
        super().__init__(model, dict(lr=lr, momentum=momentum, weight_decay=weight_decay, nesterov=nesterov))
        self.mom = None

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if closure is not None and (self._done or self._applied):
            raise Vc3dError("FlatSGD: a closure cannot be combined with arm_backward()")
        eng = self._flat()
        self._finish(eng)
        eng.refresh_packed_weights()
        return loss

    def _prepare(self, eng):
        dev = eng.flat_param.device
        if self.mom is None or self.mom.numel() != eng.flat_param.numel() or self.mom.device != dev:
            self.mom = torch.zeros_like(eng.flat_param) if self.param_groups[0]["momentum"] > 0 else None

    def _state_key(self):
        return (0 if self.mom is None else self.mom.data_ptr(), self.param_groups[0]["momentum"] > 0)

    def _patch(self, cl, idxs):
        """this step's scalars into the recorded vc_sgd_step commands (ints: ..., n, nesterov, stream; floats: lr, momentum, wd)"""
        g = self.param_groups[0]
        for i in idxs:
            c = cl.array[i]
            c.iarg[5] = 1 if g["nesterov"] else 0
            c.farg[0], c.farg[1], c.farg[2] = g["lr"], g["momentum"], g["weight_decay"]

    def _apply(self, eng, a, b):
        g = self.param_groups[0]
        ops.sgd_step(eng.flat_param[a:b], eng.flat_grad[a:b], None if self.mom is None else self.mom[a:b], eng.flat_shadow[a:b],
                     g["lr"], g["momentum"], g["weight_decay"], g["nesterov"])

    def state_dict(self):
        sd = super().state_dict()
        sd["flat_momentum"] = None if self.mom is None else self.mom.clone()
        return sd

    def load_state_dict(self, sd):
        sd = dict(sd)
        mom = sd.pop("flat_momentum", None)
        super().load_state_dict(sd)
        dev = next(self.core.parameters()).device
        self.mom = None if mom is None else mom.detach().to(device=dev, dtype=torch.float32).clone()
