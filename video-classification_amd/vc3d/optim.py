"""Fused SGD / Adam over the engine's flat buffers (what create_optimizer returns for 'sgd' / 'adam' on the GPU).

torch.optim.SGD arithmetic (momentum, weight decay, nesterov; dampening 0) in ONE launch for the
whole model (vc_sgd_step), which also writes the bf16 shadow of the new parameters; a second
launch (vc_pack_dgrad_batched) makes every conv's dgrad operand.  Replaces ~10 multi-tensor
launches plus one repack launch per conv per step.  Subclasses torch.optim.Optimizer, so
``zero_grad``, ``param_groups[0]['lr']`` and LR schedulers work as with the reference's optimizer
(/root/reference/lib/utils/utils.py:22-29, lib/core/function.py:80-82).  FlatAdam is the same for
torch.optim.Adam, the reference's default optimizer (config.py:43, utils.py:30-33): one vc_adam_step launch."""
import torch

from . import ops
from .capi import Vc3dError


class _FlatOptimizer(torch.optim.Optimizer):
    """what the two fused optimizers share: the engine's flat parameter / gradient buffers and their checks"""

    def __init__(self, model, defaults):
        core = model.module if hasattr(model, "module") else model
        if not hasattr(core, "backbone") or not hasattr(core.backbone, "engine"):
            raise Vc3dError("%s needs a model built by create_model (backbone with a vc3d engine)" % type(self).__name__)
        self.core = core
        super().__init__(list(core.parameters()), defaults)

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
        eng = self._flat()
        if self.exp_avg is None or self.exp_avg.numel() != eng.flat_param.numel() or self.exp_avg.device != eng.flat_param.device:
            self.exp_avg, self.exp_avg_sq = torch.zeros_like(eng.flat_param), torch.zeros_like(eng.flat_param)
        g = self.param_groups[0]
        self.steps += 1
        ops.adam_step(eng.flat_param, eng.flat_grad, self.exp_avg, self.exp_avg_sq, eng.flat_shadow, g["lr"], g["betas"][0],
                      g["betas"][1], g["eps"], g["weight_decay"], self.steps)
        eng.refresh_packed_weights()
        return loss

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
        eng = self._flat()
        dev = eng.flat_param.device
        if self.mom is None or self.mom.numel() != eng.flat_param.numel() or self.mom.device != dev:
            self.mom = torch.zeros_like(eng.flat_param) if self.param_groups[0]["momentum"] > 0 else None
        g = self.param_groups[0]
        ops.sgd_step(eng.flat_param, eng.flat_grad, self.mom, eng.flat_shadow, g["lr"], g["momentum"],
                     g["weight_decay"], g["nesterov"])
        eng.refresh_packed_weights()
        return loss

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
