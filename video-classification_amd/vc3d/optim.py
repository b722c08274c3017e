"""Fused SGD over the engine's flat buffers (what create_optimizer returns for 'sgd' on the GPU).

torch.optim.SGD arithmetic (momentum, weight decay, nesterov; dampening 0) in ONE launch for the
whole model (vc_sgd_step), which also writes the bf16 shadow of the new parameters; a second
launch (vc_pack_dgrad_batched) makes every conv's dgrad operand.  Replaces ~10 multi-tensor
launches plus one repack launch per conv per step.  Subclasses torch.optim.Optimizer, so
``zero_grad``, ``param_groups[0]['lr']`` and LR schedulers work as with the reference's optimizer
(/root/reference/lib/utils/utils.py:22-29, lib/core/function.py:80-82)."""
import torch

from . import ops
from .capi import Vc3dError


class FlatSGD(torch.optim.Optimizer):
    def __init__(self, model, lr, momentum=0.0, weight_decay=0.0, nesterov=False):
        core = model.module if hasattr(model, "module") else model
        if not hasattr(core, "backbone") or not hasattr(core.backbone, "engine"):
            raise Vc3dError("FlatSGD needs a model built by create_model (backbone with a vc3d engine)")
        if nesterov and momentum <= 0:
            raise ValueError("Nesterov momentum requires a momentum")
        self.core = core
        super().__init__(list(core.parameters()), dict(lr=lr, momentum=momentum, weight_decay=weight_decay,
                                                       nesterov=nesterov))
        self.mom = None

    def _engine_ready(self):
        eng = self.core.backbone.engine
        head = (self.core.clf_haed.weight, self.core.clf_haed.bias)
        eng.head_params = head
        dev = head[0].device
        if eng.flat_grad is None or eng.flat_grad.device != dev:
            eng.setup_grads(extra_first=head)
        known = {id(p) for p in eng.plist}
        for p in self.param_groups[0]["params"]:
            if id(p) not in known:
                raise Vc3dError("FlatSGD: a parameter is not part of the engine's flat buffer")
        if eng.flat_shadow is None:
            eng.setup_flat_weights()
        if self.mom is None or self.mom.numel() != eng.flat_param.numel() or self.mom.device != dev:
            self.mom = torch.zeros_like(eng.flat_param) if self.param_groups[0]["momentum"] > 0 else None
        return eng

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        eng = self._engine_ready()
        for p in eng.plist:  # gradients that did not come from the engine's backward: move them in
            v = eng.grad_views[id(p)]
            if p.grad is None or not p.requires_grad:
                # torch.optim.SGD skips such a parameter altogether (no weight decay, no momentum update); the one-launch
                # step over the flat buffer cannot, so this is refused rather than silently trained
                raise Vc3dError("FlatSGD: every parameter needs a gradient (frozen or unused parameters are not "
                                "supported by the fused step; use torch.optim.SGD for them)")
            if p.grad.data_ptr() != v.data_ptr():
                v.copy_(p.grad)
                p.grad = v
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
