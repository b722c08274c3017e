"""The two autograd nodes of the product path: the whole backbone, and the classifier head.

The backbone node owns its backward (engine.backward) and writes parameter gradients
straight into the engine's flat gradient buffer, assigning ``param.grad`` itself; it returns
None for the parameter inputs, so ``torch.autograd.grad`` on backbone parameters is not
supported (``loss.backward()`` + optimizer, the reference's loop, is).
"""
import torch

from . import ops


class _BackboneFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, engine, training, anchor, *params):
        # `params`: ONE parameter that requires a gradient (or none): it ties this node into the graph so that backward
        # is called; the gradients themselves are written by the engine, not returned (62 inputs cost autograd ~50 us)
        # `anchor` = whether activations must be kept (grad mode is off inside Function.forward)
        # (eval mode with gradients on keeps the activations too: a backward then runs with frozen BatchNorm statistics,
        # engine._backward; plain inference belongs under torch.no_grad(), which takes the fused inference epilogues)
        feat, saved = engine.forward(x, training, save=bool(anchor))
        ctx.engine = engine
        ctx.saved = saved
        ctx.n_params = len(params)
        return feat

    @staticmethod
    def backward(ctx, dfeat):
        eng = ctx.engine
        if ctx.saved is None:
            raise RuntimeError("backbone forward ran without saving activations")
        eng._bind_grads()
        eng.backward(ctx.saved, dfeat)
        ctx.saved = None
        return (None, None, None, None) + (None,) * ctx.n_params


def backbone_apply(engine, x, training):
    params = engine.params_backward_order()
    engine._fn_params = params
    if engine.flat_grad is None or engine.flat_grad.device != params[0].device:
        engine.setup_grads(extra_first=getattr(engine, "head_params", ()))
    need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in params)
    if torch.is_grad_enabled() and getattr(x, "requires_grad", False):
        # the stem has no data-gradient kernel (training never needs d loss / d clip): refuse rather than hand back None
        from .capi import Vc3dError
        raise Vc3dError("the vc3d backbone does not compute the gradient with respect to the input clip "
                        "(pass the clip without requires_grad)")
    tie = next((p for p in params if p.requires_grad), None) if need_grad else None
    if tie is None:
        return _BackboneFn.apply(x, engine, training, need_grad)
    return _BackboneFn.apply(x, engine, training, need_grad, tie)


class _LinearFn(torch.autograd.Function):
    """classifier head; when ``engine`` is given, W/b gradients go to its flat buffer."""

    @staticmethod
    def forward(ctx, feat, W, b, engine):
        ctx.save_for_backward(feat, W)
        ctx.has_bias = b is not None
        ctx.engine = engine
        ctx.params = (W, b)
        return ops.linear_fwd(feat.contiguous(), W.detach(), None if b is None else b.detach())

    @staticmethod
    def backward(ctx, dlogits):
        feat, W = ctx.saved_tensors
        Wp, bp = ctx.params
        eng = ctx.engine
        if eng is not None and eng.flat_grad is not None and id(Wp) in eng.grad_views:
            eng._bind_grads()
            dW = eng.gview(Wp)
            db = eng.gview(bp) if bp is not None else None
            dfeat = ops.linear_bwd(dlogits.contiguous(), feat, W.detach(), dW, db,
                                   want_dfeat=ctx.needs_input_grad[0])
            eng._ready(Wp, *( [bp] if bp is not None else []))
            return dfeat, None, None, None
        dW = torch.zeros_like(W)
        db = torch.zeros(W.shape[0], dtype=W.dtype, device=W.device) if ctx.has_bias else None
        dfeat = ops.linear_bwd(dlogits.contiguous(), feat, W.detach(), dW, db, want_dfeat=ctx.needs_input_grad[0])
        return dfeat, dW, db, None


def linear_apply(feat, W, b, engine=None):
    return _LinearFn.apply(feat, W, b, engine)


class _SoftmaxCEFn(torch.autograd.Function):
    """mean cross-entropy with its gradient made in the forward's launch (ops.softmax_ce)"""

    @staticmethod
    def forward(ctx, logits, labels):
        loss, dlog = ops.softmax_ce(logits.contiguous(), labels.contiguous(), want_grad=ctx.needs_input_grad[0])
        ctx.save_for_backward(dlog)
        return loss.view(())

    @staticmethod
    def backward(ctx, g):
        (dlog,) = ctx.saved_tensors
        return dlog * g, None


def cross_entropy(logits, labels):
    """F.cross_entropy(logits, labels) (mean reduction, class-index targets) for fp32 logits [N][K] on the GPU: one launch
    forward, one multiply backward."""
    return _SoftmaxCEFn.apply(logits, labels)

