"""LSTM cell on the vc3d kernels (reference lib/models/lstm.py:16,25-27: nn.LSTMCell).

gates[N][4H] = [x h] [W_ih W_hh]^T is ONE bf16 MFMA GEMM: a 1x1x1 conv over N positions with Ci = Fin + H input and
Co = 4H output channels (vc_conv3d_fwd); its backward is the same conv's dgrad (d[x h]) and wgrad (d[W_ih W_hh], fp32).
The gate non-linearities, the state update and their backward are one bandwidth pass each (vc_lstm_pointwise_fwd / _bwd).
Biases, the cell state and every gradient of a parameter stay fp32; gates and their gradients cross the GEMMs as bf16.
"""
import torch

from . import ops
from .capi import Vc3dError, check, cur_stream, lib, ptr

F32, BF16 = torch.float32, torch.bfloat16


class _PackedLSTMWeights:
    """bf16 [4H][Fin+H] forward / wgrad operand and its [Fin+H][4H] dgrad twin, rebuilt when a master weight changed."""

    def __init__(self):
        self.key = None
        self.wf = self.wd = None

    def get(self, w_ih, w_hh):
        key = (w_ih._version, w_hh._version, w_ih.data_ptr(), w_hh.data_ptr(), w_ih.device)
        if key != self.key:
            wcat = torch.cat([w_ih.detach(), w_hh.detach()], 1).float().contiguous()
            co, ci = wcat.shape
            w5 = wcat.view(co, 1, 1, 1, ci).permute(0, 4, 1, 2, 3)  # [Co][Ci][1][1][1] in channels_last_3d storage
            self.wf, self.wd = ops.pack_weight(w5, want_dgrad=True)
            self.key = key
        return self.wf, self.wd


class _LSTMCellFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, h_prev, c_prev, w_ih, w_hh, b_ih, b_hh, packed):
        if not x.is_cuda:
            raise Vc3dError("the vc3d LSTM cell runs on the GPU only (no CPU fallback)")
        N, fin = x.shape
        H = h_prev.shape[1]
        if (fin + H) % 64 != 0 or (4 * H) % 64 != 0:
            raise Vc3dError("LSTM cell: input + hidden width and 4 x hidden must be multiples of 64 (got %d, %d)" % (fin, H))
        wf, wd = packed.get(w_ih, w_hh)
        xh = torch.cat([x.detach(), h_prev.detach()], 1).to(BF16).contiguous()
        d = ops.conv_desc(N, fin + H, 1, 1, 1, 4 * H, 1, 1, 0)
        gates = ops.conv3d_fwd(d, xh.view(N, 1, 1, 1, fin + H), wf)
        h = torch.empty(N, H, dtype=F32, device=x.device)
        c = torch.empty(N, H, dtype=F32, device=x.device)
        saved = torch.empty(N, 5 * H, dtype=F32, device=x.device)
        cp = c_prev.detach().float().contiguous()
        check(lib.vc_lstm_pointwise_fwd(ptr(gates), ptr(b_ih.detach()), ptr(b_hh.detach()), ptr(cp), ptr(h), ptr(c), None,
                                        ptr(saved), N, H, cur_stream()), "vc_lstm_pointwise_fwd")
        ctx.save_for_backward(xh, saved, cp)
        ctx.d, ctx.wd, ctx.fin, ctx.H = d, wd, fin, H
        return h, c

    @staticmethod
    def backward(ctx, dh, dc):
        xh, saved, cp = ctx.saved_tensors
        N, H, fin, d = xh.shape[0], ctx.H, ctx.fin, ctx.d
        dev = xh.device
        dgates = torch.empty(N, 4 * H, dtype=BF16, device=dev)
        dc_prev = torch.empty(N, H, dtype=F32, device=dev)
        db_ih = torch.zeros(4 * H, dtype=F32, device=dev)
        db_hh = torch.zeros(4 * H, dtype=F32, device=dev)
        check(lib.vc_lstm_pointwise_bwd(ptr(None if dh is None else dh.float().contiguous()),
                                        ptr(None if dc is None else dc.float().contiguous()), ptr(saved), ptr(cp), ptr(dgates),
                                        ptr(dc_prev), ptr(db_ih), ptr(db_hh), N, H, cur_stream()), "vc_lstm_pointwise_bwd")
        dg5 = dgates.view(N, 1, 1, 1, 4 * H)
        dxh = ops.conv3d_dgrad(d, dg5, ctx.wd).view(N, fin + H).float()
        dw = torch.zeros(4 * H, 1, 1, 1, fin + H, dtype=F32, device=dev)
        ops.conv3d_wgrad(d, xh.view(N, 1, 1, 1, fin + H), dg5, dw)
        dw = dw.view(4 * H, fin + H)
        return (dxh[:, :fin].contiguous(), dxh[:, fin:].contiguous(), dc_prev, dw[:, :fin].contiguous(),
                dw[:, fin:].contiguous(), db_ih, db_hh, None)


def lstm_cell(x, h_prev, c_prev, w_ih, w_hh, b_ih, b_hh, packed):
    return _LSTMCellFn.apply(x.float(), h_prev, c_prev, w_ih, w_hh, b_ih, b_hh, packed)
