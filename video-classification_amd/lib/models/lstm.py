"""LSTM with a state kept in the module: the reference's interface (/root/reference/lib/models/lstm.py:12-40) on the
vc3d kernels.  ``self.lstm`` is an ``nn.LSTMCell`` used as the PARAMETER HOLDER only (same state_dict keys as the
reference: lstm.lstm.weight_ih / weight_hh / bias_ih / bias_hh); the arithmetic is vc3d.lstm.lstm_cell (bf16 MFMA GEMM
+ one fused element-wise pass).

Differences from the reference, on purpose: ``reset`` allocates on the parameters' device instead of a hard-coded
``.cuda()`` (lstm.py:22-23); the state handed to the next step is detached (the reference keeps the graph of every
earlier step alive, and its loop (function.py:65-82) calls backward once per step: torch raises on the second call);
the input width is the backbone's feature width (the reference pairs ResNet-18's 512 features with LSTM_INDIM = 2048,
config.py:25,29)."""
import torch
import torch.nn as nn

from vc3d.lstm import _PackedLSTMWeights, lstm_cell


class LSTM(nn.Module):
    def __init__(self, in_dim, out_dim):
        super().__init__()
        self.lstm = nn.LSTMCell(in_dim, out_dim)
        self.out_dim = out_dim
        self.h = None
        self.c = None
        self._packed = _PackedLSTMWeights()

    def reset(self, batch_size):
        dev = self.lstm.weight_ih.device
        self.h = torch.zeros((batch_size, self.out_dim), device=dev)
        self.c = torch.zeros((batch_size, self.out_dim), device=dev)

    def forward(self, input):
        if self.h is None or self.h.shape[0] != input.shape[0] or self.h.device != input.device:
            self.reset(input.shape[0])
        p = self.lstm
        h, c = lstm_cell(input, self.h, self.c, p.weight_ih, p.weight_hh, p.bias_ih, p.bias_hh, self._packed)
        self.h, self.c = h.detach(), c.detach()
        return h


def getLSTM(config=None, in_dim=None):
    """reference lstm.py:32-40 (getLSTM() reads the global config); in_dim defaults to MODEL.LSTM_INDIM."""
    if config is None:
        from core.config import config
    return LSTM(in_dim or config.MODEL.LSTM_INDIM, config.MODEL.LSTM_OUTDIM)
