"""Training loss with the reference's signature and arithmetic
(/root/reference/lib/core/loss.py:7-30): cross-entropy on the class scores plus the
weighted actor and critic terms.  With all-zero ``log_prob`` / ``advantages`` (what the
build's ``rollout`` returns: the reference's RL rollout is an empty stub) it is plain CE.
"""
import os

import torch
import torch.nn.functional as F

_FUSED_CE = os.environ.get("VC_NO_FUSED_CE") is None


class Loss(torch.nn.Module):
    def __init__(self, config):
        super().__init__()
        self.r_act = config.TRAIN.R_ACT
        self.r_crt = config.TRAIN.R_CRT

    def forward(self, score, y, log_prob, advantages):
        if (score.is_cuda and score.dim() == 2 and score.dtype == torch.float32 and y.dtype == torch.int64 and y.dim() == 1
                and y.is_cuda and score.shape[1] <= 1024 and _FUSED_CE):
            # the same mean cross-entropy, value and gradient in one launch between the forward and the backward
            # instead of torch's six (vc_softmax_ce)
            from vc3d.autograd import cross_entropy
            total = cross_entropy(score, y)
        else:
            total = F.cross_entropy(score, y)
        if log_prob is not None and advantages is not None:
            critic = (advantages * advantages).mean()
            actor = -(advantages.detach() * log_prob).mean()
            total = total + self.r_act * actor + self.r_crt * critic
        return total
