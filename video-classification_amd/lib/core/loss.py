"""Training loss with the reference's signature and arithmetic
(/root/reference/lib/core/loss.py:7-30): cross-entropy on the class scores plus the
weighted actor and critic terms.  With all-zero ``log_prob`` / ``advantages`` (what the
build's ``rollout`` returns: the reference's RL rollout is an empty stub) it is plain CE.
"""
import torch
import torch.nn.functional as F


class Loss(torch.nn.Module):
    def __init__(self, config):
        super().__init__()
        self.r_act = config.TRAIN.R_ACT
        self.r_crt = config.TRAIN.R_CRT

    def forward(self, score, y, log_prob, advantages):
        total = F.cross_entropy(score, y)
        if log_prob is not None and advantages is not None:
            critic = (advantages * advantages).mean()
            actor = -(advantages.detach() * log_prob).mean()
            total = total + self.r_act * actor + self.r_crt * critic
        return total
