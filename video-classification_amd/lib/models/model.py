"""Model wrapper with the reference's interface (/root/reference/lib/models/model.py:13-61).

``create_model(config, is_train)`` returns a module whose forward takes a clip batch
N x 3 x T x H x W and returns the reference's 3-tuple ``(action, score, value)``
(model.py:27-41): ``score`` = N x CLFDIM class logits from the classifier head on the
backbone features; ``action`` and ``value`` are N x 1 zeros, because the reference's
LSTM / actor / critic heads are an unfinished RL design outside this hot path
(SURVEY.md section 2, rows 2 and 9).  The head keeps the reference's attribute name
``clf_haed`` (model.py:24) so checkpoints keep their keys; its input width is the backbone
width (512 * expansion) since there is no LSTM in between.
"""
import torch
import torch.nn as nn

import vc3d
from models.backbone import getBackbone


class VedioClfNet(nn.Module):
    def __init__(self, config, is_train=True):
        super().__init__()
        self.backbone = getBackbone(config, is_train)
        self.use_lstm = bool(config.MODEL.get("USE_LSTM", False))
        if self.use_lstm:
            # the reference's recurrent head (model.py:18-25): LSTM -> act_head (Linear + Sigmoid), clf_haed, crit_head
            from models.lstm import getLSTM
            hid = config.MODEL.LSTM_OUTDIM
            self.lstm = getLSTM(config, in_dim=self.backbone.out_features)
            self.act_head = nn.Sequential(nn.Linear(hid, 1), nn.Sigmoid())
            self.clf_haed = nn.Linear(hid, config.MODEL.CLFDIM)
            self.crit_head = nn.Linear(hid, 1)
        else:
            self.clf_haed = nn.Linear(self.backbone.out_features, config.MODEL.CLFDIM)

    def features(self, x):
        return self.backbone(x)

    def forward(self, x):
        eng = self.backbone.engine
        if self.use_lstm:
            # reference model.py:35-41: x -> backbone -> lstm -> three heads; every Linear is the vc3d fp32 head kernel
            eng.head_params = ()
            h = self.lstm(self.backbone(x))
            a0 = self.act_head[0]
            action = torch.sigmoid(vc3d.linear_apply(h, a0.weight, a0.bias))
            score = vc3d.linear_apply(h, self.clf_haed.weight, self.clf_haed.bias)
            value = vc3d.linear_apply(h, self.crit_head.weight, self.crit_head.bias)
            return action, score, value
        eng.head_params = (self.clf_haed.weight, self.clf_haed.bias)
        feat = self.backbone(x)
        score = vc3d.linear_apply(feat, self.clf_haed.weight, self.clf_haed.bias, eng)
        # (one constant tensor per batch size, not a fill launch per step between the head and the loss: nothing writes it)
        key = (score.shape[0], score.dtype, score.device)
        if getattr(self, "_zeros_key", None) != key:
            self._zeros_key, self._zeros = key, torch.zeros(score.shape[0], 1, dtype=score.dtype, device=score.device)
        return self._zeros, score, self._zeros

    def init_weights(self):
        """the backbone loads its checkpoint in getBackbone; the head keeps nn.Linear's default."""
        return self


def create_model(config, is_train=True):
    """reference model.py:47-61."""
    model = VedioClfNet(config, is_train)
    if is_train and config.MODEL.INIT_WEIGHTS:
        model.init_weights()
    return model
