"""Host-side helpers with the reference's names (/root/reference/lib/utils/utils.py)."""
import logging
import os
import time

import torch


def create_optimizer(cfg, model):
    """SGD(momentum, wd, nesterov) or Adam(lr); None for any other name (reference utils.py:13-36)."""
    name = cfg.TRAIN.OPTIMIZER
    params = [p for p in model.parameters()]
    if name == 'sgd' and _fusable(model):
        from vc3d.optim import FlatSGD  # same arithmetic as torch.optim.SGD, one launch for the whole model
        return FlatSGD(model, lr=cfg.TRAIN.LR, momentum=cfg.TRAIN.MOMENTUM, weight_decay=cfg.TRAIN.WD,
                       nesterov=cfg.TRAIN.NESTEROV)
    if name == 'sgd':
        return torch.optim.SGD(params, lr=cfg.TRAIN.LR, momentum=cfg.TRAIN.MOMENTUM,
                               weight_decay=cfg.TRAIN.WD, nesterov=cfg.TRAIN.NESTEROV)
    if name == 'adam' and _fusable(model):
        from vc3d.optim import FlatAdam  # same arithmetic as torch.optim.Adam (default betas / eps), one launch
        return FlatAdam(model, lr=cfg.TRAIN.LR)
    if name == 'adam':
        return torch.optim.Adam(params, lr=cfg.TRAIN.LR)
    return None


def _fusable(model):
    core = model.module if hasattr(model, "module") else model
    return (hasattr(core, "backbone") and hasattr(core.backbone, "engine") and hasattr(core, "clf_haed")
            and not getattr(core, "use_lstm", False)  # the LSTM / actor / critic parameters live outside the flat buffer
            and all(p.is_cuda and p.requires_grad for p in core.parameters()))  # frozen parameters: torch skips them


class AverageMeter(object):
    """running value / sum / count / average (reference utils.py:39-54)."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count if self.count else 0


def ChooseInput(vedio, action, step=0, nsteps=1):
    """reference utils.py:57-64 is an empty stub (frame/modality picking of the unfinished RL
    design).  Here the whole clip is the model input whatever the action.
    Raw decoded frames (uint8 N x Ts x Hs x Ws x 3 on the GPU) are turned into the model's input here, in one
    GPU pass (vc_clip_pack_u8): CLIP_LEN frames every FRAME_STEP, a CROP_SIZE crop, normalised with MODEL.MEAN/STD;
    random position and horizontal flip while gradients are on (training), centred otherwise.  With nsteps > 1
    (validate's TEST.TEST_STEP inner steps) and gradients off, inner step k takes the k-th of nsteps uniformly spaced
    temporal clips of the video (centre crop): the multi-clip test protocol."""
    if vedio.dtype != torch.uint8:
        return vedio
    from core.config import config as cfg
    from vc3d import ops
    if vedio.dim() != 5 or vedio.shape[-1] != 3:
        raise ValueError("raw frames must be uint8 N x Ts x Hs x Ws x 3, got %s" % (tuple(vedio.shape),))
    N, Ts, Hs, Ws, _ = vedio.shape
    T, S, fstep = cfg.MODEL.CLIP_LEN, cfg.MODEL.CROP_SIZE, cfg.MODEL.FRAME_STEP
    span = (T - 1) * fstep + 1
    if Ts < span or Hs < S or Ws < S:
        raise ValueError("frames %dx%dx%d are smaller than the clip %dx%dx%d (step %d)" % (Ts, Hs, Ws, T, S, S, fstep))
    crops = torch.empty(N, 4, dtype=torch.int32)
    if torch.is_grad_enabled():
        crops[:, 0] = torch.randint(0, Ts - span + 1, (N,))
        crops[:, 1] = torch.randint(0, Hs - S + 1, (N,))
        crops[:, 2] = torch.randint(0, Ws - S + 1, (N,))
        crops[:, 3] = torch.randint(0, 2, (N,))
    else:
        t0 = (Ts - span) // 2 if nsteps <= 1 else int(round(step * (Ts - span) / float(nsteps - 1)))
        crops[:, 0], crops[:, 1], crops[:, 2], crops[:, 3] = t0, (Hs - S) // 2, (Ws - S) // 2, 0
    return ops.clip_pack_u8(vedio.contiguous(), crops, T, S, S, fstep, cfg.MODEL.MEAN, cfg.MODEL.STD)


def rollout(vedio, model, action, value):
    """reference utils.py:67-80 is an empty stub ("rollout according to model's policy ... lists of log_prob and
    advantages"); neither an environment nor a reward is defined upstream.  Without the recurrent head (the default)
    this returns (None, None) and Loss reduces to cross-entropy.  With it (MODEL.USE_LSTM) the minimal well-defined
    instance: the action head's output p is the probability of a Bernoulli action a ~ B(p), log_prob = log P(a) (N x 1,
    differentiable in p); advantages = gamma * V' - V (N x 1, reward-free one-step TD) where V' is the critic's value
    after TRAIN.ROLLOUT_STEP further steps of the policy under no_grad (the LSTM state is restored afterwards) and
    V = `value` keeps its graph, so r_crt * mean(adv^2) trains the critic and the actor term sees adv.detach()
    (reference loss.py:25-27)."""
    core = model.module if hasattr(model, "module") else model
    if not getattr(core, "use_lstm", False):
        return None, None
    from core.config import config as cfg
    p = action.clamp(1e-6, 1.0 - 1e-6)
    a = torch.bernoulli(p.detach())
    log_prob = a * torch.log(p) + (1.0 - a) * torch.log(1.0 - p)
    lstm = core.lstm
    keep = (lstm.h, lstm.c)
    was_training = core.training
    core.eval()  # the look-ahead must not touch the BatchNorm running statistics
    with torch.no_grad():
        v_next = value.detach()
        for _ in range(max(0, int(cfg.TRAIN.ROLLOUT_STEP))):
            _, _, v_next = model(ChooseInput(vedio, a))
    lstm.h, lstm.c = keep
    core.train(was_training)
    advantages = cfg.TRAIN.GAMMA * v_next.detach() - value
    return log_prob, advantages


def create_logger(cfg, phase='train'):
    """file + console logger; file name {DATASET}_{time}_{phase}.log under OUTPUT_DIR
    (reference utils.py:84-104)."""
    os.makedirs(cfg.OUTPUT_DIR, exist_ok=True)
    stamp = time.strftime('%Y-%m-%d-%H-%M')
    path = os.path.join(cfg.OUTPUT_DIR, '{}_{}_{}.log'.format(cfg.DATASET, stamp, phase))
    logger = logging.getLogger()
    logger.setLevel(logging.INFO)
    fmt = logging.Formatter('%(asctime)-15s: %(message)s')
    # idempotent, like the reference's logging.basicConfig: a second call in one process (train.py then valid.py, the
    # tests) replaces the handlers this function added instead of stacking new ones (every line would print twice)
    for h in [h for h in logger.handlers if getattr(h, "_vc_handler", False)]:
        logger.removeHandler(h)
        h.close()
    fh = logging.FileHandler(path)
    fh.setFormatter(fmt)
    sh = logging.StreamHandler()
    for h in (fh, sh):
        h._vc_handler = True
        logger.addHandler(h)
    return logger
