"""Per-epoch training and validation loops with the reference's signatures
(/root/reference/lib/core/function.py:35-36 ``train``, :98 ``validate``).

The reference bodies call unimplemented stubs (Act_Init, ChooseInput, rollout,
Score_Updater) and do not parse; these are the build's own.  Structure kept from the
reference: per batch, TRAIN_STEP inner steps of forward -> rollout -> criterion ->
zero_grad -> backward -> step (function.py:65-85).  Differences, on purpose:
 - the loss is accumulated on the device and read back only every PRINT_EVERY batches
   (the reference's ``loss.item()`` at :85 is a host sync per step);
 - ``validate`` returns top-1 accuracy (its caller compares the result, train.py:87-89),
   averaging logits over the views of each video when the loader yields
   N x V x 3 x T x H x W (multi-clip / multi-crop protocol).
"""
import logging
import time

import torch
import torch.utils.data.distributed

from utils.utils import AverageMeter, ChooseInput, rollout

logger = logging.getLogger(__name__)


def Act_Init():
    """reference function.py:17-22 (empty stub): the initial action; unused by the 3D path."""
    return None


class Score_Updater(object):
    """running sum of class scores over the inner steps / views of one batch
    (reference function.py:24-33 declares it, :146 calls ``update``)."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.total = None
        self.n = 0

    def update(self, score):
        self.total = score.detach().float() if self.total is None else self.total + score.detach().float()
        self.n += 1

    def mean(self):
        return self.total / max(self.n, 1)


def _device_of(model):
    return next(model.parameters()).device


def _distributed_sum(t):
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.all_reduce(t)
    return t


def train(config, train_loader, model, criterion, optimizer, epoch):
    """one epoch; returns None (reference function.py:35-94)."""
    batch_time = AverageMeter()
    losses = AverageMeter()
    model.train()
    dev = _device_of(model)
    loss_acc = torch.zeros((), device=dev)
    acc_n = 0
    end = time.time()
    for i, (vedio, target) in enumerate(train_loader):
        vedio = vedio.to(dev, non_blocking=True)
        target = target.to(dev, non_blocking=True)
        action = Act_Init()
        for step in range(config.TRAIN.TRAIN_STEP):
            inp = ChooseInput(vedio, action)
            action, score, value = model(inp)
            log_prob, advantages = rollout(vedio, model, action, value)
            loss = criterion(score, target, log_prob, advantages)
            optimizer.zero_grad()
            if hasattr(optimizer, "arm_backward"):
                optimizer.arm_backward()  # one backward per step, nothing edits the gradients: the step may start inside it
            loss.backward()
            optimizer.step()
            loss_acc += loss.detach()
            acc_n += 1
        batch_time.update(time.time() - end)
        end = time.time()
        if i % config.TRAIN.PRINT_EVERY == 0:
            losses.update(float(loss_acc) / max(acc_n, 1), acc_n * vedio.shape[0])  # the only host sync
            loss_acc.zero_()
            acc_n = 0
            logger.info('Epoch [%d][%d/%d] time %.3f (%.3f) loss %.4f (%.4f) clips/s %.1f',
                        epoch, i, len(train_loader), batch_time.val, batch_time.avg, losses.val, losses.avg,
                        vedio.shape[0] * config.TRAIN.TRAIN_STEP / max(batch_time.val, 1e-9))


def _real_samples(loader):
    """samples this rank really owns: DistributedSampler pads the index list with repeats so that every rank gets the
    same count; the repeats come last in a rank's sequence and must not be counted twice in the accuracy."""
    smp = getattr(loader, "sampler", None)
    if isinstance(smp, torch.utils.data.distributed.DistributedSampler) and not smp.drop_last:
        n = len(smp.dataset)
        return (n - smp.rank + smp.num_replicas - 1) // smp.num_replicas
    return None


def validate(config, val_loader, model, criterion, epoch):
    """returns top-1 accuracy in [0, 1] over the loader (all ranks, if distributed); top-5 and the loss are logged and
    left in ``validate.last`` (dict: top1, top5, loss, count).  Reference loop: function.py:117-149 -- TEST_STEP inner
    steps per batch whose scores a Score_Updater accumulates (the reference reads config.TRAIN.TEST_STEP at :129,
    which does not exist; config.py:65 defines TEST.TEST_STEP).  Here an inner step is one temporal clip: raw uint8
    videos are cut into TEST.TEST_STEP uniformly spaced clips by ChooseInput and their logits averaged; a batch that
    already holds float clips (nothing to choose from) is scored once, one that holds N x V views once per view."""
    batch_time = AverageMeter()
    score_updater = Score_Updater()
    model.eval()
    dev = _device_of(model)
    stat = torch.zeros(4, device=dev, dtype=torch.float64)  # top-1 correct, count, loss sum, top-5 correct
    limit = _real_samples(val_loader)
    seen = 0
    with torch.no_grad():
        end = time.time()
        for i, (vedio, target) in enumerate(val_loader):
            vedio = vedio.to(dev, non_blocking=True)
            target = target.to(dev, non_blocking=True)
            score_updater.reset()
            if vedio.dim() == 6:  # N x V views: average the logits of the V views of each video
                n, v = vedio.shape[:2]
                flat = vedio.reshape((n * v,) + tuple(vedio.shape[2:]))
                chunk = max(1, config.TEST.BATCH_SIZE)
                parts = [model(flat[s:s + chunk])[1] for s in range(0, n * v, chunk)]
                score_updater.update(torch.cat(parts).view(n, v, -1).mean(1))
            else:
                nsteps = max(1, int(config.TEST.TEST_STEP)) if vedio.dtype == torch.uint8 else 1
                action = Act_Init()
                for step in range(nsteps):
                    action, score, _ = model(ChooseInput(vedio, action, step, nsteps))
                    score_updater.update(score)
            score = score_updater.mean()
            if limit is not None:  # drop the sampler's padding repeats (they are the tail of this rank's sequence)
                keep = max(0, min(target.numel(), limit - seen))
                seen += target.numel()
                if keep == 0:
                    continue
                score, target = score[:keep], target[:keep]
            loss = criterion(score, target, None, None)
            k = min(5, score.shape[1])
            top = score.topk(k, dim=1).indices
            stat[0] += (top[:, 0] == target).sum()
            stat[1] += target.numel()
            stat[2] += loss.double() * target.numel()
            stat[3] += (top == target[:, None]).any(1).sum()
            batch_time.update(time.time() - end)
            end = time.time()
            if i % config.TEST.PRINT_EVERY == 0:
                logger.info('Test [%d][%d/%d] time %.3f (%.3f)', epoch, i, len(val_loader), batch_time.val,
                            batch_time.avg)
    stat = _distributed_sum(stat).cpu()
    n = stat[1].clamp_min(1)
    acc = float(stat[0] / n)
    validate.last = {"top1": acc, "top5": float(stat[3] / n), "loss": float(stat[2] / n), "count": int(stat[1])}
    logger.info('Test epoch %d: top-1 %.4f top-5 %.4f loss %.4f (%d videos)', epoch, acc, validate.last["top5"],
                validate.last["loss"], int(stat[1]))
    return acc


validate.last = None
