#!/usr/bin/env python3
"""Training entry point with the reference's flow (/root/reference/train.py:29-91; that file does
not parse, so the body is the build's own): logger -> create_model -> data parallel -> Loss ->
create_optimizer -> MultiStepLR -> loaders -> epoch loop of train() / validate(), keeping the best
validation accuracy (the reference leaves "save the model" as a TODO at :91; done here).

Data parallelism: the reference wraps the model in torch.nn.DataParallel over config.GPUS
(:42-43) with a global batch of BATCH_SIZE * len(gpus) (:65).  Here every GPU is its own process:
`python train.py GPUS=0,1,2,3` starts one rank per listed device itself (launch.py: fresh children,
before this process touches the GPU) -- or launch it with `python -m torch.distributed.run
--nproc-per-node G train.py`.  Each rank loads BATCH_SIZE clips per step, gradients are averaged
with bucketed RCCL all-reduce overlapped with backward (vc3d/ddp.py).  A plain `python train.py`
runs the one GPU config.GPUS names ('0').

Overrides: KEY=VALUE pairs on the command line, e.g. TRAIN.END_EPOCH=1 MODEL.RESNET_TYPE=50.
"""
import os as _os

# more hardware queues than HIP's default of 4: backward stream + weight-gradient stream + bucket stream + RCCL
# (vc3d/ddp.py); must be set before the HIP runtime initialises
if int(_os.environ.get("GPU_MAX_HW_QUEUES", "4") or 4) < 8 and _os.environ.get("VC_KEEP_HW_QUEUES") != "1":
    _os.environ["GPU_MAX_HW_QUEUES"] = "8"

import os
import sys

import _init_paths  # noqa: F401
import launch
from core.config import config


def _heavy_imports():
    """torch, the HIP library and the modules built on them: imported only in a process that will use the GPU
    (the parent of a fan-out never does)."""
    global torch, dist, train, validate, Loss, get_dataset, create_model, create_logger, create_optimizer, DataParallel3D
    import torch
    import torch.distributed as dist
    import torch.utils.data
    import torch.utils.data.distributed
    from core.function import train, validate
    from core.loss import Loss
    from dataset.dataset import get_dataset
    from models.model import create_model
    from utils.utils import create_logger, create_optimizer
    from vc3d.ddp import DataParallel3D


def apply_overrides(cfg, argv):
    for item in argv:
        key, _, val = item.partition("=")
        node = cfg
        parts = key.split(".")
        for p in parts[:-1]:
            node = node[p]
        old = node[parts[-1]]
        if isinstance(old, bool):
            val = val.lower() in ("1", "true", "yes")
        elif isinstance(old, list):
            val = [type(old[0])(v) if old else (int(v) if v.lstrip('-').isdigit() else float(v)) for v in val.split(",") if v]
        elif isinstance(old, int):  # an integer default (LR_DECAY_RATE = 1) may be overridden by a fraction
            val = int(val) if val.lstrip('-').isdigit() else float(val)
        elif old is not None:
            val = type(old)(val)
        node[parts[-1]] = val
    return cfg


def fan_out_if_needed(cfg, script, argv):
    """config.GPUS names several devices (reference train.py:42, valid.py:42) and this process is not yet a rank:
    start one rank per device and return their exit code; None when this process is to do the work itself."""
    gpus = launch.parse_gpus(cfg.GPUS)
    if launch.launched():
        return None
    dry = os.environ.get("VC_DRY_LAUNCH_PARENT") == "1"
    if len(gpus) == 1 and not dry:
        return None  # a single process drives the one device GPUS names (setup_distributed)
    return launch.fan_out(script, argv, len(gpus), device_ids=gpus, dry=dry)


def setup_distributed(cfg=None):
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if cfg is not None and world > 1 and launch.DEVICES_ENV in os.environ:
        n = len(launch.parse_gpus(cfg.GPUS))
        if n != world:
            raise SystemExit("config.GPUS=%r names %d devices but WORLD_SIZE is %d" % (cfg.GPUS, n, world))
    index = launch.local_device_index()
    if cfg is not None and not launch.launched():
        index = launch.parse_gpus(cfg.GPUS)[0]
    if index >= torch.cuda.device_count():
        raise SystemExit("device %d requested (config.GPUS / LOCAL_RANK) but %d visible" % (index, torch.cuda.device_count()))
    torch.cuda.set_device(index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", index))
    return world, int(os.environ.get("RANK", "0")), torch.device("cuda", index)


def make_loader(cfg, if_train, world, rank, length=None):
    ds = get_dataset(if_train=if_train, config=cfg, length=length)
    bs = cfg.TRAIN.BATCH_SIZE if if_train else cfg.TEST.BATCH_SIZE
    sampler = None
    if world > 1:
        sampler = torch.utils.data.distributed.DistributedSampler(ds, world, rank,
                                                                  shuffle=if_train and cfg.TRAIN.SHUFFLE)
    return torch.utils.data.DataLoader(ds, batch_size=bs, shuffle=(if_train and cfg.TRAIN.SHUFFLE and sampler is None),
                                       sampler=sampler, num_workers=cfg.WORKERS, pin_memory=True, drop_last=if_train)


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    cfg = apply_overrides(config.clone(), argv)
    code = fan_out_if_needed(cfg, os.path.abspath(__file__), argv)
    if code is not None:
        if code:
            raise SystemExit(code)
        return None
    if os.environ.get(launch.DRY_ENV) == "1":
        launch.dry_child_report()
        return None
    _heavy_imports()
    world, rank, dev = setup_distributed(cfg)
    logger = create_logger(cfg, 'train')
    model = create_model(cfg, is_train=True).to(dev)
    net = DataParallel3D(model, bucket_mb=cfg.TRAIN.BUCKET_MB, grad_dtype=cfg.TRAIN.get('GRAD_DTYPE') or None) if world > 1 else model
    criterion = Loss(cfg).to(dev)
    optimizer = create_optimizer(cfg, model)
    if optimizer is None:
        raise SystemExit("unknown TRAIN.OPTIMIZER %r" % cfg.TRAIN.OPTIMIZER)
    lr_scheduler = torch.optim.lr_scheduler.MultiStepLR(optimizer, cfg.TRAIN.LR_MILESTONES, cfg.TRAIN.LR_DECAY_RATE)
    n_train = cfg.get("SYNTH_TRAIN_LEN")
    n_val = cfg.get("SYNTH_VAL_LEN")
    train_loader = make_loader(cfg, True, world, rank, n_train)
    valid_loader = make_loader(cfg, False, world, rank, n_val)
    best_perf = -1.0
    begin_epoch = cfg.TRAIN.BEGIN_EPOCH
    ckpt_path = os.path.join(cfg.OUTPUT_DIR, 'checkpoint.pth')
    resume = cfg.TRAIN.RESUME or (ckpt_path if cfg.TRAIN.BEGIN_EPOCH > 0 and os.path.isfile(ckpt_path) else '')
    if resume:
        # reference config.py:53 has BEGIN_EPOCH but nothing restores the optimizer; here a run continues exactly where it
        # stopped: weights, BatchNorm buffers, momentum, learning-rate schedule, best accuracy, next epoch
        ck = torch.load(resume, map_location='cpu', weights_only=True)
        model.load_state_dict({(k[7:] if k.startswith('module.') else k): v for k, v in ck['state_dict'].items()})
        optimizer.load_state_dict(ck['optimizer'])
        if 'engine' in ck:
            model.backbone.engine.load_state_dict(ck['engine'])
        lr_scheduler.load_state_dict(ck['lr_scheduler'])
        best_perf = float(ck.get('best_perf', -1.0))
        begin_epoch = max(begin_epoch, int(ck['epoch']) + 1)
        logger.info('=> resumed from %s: next epoch %d, best top-1 %.4f', resume, begin_epoch, best_perf)
    for epoch in range(begin_epoch, cfg.TRAIN.END_EPOCH):
        if world > 1:
            train_loader.sampler.set_epoch(epoch)
        train(cfg, train_loader, net, criterion, optimizer, epoch)
        lr_scheduler.step()  # after the epoch's optimizer steps (the reference steps it first, train.py:81)
        perf_indicator = validate(cfg, valid_loader, net, criterion, epoch)
        if perf_indicator > best_perf:
            best_perf = perf_indicator
            if rank == 0:
                path = os.path.join(cfg.OUTPUT_DIR, 'model_best.pth')
                torch.save({'state_dict': model.state_dict(), 'epoch': epoch, 'perf': best_perf}, path)
                logger.info('=> saved best model (top-1 %.4f) to %s', best_perf, path)
        if rank == 0:
            # written beside the old file and renamed over it: a kill during the write leaves the previous epoch's
            # checkpoint intact.  'engine' (the BatchNorm shifts) is rank 0's: the shifts are per replica, so under
            # data parallelism a resumed run repeats rank 0 bit for bit and the other ranks to rounding
            torch.save({'state_dict': model.state_dict(), 'optimizer': optimizer.state_dict(),
                        'lr_scheduler': lr_scheduler.state_dict(), 'epoch': epoch, 'best_perf': best_perf,
                        'engine': model.backbone.engine.state_dict()}, ckpt_path + '.tmp')
            os.replace(ckpt_path + '.tmp', ckpt_path)
    if world > 1:
        dist.destroy_process_group()
    return best_perf


if __name__ == '__main__':
    main()
