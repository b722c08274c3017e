#!/usr/bin/env python3
"""Validation entry point with the reference's flow (/root/reference/valid.py:28-61; that file does
not parse): logger -> create_model(is_train=False) -> (optional checkpoint) -> loader -> validate().

Multi-clip protocol (build-defined; the reference has none): each video
yields TEST.NUM_CLIPS x TEST.NUM_CROPS views of 3 x CLIP_LEN x CROP x CROP; `validate` averages
the views' logits before the argmax.  Forward only, eval-mode BatchNorm (running statistics
folded into one scale/shift per channel).  One process per GPU; accuracy counts are summed over
ranks.  Overrides: KEY=VALUE pairs, plus CKPT=path for a checkpoint written by train.py.
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
from train import apply_overrides, fan_out_if_needed, setup_distributed


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    ckpt = None
    for a in list(argv):
        if a.startswith("CKPT="):
            ckpt = a[5:]
            argv.remove(a)
    cfg = apply_overrides(config.clone(), argv)
    # config.GPUS with several devices: one rank per device, as the reference's DataParallel over gpus (valid.py:42-43)
    code = fan_out_if_needed(cfg, os.path.abspath(__file__), argv + (["CKPT=" + ckpt] if ckpt else []))
    if code is not None:
        if code:
            raise SystemExit(code)
        return None
    if os.environ.get(launch.DRY_ENV) == "1":
        launch.dry_child_report()
        return None
    import torch
    import torch.distributed as dist
    import torch.utils.data
    from core.function import validate
    from core.loss import Loss
    from dataset.dataset import MultiViewClips
    from models.model import create_model
    from utils.utils import create_logger
    world, rank, dev = setup_distributed(cfg)
    create_logger(cfg, 'valid')
    model = create_model(cfg, is_train=False).to(dev)
    if ckpt:
        sd = torch.load(ckpt, map_location='cpu', weights_only=True)
        sd = sd.get('state_dict', sd)
        model.load_state_dict({(k[7:] if k.startswith('module.') else k): v for k, v in sd.items()})
    criterion = Loss(cfg).to(dev)
    ds = MultiViewClips(cfg.get("SYNTH_VAL_LEN") or 64, cfg.TEST.NUM_CLIPS * cfg.TEST.NUM_CROPS, cfg.MODEL.CLIP_LEN,
                        cfg.MODEL.CROP_SIZE, cfg.MODEL.CLFDIM)
    sampler = torch.utils.data.distributed.DistributedSampler(ds, world, rank, shuffle=False) if world > 1 else None
    loader = torch.utils.data.DataLoader(ds, batch_size=max(1, cfg.TEST.BATCH_SIZE // ds.views), shuffle=False,
                                         sampler=sampler, num_workers=cfg.WORKERS, pin_memory=True)
    perf = validate(cfg, loader, model, criterion, 0)
    if world > 1:
        dist.destroy_process_group()
    return perf


if __name__ == '__main__':
    main()
