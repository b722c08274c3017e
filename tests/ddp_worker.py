"""Worker of tests/test_ddp_gpu.py: one rank, backend nccl (RCCL), VC_DDP_FORCE_COMM=1.
Runs the same training step with and without DataParallel3D and prints the largest difference of
the flat gradient / parameter buffers (must be exactly 0: an AVG all-reduce over one rank is the identity)."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-classification_amd"))
import _init_paths  # noqa: F401,E402
from core.config import config  # noqa: E402
from core.loss import Loss  # noqa: E402
from models.model import create_model  # noqa: E402
from utils.utils import create_optimizer  # noqa: E402
from vc3d.ddp import DataParallel3D  # noqa: E402


def run(wrap):
    cfg = config.clone()
    cfg.MODEL.RESNET_TYPE = 18
    cfg.MODEL.CLFDIM = 11
    cfg.MODEL.INIT_WEIGHTS = False
    cfg.TRAIN.OPTIMIZER = "sgd"
    cfg.TRAIN.LR = 0.01
    torch.manual_seed(3)
    model = create_model(cfg, is_train=True).cuda()
    crit = Loss(cfg).cuda()
    opt = create_optimizer(cfg, model)
    net = DataParallel3D(model, bucket_mb=4) if wrap else model
    g = torch.Generator().manual_seed(5)
    x = torch.randn(8, 3, 8, 64, 64, generator=g).to(torch.bfloat16).cuda()
    y = torch.randint(0, 11, (8,), generator=g).cuda()
    net.train()
    for _ in range(1):  # one step: the fp32-atomic noise of the BatchNorm sums is not yet amplified by training
        _, score, _ = net(x)
        loss = crit(score, y, None, None)
        opt.zero_grad()
        loss.backward()
        opt.step()
    torch.cuda.synchronize()
    eng = model.backbone.engine
    nb = len(net.reducer.buckets) if wrap else 0
    return eng.flat_grad.clone(), eng.flat_param.clone(), float(loss), nb


def main():
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    g0, p0, l0, _ = run(False)
    g2, p2, l2, _ = run(False)
    g1, p1, l1, nb = run(True)
    rel = lambda a, b: float((a - b).norm() / b.norm().clamp_min(1e-30))
    print("DDP_RESULT grad_rel=%g param_rel=%g rerun_grad_rel=%g rerun_param_rel=%g loss=%g/%g buckets=%d finite=%d" % (
        rel(g1, g0), rel(p1, p0), rel(g2, g0), rel(p2, p0), l0, l1, nb, int(torch.isfinite(g1).all())), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
