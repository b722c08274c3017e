#!/usr/bin/env python3
"""How long the host takes to ENQUEUE one training step (no synchronisation inside the step): if this is well
below the GPU step time the run is GPU-bound and one process per GPU scales without host contention."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-classification_amd"))
import _init_paths  # noqa
import torch
from core.config import config
from core.loss import Loss
from models.model import create_model
from utils.utils import create_optimizer
DEPTH = int(sys.argv[1]) if len(sys.argv) > 1 else 18  # python scripts/cpu_enqueue.py [18|50]
cfg = config.clone(); cfg.MODEL.RESNET_TYPE = DEPTH; cfg.MODEL.CLFDIM = 101 if DEPTH == 18 else 400; cfg.MODEL.INIT_WEIGHTS = False
cfg.TRAIN.OPTIMIZER = "sgd"; cfg.TRAIN.LR = 0.001
model = create_model(cfg, is_train=True).cuda(); crit = Loss(cfg).cuda(); opt = create_optimizer(cfg, model)
if os.environ.get("VC_DDP_FORCE_COMM") == "1":  # one-rank RCCL rehearsal of the data-parallel wrapper
    import torch.distributed as dist
    from vc3d.ddp import DataParallel3D
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    model = DataParallel3D(model, bucket_mb=float(os.environ.get("VC_BUCKET_MB", "32")))
x = torch.randn(64, 3, 16, 112, 112).to(torch.bfloat16).cuda(); y = torch.randint(0, cfg.MODEL.CLFDIM, (64,)).cuda()
model.train()
def step():
    _, s, _ = model(x); loss = crit(s, y, None, None); opt.zero_grad(); opt.arm_backward(); loss.backward(); opt.step()
for _ in range(5): step()
torch.cuda.synchronize()
ts = []
for _ in range(10):
    torch.cuda.synchronize(); t0 = time.perf_counter(); step(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    ts.append((t1 - t0, t2 - t0))
print("R3D-%d, replay %s:" % (DEPTH, "off" if os.environ.get("VC_NO_REPLAY") else "on"), end=" ")
print("host enqueue ms: median %.3f; step ms (enqueue + drain): median %.3f" % (1e3 * sorted(t[0] for t in ts)[5], 1e3 * sorted(t[1] for t in ts)[5]))
if os.environ.get("VC_ENQUEUE_PROFILE") == "1":  # where the host time of a step goes (cProfile, 20 steps)
    import cProfile, pstats
    pr = cProfile.Profile()
    torch.cuda.synchronize()
    pr.enable()
    for _ in range(20):
        step()
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(28)
