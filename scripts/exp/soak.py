"""Soak: a few thousand training steps (plain, or through the data-parallel wrapper on one rank with VC_DDP_FORCE_COMM=1):
device memory, host RSS and the step time must stay flat."""
import os, sys, time, resource
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "video-classification_amd"))
import _init_paths  # noqa
import torch
from core.config import config
from core.loss import Loss
from models.model import create_model
from utils.utils import create_optimizer
STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
cfg = config.clone(); cfg.MODEL.RESNET_TYPE = 18; cfg.MODEL.CLFDIM = 101; cfg.MODEL.INIT_WEIGHTS = False
cfg.TRAIN.OPTIMIZER = "sgd"; cfg.TRAIN.LR = 0.001
model = create_model(cfg, is_train=True).cuda(); crit = Loss(cfg).cuda(); opt = create_optimizer(cfg, model)
if os.environ.get("VC_DDP_FORCE_COMM") == "1":
    import torch.distributed as dist
    from vc3d.ddp import DataParallel3D
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29578")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    model = DataParallel3D(model)
x = torch.randn(32, 3, 16, 112, 112).to(torch.bfloat16).cuda(); y = torch.randint(0, 101, (32,)).cuda()
model.train()
def step():
    _, s, _ = model(x); loss = crit(s, y, None, None); opt.zero_grad(); opt.arm_backward(); loss.backward(); opt.step(); return loss
for _ in range(10): step()
torch.cuda.synchronize()
def snap(tag, t0, n):
    torch.cuda.synchronize()
    print("%s: %d steps, %.3f ms/step, device MB allocated %.0f reserved %.0f, host RSS MB %.0f" % (
        tag, n, 1e3 * (time.perf_counter() - t0) / max(n, 1), torch.cuda.memory_allocated() / 2**20,
        torch.cuda.memory_reserved() / 2**20, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024), flush=True)
for block in range(4):
    t0 = time.perf_counter()
    for _ in range(STEPS // 4): loss = step()
    snap("block %d (loss %.4f)" % (block, float(loss)), t0, STEPS // 4)
