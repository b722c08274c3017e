import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "video-classification_amd"))
import _init_paths  # noqa
import torch
from core.config import config
from core.loss import Loss
from models.model import create_model
from utils.utils import create_optimizer
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 18
cfg = config.clone(); cfg.MODEL.RESNET_TYPE = depth; cfg.MODEL.CLFDIM = 101; cfg.MODEL.INIT_WEIGHTS = False
cfg.TRAIN.OPTIMIZER = "sgd"; cfg.TRAIN.LR = 0.001
model = create_model(cfg, is_train=True).cuda(); crit = Loss(cfg).cuda(); opt = create_optimizer(cfg, model)
x = torch.randn(64, 3, 16, 112, 112).to(torch.bfloat16).cuda(); y = torch.randint(0, 101, (64,)).cuda()
model.train()
def step():
    _, s, _ = model(x); loss = crit(s, y, None, None); opt.zero_grad(); loss.backward(); opt.step()
for _ in range(5): step()
torch.cuda.synchronize()
t = model.backbone.engine._tape
import collections
for name in ("fwd", "bwd"):
    L = t[name]
    ts = []
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter(); L.replay(); ts.append(time.perf_counter() - t0)
    print(name, "commands", L.n, "replay host ms", ["%.3f" % (1e3 * v) for v in ts])
    # per-entry-point cost: replay each command alone 20 times
    from vc3d import capi
    import ctypes as C
    cost = collections.defaultdict(lambda: [0, 0.0])
    for i in range(L.n):
        one = (capi.Cmd * 1)(); C.memmove(one, C.byref(L.array[i]), C.sizeof(capi.Cmd))
        if one[0].kind != 0: continue
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): capi.lib.vc_replay(one, 1, L.events, L.n_events, -1, None, None, None)
        dt = (time.perf_counter() - t0) / 10
        cost[L.names[i]][0] += 1; cost[L.names[i]][1] += dt
    for k, (n, s) in sorted(cost.items(), key=lambda kv: -kv[1][1]):
        print("   %-32s x%3d  %7.1f us total  %5.1f us each" % (k, n, 1e6 * s, 1e6 * s / n))
torch.cuda.synchronize()
