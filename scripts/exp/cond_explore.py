import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "video-classification_amd"), os.path.join(ROOT, "video-classification_amd", "lib")):
    sys.path.insert(0, p)
import torch, torch.nn.functional as F
from helpers import build_pair
from test_determinism_gpu import _structured_batch
from utils.utils import create_optimizer
from vc3d import ops
ops.set_deterministic(True)
for amp in (0.1, 0.2):
    oracle, prod, cfg = build_pair(18, 16)
    cfg.TRAIN.OPTIMIZER, cfg.TRAIN.LR, cfg.TRAIN.MOMENTUM, cfg.TRAIN.WD = "sgd", 0.005, 0.9, 1e-4
    opt = create_optimizer(cfg, prod); prod.train()
    bs = [_structured_batch(16, 8, 64, 16, 100 + i, amp) for i in range(10)]
    bs = [(x.to(torch.bfloat16).cuda(), y.cuda()) for x, y in bs]
    xv, yv = _structured_batch(16, 8, 64, 16, 777, amp); xv, yv = xv.to(torch.bfloat16).cuda(), yv.cuda()
    out = []
    for step in range(100):
        x, y = bs[step % 10]
        _, sc, _ = prod(x); loss = F.cross_entropy(sc, y); opt.zero_grad(); loss.backward(); opt.step()
        if step % 5 == 4:
            prod.eval()
            with torch.no_grad(): lv = F.cross_entropy(prod(xv)[1], yv)
            prod.train()
            out.append("%d:%.2f/%.2f" % (step + 1, float(loss), float(lv)))
    print("amp", amp, " ".join(out), flush=True)
