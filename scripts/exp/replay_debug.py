import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "video-classification_amd"), os.path.join(ROOT, "video-classification_amd", "lib")):
    sys.path.insert(0, p)
import torch
from helpers import make_config
import vc3d.engine as eng_mod
from vc3d import ops
from core.loss import Loss
from models.model import create_model
from utils.utils import create_optimizer
ops.set_deterministic(True)
def run(replay, nsteps):
    eng_mod.REPLAY = replay
    cfg = make_config(18, 11, OPTIMIZER="sgd", LR=0.01)
    torch.manual_seed(3)
    model = create_model(cfg, is_train=True).cuda().train()
    crit, opt = Loss(cfg).cuda(), create_optimizer(cfg, model)
    g = torch.Generator().manual_seed(5)
    for i in range(nsteps):
        x = torch.randn(4, 3, 8, 64, 64, generator=g).to(torch.bfloat16).cuda()
        y = torch.randint(0, 11, (4,), generator=g).cuda()
        _, score, _ = model(x)
        loss = crit(score, y, None, None)
        opt.zero_grad(); loss.backward()
        if i < nsteps - 1: opt.step()
    torch.cuda.synchronize()
    eng = model.backbone.engine
    names = {id(p): n for n, p in model.named_parameters()}
    return {names[id(p)]: eng.flat_grad[off:off + n].clone() for p, off, n in eng.grad_ranges}, score.detach().clone(), model
for nsteps in (2, 3):
    a, sa, _ = run(False, nsteps); b, sb, m = run(True, nsteps)
    print("nsteps", nsteps, "logits equal", torch.equal(sa, sb), "tape", m.backbone.engine._tape and m.backbone.engine._tape["serial"])
    bad = [(k, float((a[k] - b[k]).norm() / a[k].norm().clamp_min(1e-30))) for k in a if not torch.equal(a[k], b[k])]
    print("  differing grads:", len(bad), "of", len(a), bad[:12])
