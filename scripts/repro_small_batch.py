"""GPU check: run-to-run reproducibility of one training step at small batch (scripts/, not a test)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "video-classification_amd"))
import _init_paths  # noqa
from ddp_worker import build, batch, rel, per_param

x, y = batch()
for n in (8, 4, 2):
    outs = []
    for rep in range(3):
        cfg, model, crit, opt = build()
        model.train()
        opt.zero_grad()
        eng = model.backbone.engine
        eng.trace = {} if os.environ.get("TRACE") else None
        _, score, _ = model(x[:n])
        crit(score, y[:n], None, None).backward()
        torch.cuda.synchronize()
        outs.append((score.detach().float().clone(), eng.flat_grad.clone(), eng, model))
    for rep in (1, 2):
        worst, med = per_param(outs[0][2], outs[0][3], outs[rep][1], outs[0][1])
        print("N=%d rep %d: logits rel %.3g grad rel %.3g worst %s median %.3g" % (
            n, rep, rel(outs[rep][0], outs[0][0]), rel(outs[rep][1], outs[0][1]), ["%s=%.2g" % (nm, v) for v, nm in worst[:3]], med[0]))
