"""GPU check: find the first layer whose tensors differ between two runs of the same forward (scripts/, not a test)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "video-classification_amd"))
import _init_paths  # noqa
from ddp_worker import build, batch, rel

n = int(os.environ.get("NB", "4"))
x, y = batch()
ref = None
for rep in range(6):
    cfg, model, crit, opt = build()
    model.train()
    eng = model.backbone.engine
    eng.trace = {}
    with torch.no_grad():
        _, score, _ = model(x[:n])
    torch.cuda.synchronize()
    tr = {k: {kk: vv.detach().float().cpu().clone() for kk, vv in v.items() if torch.is_tensor(vv)} for k, v in eng.trace.items()}
    if ref is None:
        ref = tr
        order = list(tr.keys())
        continue
    first = None
    for k in order:
        for kk in ref[k]:
            a, b = tr[k][kk], ref[k][kk]
            d = float((a - b).abs().max())
            if d > 0 and first is None:
                first = (k, kk, d, rel(a, b), int((a != b).sum()), a.numel())
    print("rep", rep, "first difference:", first, "logits rel %.3g" % rel(tr["final"]["feat"], ref["final"]["feat"]), flush=True)
    if first is not None and first[1] == "y":
        a, b = tr[first[0]]["y"], ref[first[0]]["y"]
        idx = (a != b).nonzero()
        print("   differing y entries: n-range", idx[:, 0].unique().tolist(), "t", idx[:, 1].unique().tolist(), "h", idx[:, 2].unique().tolist()[:12],
              "w", idx[:, 3].unique().tolist()[:12], "c-range", int(idx[:, 4].min()), int(idx[:, 4].max()), "shape", tuple(a.shape))
