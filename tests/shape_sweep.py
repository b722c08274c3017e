"""Ad-hoc robustness sweep over clip shapes (not collected by pytest): one training step + eval logits against the oracle.
usage (GPU box, repo root): python tests/shape_sweep.py"""
import os
import sys
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, ".."))
for _d in ("video-classification_amd", os.path.join("video-classification_amd", "lib")):
    sys.path.insert(0, os.path.join(HERE, "..", _d))
from helpers import build_pair, rel_l2  # noqa: E402
from oracle.r3d_torch import synthetic_batch  # noqa: E402

SHAPES = [(18, 1, 1, 32, 32), (18, 1, 3, 34, 50), (18, 2, 7, 98, 130), (18, 5, 2, 66, 34), (18, 7, 4, 48, 80),
          (18, 1, 16, 112, 112), (50, 1, 1, 32, 32), (50, 3, 5, 50, 34), (50, 2, 3, 96, 160), (18, 1, 2, 16, 16)]
for depth, n, t, h, w in SHAPES:
    try:
        oracle, prod, _ = build_pair(depth, 101)
        x, y = synthetic_batch(n, t, h, w, 101, 11)
        oracle.train(), prod.train()
        lo_ref = oracle(x)
        loss_ref = F.cross_entropy(lo_ref, y)
        _, lo, _ = prod(x.cuda())
        loss = F.cross_entropy(lo, y.cuda())
        loss.backward()
        torch.cuda.synchronize()
        fin = all(bool(torch.isfinite(p.grad).all()) for p in prod.parameters())
        oracle.eval(), prod.eval()
        with torch.no_grad():
            e_eval = rel_l2(prod(x.cuda())[1], oracle(x))
        print("R3D-%d %dx%dx%dx%d: train logits %.3e loss %.4f/%.4f grads finite %s eval logits %.3e"
              % (depth, n, t, h, w, rel_l2(lo.detach(), lo_ref.detach()), float(loss.detach()), float(loss_ref), fin, e_eval), flush=True)
    except Exception as e:  # noqa: BLE001
        print("R3D-%d %dx%dx%dx%d: %s: %s" % (depth, n, t, h, w, type(e).__name__, str(e)[:200]), flush=True)
