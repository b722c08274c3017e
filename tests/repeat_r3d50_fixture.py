"""Run-to-run spread of the quantities test_golden_gpu.py::test_r3d50_fixture_geometries bounds (the product's BatchNorm
sums are float atomics: runs differ in the last bits, and a 2-clip R3D-50 at random initialisation amplifies that).
usage (GPU box, repo root): python tests/repeat_r3d50_fixture.py [runs] [full|long|r18]"""
import os
import sys
import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
for _d in ("video-classification_amd", os.path.join("video-classification_amd", "lib")):
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", _d))
from helpers import build_pair, rel_l2  # noqa: E402
from test_golden_gpu import load_golden  # noqa: E402
from oracle.r3d_torch import synthetic_batch  # noqa: E402


def r18(runs):
    """test_r3d18_full_geometry_logits_loss_and_gradient_norms: worst per-parameter gradient-norm deviation"""
    g = load_golden("r3d18_full")
    x, y = synthetic_batch(2, 16, 112, 112, 101, 2468)
    for r in range(runs):
        _, prod, _ = build_pair(18, 101)
        prod.train()
        _, lo, _ = prod(x.cuda())
        loss = F.cross_entropy(lo, y.cuda())
        loss.backward()
        torch.cuda.synchronize()
        worst = {}
        for n, p in prod.named_parameters():
            ref = float(g["gnorm/" + n.replace("clf_haed.", "clf_head.")])
            kind = "bn" if (".bn" in n or "downsample.1" in n) else "w"
            worst[kind] = max(worst.get(kind, 0.0), abs(float(p.grad.double().norm()) - ref) / max(ref, 1e-12))
        print("run %d: logits %.4f  loss %.4f  worst w %.4f  worst bn %.4f"
              % (r, rel_l2(lo.detach(), g["logits"]), abs(float(loss.detach()) - g["loss"]) / g["loss"], worst["w"], worst["bn"]), flush=True)


def main():
    runs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    if len(sys.argv) > 2 and sys.argv[2] == "r18":
        return r18(runs)
    long_ = len(sys.argv) > 2 and sys.argv[2] == "long"
    g = load_golden("r3d50_long" if long_ else "r3d50_full")
    x, y = synthetic_batch(1, 64, 224, 224, 400, 9753) if long_ else synthetic_batch(2, 16, 112, 112, 400, 1357)
    e_emu = rel_l2(g["emu/logits"], g["logits"])
    d_emu = abs(g["emu/loss"] - g["loss"]) / g["loss"]
    b_emu = rel_l2(g["emu/grad/clf_head.bias"], g["grad/clf_head.bias"])
    print("bars: logits %.4f  loss %.4f  bias-grad %.4f" % (1.25 * e_emu + 1e-2, max(2e-2, 1.5 * d_emu + 5e-3), 1.5 * b_emu + 1e-2))
    for r in range(runs):
        _, prod, _ = build_pair(50, 400)
        prod.train()
        _, lo, _ = prod(x.cuda())
        loss = F.cross_entropy(lo, y.cuda())
        loss.backward()
        torch.cuda.synchronize()
        dev, ratio = {"w": [], "bn": []}, {"w": [], "bn": []}
        for nme, p in prod.named_parameters():
            ref = float(g["gnorm/" + nme.replace("clf_haed.", "clf_head.")])
            kind = "bn" if (".bn" in nme or "downsample.1" in nme) else "w"
            dev[kind].append(abs(float(p.grad.double().norm()) - ref) / max(ref, 1e-12))
            ratio[kind].append(float(p.grad.double().norm()) / max(ref, 1e-12))
        q = {k: (float(np.median(v)), float(np.percentile(v, 90))) for k, v in dev.items()}
        sc = float(np.median(ratio["w"]))
        rel = {k: np.abs(np.array(v) / sc - 1.0) for k, v in ratio.items()}
        print("        common scale %.4f; relative to it: w (%.3f, %.3f, max %.3f)  bn (%.3f, %.3f, max %.3f)"
              % (sc, np.median(rel["w"]), np.percentile(rel["w"], 90), rel["w"].max(), np.median(rel["bn"]),
                 np.percentile(rel["bn"], 90), rel["bn"].max()))
        gb = dict(prod.named_parameters())["clf_haed.bias"].grad
        print("run %d: logits %.4f  loss %.4f  w (%.3f, %.3f)  bn (%.3f, %.3f)  bias-grad %.4f"
              % (r, rel_l2(lo.detach(), g["logits"]), abs(float(loss.detach()) - g["loss"]) / g["loss"], q["w"][0], q["w"][1],
                 q["bn"][0], q["bn"][1], rel_l2(gb, g["grad/clf_head.bias"])), flush=True)


main()
