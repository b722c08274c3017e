"""spread of the frozen-statistics gradient check (tests/test_net_gpu.py) over seeds: is a 0.2 norm bar inside the noise?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = os.path.join(ROOT, "video-classification_amd")
for q in (ROOT, PKG, os.path.join(PKG, "lib"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, q)
import torch
import torch.nn.functional as F
from helpers import build_pair
from oracle.r3d_torch import synthetic_batch

for seed in (21, 22, 23, 24, 25, 26):
    torch.manual_seed(seed)
    oracle, prod, _ = build_pair(50, 101)
    x, y = synthetic_batch(4, 4, 64, 64, 101, seed)
    oracle.train(); prod.train()
    with torch.no_grad():
        oracle(x); prod(x.cuda())
    oracle.eval(); prod.eval()
    F.cross_entropy(oracle(x), y).backward()
    _, lo, _ = prod(x.cuda())
    F.cross_entropy(lo, y.cuda()).backward()
    torch.cuda.synchronize()
    g_ref = {n: p.grad for n, p in oracle.named_parameters()}
    worst, wn = 0.0, None
    for n, p in prod.named_parameters():
        r = g_ref[n.replace("clf_haed.", "clf_head.")]
        d = abs(float(p.grad.float().norm().cpu() / r.norm()) - 1.0)
        if d > worst:
            worst, wn = d, n
    print("seed %d worst norm deviation %.3f at %s" % (seed, worst, wn), flush=True)
