#!/usr/bin/env python3
"""Host-side cost of one forward-only (eval) step: enqueue time against the drained step, and a cProfile of the enqueue.
  python scripts/host_eval_profile.py [depth] [classes]"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-classification_amd"))
import _init_paths  # noqa
import torch
from core.config import config
from models.model import create_model

depth = int(sys.argv[1]) if len(sys.argv) > 1 else 50
cfg = config.clone()
cfg.MODEL.RESNET_TYPE = depth
cfg.MODEL.CLFDIM = int(sys.argv[2]) if len(sys.argv) > 2 else 400
cfg.MODEL.INIT_WEIGHTS = False
model = create_model(cfg, is_train=True).cuda().eval()
x = torch.randn(64, 3, 16, 112, 112).to(torch.bfloat16).cuda()


def step():
    with torch.no_grad():
        model(x)


for _ in range(5):
    step()
torch.cuda.synchronize()
ts = []
for _ in range(20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    ts.append((t1 - t0, time.perf_counter() - t0))
print("host enqueue ms: median %.3f; step ms (enqueue + drain): median %.3f" % (
    1e3 * sorted(t[0] for t in ts)[10], 1e3 * sorted(t[1] for t in ts)[10]))
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
