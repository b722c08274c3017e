#!/usr/bin/env python3
"""Stem forward / weight gradient alone (64 x 3 x 16 x 112 x 112 by default), for kernel tuning and the ablation builds
(VC3D_LIB=... selects a variant library).  python scripts/bench_stem.py [N T H W]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "video-classification_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from vc3d import ops  # noqa: E402

N, T, H, W = [int(v) for v in sys.argv[1:5]] if len(sys.argv) >= 5 else (64, 16, 112, 112)
x = torch.randn(N, 3, T, H, W, device="cuda").to(torch.bfloat16)
w = (torch.randn(64, 3, 7, 7, 7, device="cuda") * 0.05).contiguous(memory_format=torch.channels_last_3d)
xs = ops.stem_pack_input(x)
ws = ops.stem_pack_weight(w)
stats = ops.new_stats(64, "cuda")
y = torch.empty(N, T, H // 2, W // 2, 64, dtype=torch.bfloat16, device="cuda")


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


flops = 2.0 * N * T * (H // 2) * (W // 2) * 64 * 1029
ms = timeit(lambda: ops.stem_fwd(xs, ws, N, T, H, W, 64, stats=stats, out=y))
print("stem fwd (sums)  %.3f ms  %.0f TFLOP/s" % (ms, flops / ms / 1e9))
ms = timeit(lambda: ops.stem_fwd(xs, ws, N, T, H, W, 64, out=y))
print("stem fwd         %.3f ms  %.0f TFLOP/s" % (ms, flops / ms / 1e9))
