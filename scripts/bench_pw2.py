#!/usr/bin/env python3
"""Pointwise (1x1x1) conv micro-benchmark at the R3D-50 bench geometries: forward (with / without the fused BatchNorm
sums and finalize), dgrad (plain / accumulating), wgrad.  Prints us per launch and the HBM rate of the algorithmic
bytes (x + w + y, + the old output when accumulating)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "video-classification_amd"), os.path.join(ROOT, "video-classification_amd", "lib")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from vc3d import ops  # noqa: E402


def timeit(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


SHAPES = [(64, 8, 28, 28, 256), (256, 8, 28, 28, 64), (128, 4, 14, 14, 512), (512, 4, 14, 14, 128),
          (256, 2, 7, 7, 1024), (1024, 2, 7, 7, 256), (512, 1, 4, 4, 2048), (2048, 1, 4, 4, 512)]


def main():
    sel = sys.argv[1:] or ["fwd", "fwd_stats", "fwd_fin", "dgrad", "dgrad_acc", "wgrad"]
    N = 64
    for Ci, T, H, W, Co in SHAPES:
        d = ops.conv_desc(N, Ci, T, H, W, Co, 1, 1, 0)
        M = N * T * H * W
        x = torch.randn(N, T, H, W, Ci, device="cuda").to(torch.bfloat16)
        dy = torch.randn(N, T, H, W, Co, device="cuda").to(torch.bfloat16)
        w = (torch.randn(Co, Ci, 1, 1, 1, device="cuda") * 0.05).contiguous(memory_format=torch.channels_last_3d)
        wf, wd = ops.pack_weight(w)
        dw = torch.zeros(Co, 1, 1, 1, Ci, device="cuda")
        y = torch.empty(N, T, H, W, Co, device="cuda", dtype=torch.bfloat16)
        dx = torch.zeros_like(x)
        stats = ops.new_stats(Co, x.device)
        g, b = torch.ones(Co, device="cuda"), torch.zeros(Co, device="cuda")
        rm, rv = torch.zeros(Co, device="cuda"), torch.ones(Co, device="cuda")
        ticket = torch.zeros(64, dtype=torch.int32, device="cuda")
        params = torch.empty(4 * Co, device="cuda")
        fin, _ = ops.bn_fin(ticket, M, g, b, rm, rv, 0.1, 1e-5, params=params)
        by = 2.0 * (M * Ci + M * Co + Ci * Co)
        line = "Ci %4d Co %4d M %6d :" % (Ci, Co, M)
        for s in sel:
            if s == "fwd":
                us, byt = timeit(lambda: ops.conv3d_fwd(d, x, wf, out=y)), by
            elif s == "fwd_stats":
                us, byt = timeit(lambda: ops.conv3d_fwd(d, x, wf, stats=stats, out=y)), by
            elif s == "fwd_fin":
                us, byt = timeit(lambda: ops.conv3d_fwd(d, x, wf, stats=stats, out=y, fin=fin)), by
            elif s == "dgrad":
                us, byt = timeit(lambda: ops.conv3d_dgrad(d, dy, wd, out=dx)), by
            elif s == "dgrad_acc":
                us, byt = timeit(lambda: ops.conv3d_dgrad(d, dy, wd, out=dx, accumulate=True)), by + 2.0 * M * Ci
            elif s == "wgrad":
                us, byt = timeit(lambda: ops.conv3d_wgrad(d, x, dy, dw)), by
            line += "  %s %6.1f us %5.2f TB/s" % (s, us, byt / us / 1e6)
        print(line, flush=True)


if __name__ == "__main__":
    main()
