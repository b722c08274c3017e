#!/usr/bin/env python3
"""BatchNorm pass micro-benchmark at the R3D-50 block-output sizes: bn_apply (+residual, ReLU, mask bits), the backward
reduction (mask 3 and mask 2) and the backward apply; us per launch and TB/s of algorithmic bytes."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "video-classification_amd"), os.path.join(ROOT, "video-classification_amd", "lib")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from vc3d import ops, capi  # noqa: E402
from vc3d.capi import lib, ptr, check  # noqa: E402


def timeit(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for rows, C in [(401408, 256), (50176, 512), (6272, 1024), (1024, 2048), (401408, 64), (50176, 128)]:
    n = rows * C
    y = torch.randn(rows, C, device="cuda").to(torch.bfloat16)
    dz = torch.randn(rows, C, device="cuda").to(torch.bfloat16)
    res = torch.randn(rows, C, device="cuda").to(torch.bfloat16)
    out = torch.empty_like(y)
    dy = torch.empty_like(y)
    gout = torch.empty_like(y)
    bits = torch.empty(n // 8, dtype=torch.uint8, device="cuda")
    params = torch.randn(4, C, device="cuda")
    params[1].abs_().add_(0.5)
    coef = torch.randn(2, C, device="cuda") * 0.01
    sums = ops.new_stats(C, y.device)
    s = ops.cur_stream()
    T = 2.0 * n
    res_ = []
    us = timeit(lambda: ops.bn_apply(y, params, C, residual=res, relu=True, out=out, relu_bits=bits))
    res_.append(("apply+res+bits", us, (3 * T + n / 8) / us / 1e6))
    us = timeit(lambda: ops.bn_apply(y, params, C, relu=True, out=out))
    res_.append(("apply", us, 2 * T / us / 1e6))
    us = timeit(lambda: check(lib.vc_bn_bwd_reduce(ptr(y), ptr(dz), ptr(bits), ptr(params), ptr(sums), rows, C, 3, s), "r"))
    res_.append(("reduce m3", us, (2 * T + n / 8) / us / 1e6))
    us = timeit(lambda: check(lib.vc_bn_bwd_reduce(ptr(y), ptr(dz), None, ptr(params), ptr(sums), rows, C, 2, s), "r"))
    res_.append(("reduce m2", us, 2 * T / us / 1e6))
    us = timeit(lambda: check(lib.vc_bn_bwd_apply(ptr(y), ptr(dz), ptr(bits), ptr(params), ptr(coef), ptr(dy), ptr(gout), n, C, 3, s), "a"))
    res_.append(("bwd apply m3+g", us, (4 * T + n / 8) / us / 1e6))
    us = timeit(lambda: check(lib.vc_bn_bwd_apply(ptr(y), ptr(dz), None, ptr(params), ptr(coef), ptr(dy), None, n, C, 2, s), "a"))
    res_.append(("bwd apply m2", us, 3 * T / us / 1e6))
    print("rows %6d C %4d (%5.1f MB): " % (rows, C, T / 1e6) + "  ".join("%s %.1f us %.2f TB/s" % r for r in res_), flush=True)
