#!/usr/bin/env python3
"""Micro-benchmark of one conv shape through the C ABI (forward / dgrad / wgrad), for kernel tuning.
  python scripts/bench_conv.py N Ci T H W Co k s p [--force "NW,TT,SEG,splits" ...]
Prints ms and TFLOP/s per variant (VC_HALO_FORCE is read by the planner at every call)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "video-classification_amd"), os.path.join(ROOT, "video-classification_amd", "lib")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from vc3d import ops  # noqa: E402


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


def main():
    a = sys.argv[1:]
    geom = [int(v) for v in a[:9]]
    forces = [None]
    ops_sel = "fwd,dgrad,wgrad"
    no_stats = False
    i = 9
    while i < len(a):
        if a[i] == "--force":
            forces.append(a[i + 1])
            i += 2
        elif a[i] == "--ops":
            ops_sel = a[i + 1]
            i += 2
        elif a[i] == "--no-stats":
            no_stats = True
            i += 1
        else:
            raise SystemExit("bad arg " + a[i])
    N, Ci, T, H, W, Co, k, s, p = geom
    d = ops.conv_desc(N, Ci, T, H, W, Co, k, s, p)
    To, Ho, Wo = d.out_dims()
    x = torch.randn(N, T, H, W, Ci, device="cuda").to(torch.bfloat16)
    dy = torch.randn(N, To, Ho, Wo, Co, device="cuda").to(torch.bfloat16)
    w = (torch.randn(Co, Ci, k, k, k, device="cuda") * 0.05).contiguous(memory_format=torch.channels_last_3d)
    wf, wd = ops.pack_weight(w)
    dw = torch.zeros(Co, k, k, k, Ci, device="cuda")
    stats = None if no_stats else ops.new_stats(Co, x.device)
    flops = 2.0 * N * To * Ho * Wo * Co * Ci * k ** 3
    y = torch.empty(N, To, Ho, Wo, Co, device="cuda", dtype=torch.bfloat16)
    dx = torch.empty_like(x)
    for f in forces:
        if f is None:
            os.environ.pop("VC_HALO_FORCE", None)
        else:
            os.environ["VC_HALO_FORCE"] = f
        line = "%-16s" % (f or "planner")
        if "fwd" in ops_sel:
            ms = timeit(lambda: ops.conv3d_fwd(d, x, wf, stats=stats, out=y))
            line += "  fwd %.3f ms %6.0f TF" % (ms, flops / ms / 1e9)
        if "dgrad" in ops_sel:
            ms = timeit(lambda: ops.conv3d_dgrad(d, dy, wd, out=dx))
            line += "  dgrad %.3f ms %6.0f TF" % (ms, flops / ms / 1e9)
        if "dacc" in ops_sel:  # accumulating dgrad (dx += ...)
            ms = timeit(lambda: ops.conv3d_dgrad(d, dy, wd, out=dx, accumulate=True))
            line += "  dgrad+= %.3f ms %6.0f TF" % (ms, flops / ms / 1e9)
        if "fwdns" in ops_sel:  # forward without the BatchNorm sums
            ms = timeit(lambda: ops.conv3d_fwd(d, x, wf, out=y))
            line += "  fwd(no sums) %.3f ms %6.0f TF" % (ms, flops / ms / 1e9)
        if "wgrad" in ops_sel:
            ms = timeit(lambda: ops.conv3d_wgrad(d, x, dy, dw))
            line += "  wgrad %.3f ms %6.0f TF" % (ms, flops / ms / 1e9)
        print(line, flush=True)


if __name__ == "__main__":
    main()
