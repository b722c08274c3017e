#!/usr/bin/env python3
"""Tuning aid: per-phase clocks of the wgrad9 kernel (vc_debug_wgrad9_stamps).  The default build records the loop total
only; the per-phase split needs wgrad9_conv.hip built with -DVC_W9_STAMPS (which puts the phases back into blocks of
their own)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "video-classification_amd")):
    sys.path.insert(0, p)
import torch
from vc3d import ops, capi
N, Ci, T, H, W, Co = [int(v) for v in sys.argv[1:7]]
d = ops.conv_desc(N, Ci, T, H, W, Co, 3, 1, 1)
x = torch.randn(N, T, H, W, Ci, device="cuda").to(torch.bfloat16)
dy = torch.randn(N, T, H, W, Co, device="cuda").to(torch.bfloat16)
dw = torch.zeros(Co, 3, 3, 3, Ci, device="cuda")
for _ in range(3):
    ops.conv3d_wgrad(d, x, dy, dw)
buf = torch.zeros(4096 * 8, dtype=torch.int64, device="cuda")
raw = ctypes.CDLL(capi.LIB_PATH)
raw.vc_debug_wgrad9_stamps.argtypes = [ctypes.c_void_p]
raw.vc_debug_wgrad9_stamps(buf.data_ptr())
ops.conv3d_wgrad(d, x, dy, dw)
torch.cuda.synchronize()
raw.vc_debug_wgrad9_stamps(None)
b = buf.view(-1, 8).cpu()
b = b[b[:, 5] != 0].double()
print("workgroups", b.shape[0], "chunks per WG", b[:, 5].median().item())
for k, nm in enumerate(["issue loads", "compute", "store to LDS", "barrier"]):
    per = b[:, k] / b[:, 5]
    print("%-14s per chunk: median %7.0f  max %7.0f" % (nm, per.median(), per.max()))
print("loop total per WG: median %.0f  max %.0f;  start spread %.0f" % (b[:, 4].median(), b[:, 4].max(), (b[:, 6].max() - b[:, 6].min())))
