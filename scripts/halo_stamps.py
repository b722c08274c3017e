#!/usr/bin/env python3
"""Tuning aid: per-phase clock stamps of the halo conv kernel (vc_debug_halo_stamps)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "video-classification_amd")):
    sys.path.insert(0, p)
import torch
from vc3d import ops, capi
N, Ci, T, H, W, Co = [int(v) for v in sys.argv[1:7]]
if len(sys.argv) > 7:
    os.environ["VC_HALO_FORCE"] = sys.argv[7]
d = ops.conv_desc(N, Ci, T, H, W, Co, 3, 1, 1)
x = torch.randn(N, T, H, W, Ci, device="cuda").to(torch.bfloat16)
w = (torch.randn(Co, Ci, 3, 3, 3, device="cuda") * 0.05).contiguous(memory_format=torch.channels_last_3d)
wf, wd = ops.pack_weight(w)
stats = ops.new_stats(Co, x.device)
for _ in range(3):
    y = ops.conv3d_fwd(d, x, wf, stats=stats)
buf = torch.zeros(8192 * 8, dtype=torch.int64, device="cuda")
raw = ctypes.CDLL(capi.LIB_PATH)
raw.vc_debug_halo_stamps.argtypes = [ctypes.c_void_p]
raw.vc_debug_halo_stamps(buf.data_ptr())
y = ops.conv3d_fwd(d, x, wf, stats=stats)
torch.cuda.synchronize()
raw.vc_debug_halo_stamps(None)
b = buf.view(-1, 8).cpu()
b = b[b[:, 0] != 0]
t0 = b[:, 0].min()
names = ["masks", "issue c0", "c0 landed", "c0 tap loop", "c1 issue+landed", "c1 tap loop", "epilogue"]
print("workgroups", b.shape[0])
two = bool((b[:, 5] != 0).any())
for k in range(7):
    if not two and k in (4, 5):
        continue
    lo = b[:, k] if (two or k != 6) else b[:, 4]
    dlt = (b[:, k + 1] - lo).double()
    print("%-16s median %8.0f  mean %8.0f  max %8.0f" % (names[k], dlt.median(), dlt.mean(), dlt.max()))
tot = (b[:, 7] - b[:, 0]).double()
print("per-WG total     median %8.0f  mean %8.0f" % (tot.median(), tot.mean()))
