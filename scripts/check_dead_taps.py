"""GPU check: weight gradients of taps that only ever read zero padding stay exactly 0 in the split (workspace) form,
whatever the workspace held before (scripts/, not a test: run on the GPU box)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-classification_amd"))
from vc3d import ops  # noqa: E402

torch.manual_seed(0)
for (N, Ci, T, H, W, Co, k, s, p) in [(64, 512, 1, 4, 4, 512, 3, 1, 1), (64, 256, 2, 7, 7, 512, 3, 2, 1), (8, 64, 1, 9, 9, 64, 3, 1, 1)]:
    d = ops.conv_desc(N, Ci, T, H, W, Co, k, s, p)
    To, Ho, Wo = d.out_dims()
    x = torch.randn(N, T, H, W, Ci, device="cuda").bfloat16()
    dy = torch.randn(N, To, Ho, Wo, Co, device="cuda").bfloat16()
    need = int(ops.lib.vc_conv3d_workspace_bytes(d, 2))
    # poison whatever workspace the wrapper will hand out
    ops._WS.clear()
    ws, nb = ops._workspace(d, 2, x.device)
    if ws is not None:
        list(ops._WS.values())[0].view(torch.float32).fill_(1.0e3)
    dw = torch.zeros(Co, k, k, k, Ci, device="cuda")
    ops.conv3d_wgrad(d, x, dy, dw)
    ref = torch.nn.grad.conv3d_weight(x.permute(0, 4, 1, 2, 3).float(), (Co, Ci, k, k, k), dy.permute(0, 4, 1, 2, 3).float(),
                                      s, p).permute(0, 2, 3, 4, 1)
    err = float((dw - ref).norm() / ref.norm())
    dead = ref.abs().amax(dim=(0, 4)) == 0
    print("geom", (N, Ci, T, H, W, Co, k, s), "workspace MiB %.1f" % (need / 2 ** 20), "rel err %.3e" % err,
          "dead taps", int(dead.sum()), "max |dw| on dead taps %.3e" % float(dw[:, dead].abs().max() if dead.any() else 0))
