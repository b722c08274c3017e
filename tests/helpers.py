"""shared helpers for the parity tests (test infrastructure)."""
import torch


def rel_l2(a, b):
    a = a.detach().double().flatten().cpu()
    b = b.detach().double().flatten().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def make_config(resnet_type=18, classes=101, **train):
    from core.config import config
    cfg = config.clone()
    cfg.MODEL.RESNET_TYPE = resnet_type
    cfg.MODEL.CLFDIM = classes
    cfg.MODEL.INIT_WEIGHTS = False
    for k, v in train.items():
        cfg.TRAIN[k] = v
    return cfg


def oracle_to_product_state(oracle_sd):
    """oracle keys backbone.* / clf_head.* -> product keys backbone.* / clf_haed.*"""
    out = {}
    for k, v in oracle_sd.items():
        out[k.replace("clf_head.", "clf_haed.")] = v.clone()
    return out


def build_pair(resnet_type=18, classes=101, seed=0, device="cuda"):
    """(oracle fp32 CPU model, product model on `device`) with identical weights."""
    from oracle.r3d_torch import R3DOracle, init_weights_
    from models.model import create_model
    oracle = init_weights_(R3DOracle(resnet_type, classes), seed)
    cfg = make_config(resnet_type, classes)
    prod = create_model(cfg, is_train=True)
    missing, unexpected = prod.load_state_dict(oracle_to_product_state(oracle.state_dict()), strict=True)
    assert not missing and not unexpected
    return oracle, prod.to(device), cfg


# ------------------------------------------------------------------ golden fixtures
def load_golden(name):
    """tests/golden/<name>.npz -> {key: torch tensor / python scalar}.  Keys ending in ':bf16' hold
    bf16 bit patterns (uint16) and come back as fp32 tensors under the key without the suffix."""
    import os
    import numpy as np
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".npz")
    out = {}
    with np.load(path, allow_pickle=False) as z:
        for k in z.files:
            a = z[k]
            if k.endswith(":bf16"):
                t = torch.from_numpy(a.view(np.int16).copy()).view(torch.bfloat16).float()
                out[k[:-5]] = t
            elif a.dtype.kind in "US":
                out[k] = str(a)
            elif a.ndim == 0:
                out[k] = a.item()
            else:
                out[k] = torch.from_numpy(a.copy())
    return out


def params_sha(params):
    import hashlib
    return hashlib.sha256(b"".join(p.detach().contiguous().numpy().tobytes() for p in params)).hexdigest()


def seeded_2d_weights(model3d, seed):
    """The 2D weights oracle/make_golden.py gave the reference's 2D BackBone (same module order,
    same RNG stream), placed at the centre temporal tap of `model3d` (a 3D backbone with the
    reference's attribute names; other taps zero).  Returns the list of 2D parameter tensors in
    parameter order, for the SHA check."""
    import torch.nn as nn
    g = torch.Generator().manual_seed(seed)
    flat2d = []
    for m in model3d.modules():
        if isinstance(m, nn.Conv3d):
            Co, Ci, kt, kh, kw = m.weight.shape
            fan_out = Co * kh * kw
            w2 = torch.randn(Co, Ci, kh, kw, generator=g) * (2.0 / fan_out) ** 0.5
            with torch.no_grad():
                m.weight.zero_()
                m.weight[:, :, kt // 2] = w2.to(m.weight.device)
            flat2d.append((m.weight, w2))
        elif isinstance(m, nn.BatchNorm3d):
            w = 1.0 + 0.1 * torch.randn(m.weight.shape, generator=g)
            b = 0.1 * torch.randn(m.bias.shape, generator=g)
            with torch.no_grad():
                m.weight.copy_(w)
                m.bias.copy_(b)
            flat2d.append((m.weight, w))
            flat2d.append((m.bias, b))
    # parameter order of the 2D model = named_parameters order of the 3D one
    by_id = {id(p): v for p, v in flat2d}
    return [by_id[id(p)] for p in model3d.parameters() if id(p) in by_id]
