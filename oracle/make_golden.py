#!/usr/bin/env python3
"""ORACLE tooling (test infrastructure, not product code): generates tests/golden/*.npz.

Run in the BUILD CONTAINER only (it imports the reference from /root/reference, which does
not exist on the GPU box):

    python oracle/make_golden.py

What it writes, and what pins what:

  ref2d_r18.npz / ref2d_r50.npz   -- outputs of the REFERENCE ITSELF: its 2D BackBone
      (/root/reference/lib/models/backbone.py:106-157, BasicBlock :33-62 / Bottleneck :65-103)
      imported and run here in train mode on a seeded 2x3x224x224 input: features, BN running
      statistics, per-parameter gradient norms of loss = mean(out^2).  The oracle's 3D network
      (oracle/r3d_torch.py) with every 2D kernel placed at the centre temporal tap must
      reproduce them on the input viewed as 2x3x1x224x224 (tests/test_oracle_cpu.py): this
      pins the oracle's topology, strides, BN placement and momentum to the reference.
  loss_ref.npz                    -- outputs of the reference's Loss
      (/root/reference/lib/core/loss.py:7-30) on three seeded cases.
  ops3d.npz                       -- stock torch.nn CPU fp32 3D ops (the arithmetic the
      reference does not contain; "parity unpinned" by the reference, pinned here to
      torch 2.10.0 CPU): conv3d fwd/dgrad/wgrad for every (kernel, stride, pad) of the layer
      table, train-mode BatchNorm3d fwd/bwd with running statistics, MaxPool3d fwd/bwd,
      global average pool, Linear.  Inputs are stored with the outputs.
  r3d18_small.npz                 -- whole-net oracle R3D-18 on a seeded 4x3x8x64x64 batch (2 clips of 32x32 leave late BN layers with 2 values per channel: ill-conditioned):
      logits, loss, per-parameter gradient norms, BN running statistics after one step.
  r3d18_full.npz                  -- the same at the bench geometry, 2x3x16x112x112, 101 classes: logits, loss, gradient norms.
  r3d50_full.npz / r3d50_long.npz -- R3D-50, 400 classes, at BASELINE config 3's geometry (2x3x16x112x112) and config 4's
      (1x3x64x224x224): logits, loss, per-parameter gradient norms of the fp32 oracle, and the same from the oracle's
      bf16-storage emulation (keys emu/...).

Inputs and weights that are too large to commit are regenerated from seeds; a SHA-256 of
their bytes is stored so a test can tell "RNG stream changed" from "arithmetic changed".
"""
import hashlib
import importlib.util
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.r3d_torch import R3DOracle, init_weights_, synthetic_batch  # noqa: E402

REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")


def sha(t):
    return hashlib.sha256(t.detach().contiguous().numpy().tobytes()).hexdigest()


def load_ref(relpath, name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, relpath))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def seeded_2d_weights_(model, seed):
    """same rule as oracle.r3d_torch.init_weights_, on the reference's 2D modules."""
    g = torch.Generator().manual_seed(seed)
    for m in model.modules():
        if isinstance(m, nn.Conv2d):
            fan_out = m.out_channels * m.kernel_size[0] * m.kernel_size[1]
            with torch.no_grad():
                m.weight.copy_(torch.randn(m.weight.shape, generator=g) * (2.0 / fan_out) ** 0.5)
        elif isinstance(m, nn.BatchNorm2d):
            with torch.no_grad():
                m.weight.copy_(1.0 + 0.1 * torch.randn(m.weight.shape, generator=g))
                m.bias.copy_(0.1 * torch.randn(m.bias.shape, generator=g))
    return model


def ref2d(block_name, layers, tag, seed):
    rb = load_ref("lib/models/backbone.py", "ref_backbone")
    torch.manual_seed(seed)
    model = rb.BackBone(getattr(rb, block_name), layers)
    seeded_2d_weights_(model, seed)
    model.train()
    g = torch.Generator().manual_seed(seed + 1)
    x = torch.randn(2, 3, 224, 224, generator=g)
    wsha = hashlib.sha256(b"".join(p.detach().numpy().tobytes() for p in model.parameters())).hexdigest()
    out = model(x)
    loss = out.pow(2).mean()
    loss.backward()
    rec = {"seed": np.int64(seed), "x_sha": np.array(sha(x)), "w_sha": np.array(wsha),
           "out": out.detach().numpy(), "loss": np.float64(float(loss))}
    for n, p in model.named_parameters():
        rec["gnorm/" + n] = np.float64(float(p.grad.double().norm()))
    for n in ("bn1.weight", "bn1.bias", "layer4.1.bn2.weight" if block_name == "BasicBlock" else "layer4.2.bn3.weight"):
        rec["grad/" + n] = dict(model.named_parameters())[n].grad.detach().numpy()
    for n, b in model.named_buffers():
        if n in ("bn1.running_mean", "bn1.running_var", "layer2.0.downsample.1.running_var"):
            rec["buf/" + n] = b.detach().numpy()
    np.savez_compressed(os.path.join(OUT, tag + ".npz"), **rec)
    print(tag, "out", tuple(out.shape), "loss %.6f" % float(loss), "params", sum(p.numel() for p in model.parameters()))


def loss_ref():
    rl = load_ref("lib/core/loss.py", "ref_loss")
    rec = {}
    for i, (r_act, r_crt, zero) in enumerate([(1.0, 1.0, False), (0.5, 2.0, False), (1.0, 1.0, True)]):
        cfg = types.SimpleNamespace(TRAIN=types.SimpleNamespace(R_ACT=r_act, R_CRT=r_crt))
        crit = rl.Loss(cfg)
        g = torch.Generator().manual_seed(100 + i)
        score = torch.randn(6, 11, generator=g)
        y = torch.randint(0, 11, (6,), generator=g)
        lp = torch.zeros(6, 4) if zero else -torch.rand(6, 4, generator=g)
        adv = torch.zeros(6, 4) if zero else torch.randn(6, 4, generator=g)
        out = crit(score, y, lp, adv)
        rec.update({"score%d" % i: score.numpy(), "y%d" % i: y.numpy(), "log_prob%d" % i: lp.numpy(),
                    "adv%d" % i: adv.numpy(), "r%d" % i: np.array([r_act, r_crt]), "loss%d" % i: np.float64(float(out))})
    np.savez_compressed(os.path.join(OUT, "loss_ref.npz"), **rec)
    print("loss_ref", [float(rec["loss%d" % i]) for i in range(3)])


def rbf(t):
    return t.to(torch.bfloat16).to(torch.float32)


def bits(t):
    """bf16-representable fp32 tensor -> its bf16 bit pattern (uint16): exact and half the bytes.
    tests/helpers.py:golden_array() undoes this for keys ending in ':bf16'."""
    t = t.detach()
    assert torch.equal(rbf(t), t)
    return t.to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)


def ops3d():
    """inputs are bf16-representable so the GPU path (bf16 storage) sees exactly these numbers."""
    rec = {}
    g = torch.Generator().manual_seed(7)
    cases = [  # name, N, Ci, T, H, W, Co, k, s, p
        ("c3s1", 2, 64, 4, 6, 5, 64, 3, 1, 1),
        ("c3s2", 2, 64, 5, 9, 7, 64, 3, 2, 1),
        ("c1s2", 2, 64, 4, 8, 8, 128, 1, 2, 0),
        ("c1s1", 2, 128, 2, 5, 5, 64, 1, 1, 0),
        ("c3t1", 2, 64, 1, 6, 6, 64, 3, 1, 1),
    ]
    for name, N, Ci, T, H, W, Co, k, s, p in cases:
        x = rbf(torch.randn(N, Ci, T, H, W, generator=g)).requires_grad_()
        w = rbf(torch.randn(Co, Ci, k, k, k, generator=g) * (2.0 / (Co * k ** 3)) ** 0.5).requires_grad_()
        y = F.conv3d(x, w, None, s, p)
        dy = rbf(torch.randn(y.shape, generator=g))
        y.backward(dy)
        rec.update({name + "/geom": np.array([N, Ci, T, H, W, Co, k, s, p]), name + "/x:bf16": bits(x),
                    name + "/w:bf16": bits(w), name + "/y": y.detach().numpy(), name + "/dy:bf16": bits(dy),
                    name + "/dx": x.grad.numpy(), name + "/dw": w.grad.numpy()})
    # stem 7x7x7 stride (1,2,2) pad 3 on 3 channels
    x = rbf(torch.randn(2, 3, 4, 16, 16, generator=g))
    w = rbf(torch.randn(64, 3, 7, 7, 7, generator=g) * 0.05).requires_grad_()
    y = F.conv3d(x, w, None, (1, 2, 2), 3)
    dy = rbf(torch.randn(y.shape, generator=g))
    y.backward(dy)
    rec.update({"stem/x:bf16": bits(x), "stem/w:bf16": bits(w), "stem/y": y.detach().numpy(), "stem/dy:bf16": bits(dy),
                "stem/dw": w.grad.numpy()})
    # train-mode BN (+ReLU) fwd/bwd with running statistics
    bn = nn.BatchNorm3d(64, momentum=0.1)
    with torch.no_grad():
        bn.weight.copy_(1 + 0.2 * torch.randn(64, generator=g))
        bn.bias.copy_(0.2 * torch.randn(64, generator=g))
    xb = rbf(torch.randn(2, 64, 3, 5, 5, generator=g) * 1.5 + 0.3).requires_grad_()
    z = torch.relu(bn(xb))
    dz = rbf(torch.randn(z.shape, generator=g))
    z.backward(dz)
    rec.update({"bn/x:bf16": bits(xb), "bn/gamma": bn.weight.detach().numpy(), "bn/beta": bn.bias.detach().numpy(),
                "bn/z": z.detach().numpy(), "bn/dz:bf16": bits(dz), "bn/dx": xb.grad.numpy(),
                "bn/dgamma": bn.weight.grad.numpy(), "bn/dbeta": bn.bias.grad.numpy(),
                "bn/running_mean": bn.running_mean.numpy(), "bn/running_var": bn.running_var.numpy()})
    # max pool k3 s2 p1 and global average pool
    xp = rbf(torch.randn(2, 64, 5, 7, 9, generator=g)).requires_grad_()
    yp = F.max_pool3d(xp, 3, 2, 1)
    dyp = rbf(torch.randn(yp.shape, generator=g))
    yp.backward(dyp)
    rec.update({"pool/x:bf16": bits(xp), "pool/y": yp.detach().numpy(), "pool/dy:bf16": bits(dyp),
                "pool/dx": xp.grad.numpy()})
    xa = rbf(torch.randn(3, 128, 2, 3, 3, generator=g))
    rec.update({"avg/x:bf16": bits(xa), "avg/y": xa.mean(dim=(2, 3, 4)).numpy()})
    # classifier head
    feat = torch.randn(5, 128, generator=g)
    Wl = torch.randn(11, 128, generator=g) * 0.1
    bl = torch.randn(11, generator=g) * 0.1
    rec.update({"fc/feat": feat.numpy(), "fc/W": Wl.numpy(), "fc/b": bl.numpy(), "fc/out": F.linear(feat, Wl, bl).numpy()})
    np.savez_compressed(os.path.join(OUT, "ops3d.npz"), **rec)
    print("ops3d", len(rec), "arrays")


def r3d18_small():
    model = init_weights_(R3DOracle(18, 101), 0).train()
    x, y = synthetic_batch(4, 8, 64, 64, 101, 4321)
    wsha = hashlib.sha256(b"".join(p.detach().numpy().tobytes() for p in model.parameters())).hexdigest()
    logits = model(x)
    loss = F.cross_entropy(logits, y)
    loss.backward()
    rec = {"x_sha": np.array(sha(x)), "w_sha": np.array(wsha), "y": y.numpy(), "logits": logits.detach().numpy(),
           "loss": np.float64(float(loss))}
    for n, p in model.named_parameters():
        rec["gnorm/" + n] = np.float64(float(p.grad.double().norm()))
    rec["buf/backbone.bn1.running_mean"] = model.backbone.bn1.running_mean.numpy()
    rec["buf/backbone.bn1.running_var"] = model.backbone.bn1.running_var.numpy()
    np.savez_compressed(os.path.join(OUT, "r3d18_small.npz"), **rec)
    print("r3d18_small loss %.6f" % float(loss))


def r3d18_full():
    """the bench geometry (16x112x112 clips, 101 classes) at batch 2: every kernel of the product path runs in the
    regime it is tuned for (image-resident stem kernels at 56-wide rows, padded-halo convs, wgrad9, T = 1 in layer4)"""
    model = init_weights_(R3DOracle(18, 101), 0).train()
    x, y = synthetic_batch(2, 16, 112, 112, 101, 2468)
    wsha = hashlib.sha256(b"".join(p.detach().numpy().tobytes() for p in model.parameters())).hexdigest()
    logits = model(x)
    loss = F.cross_entropy(logits, y)
    loss.backward()
    rec = {"x_sha": np.array(sha(x)), "w_sha": np.array(wsha), "y": y.numpy(), "logits": logits.detach().numpy(),
           "loss": np.float64(float(loss))}
    for n, p in model.named_parameters():
        rec["gnorm/" + n] = np.float64(float(p.grad.double().norm()))
    np.savez_compressed(os.path.join(OUT, "r3d18_full.npz"), **rec)
    print("r3d18_full loss %.6f" % float(loss))


def _whole_net(depth, classes, n, t, hw, seed, tag, emulate=True):
    """fp32 oracle logits / loss / per-parameter gradient norms (+ two whole gradient tensors) and the same from the
    oracle's bf16-storage emulation (the yardstick for what bf16 storage + bf16 MFMA operands cost at this size)."""
    from oracle.r3d_torch import forward_emulated
    rec = {}
    for kind in (("fp32", "emu") if emulate else ("fp32",)):
        model = init_weights_(R3DOracle(depth, classes), 0).train()
        x, y = synthetic_batch(n, t, hw, hw, classes, seed)
        if kind == "fp32":
            rec["x_sha"] = np.array(sha(x))
            rec["w_sha"] = np.array(hashlib.sha256(b"".join(p.detach().numpy().tobytes() for p in model.parameters())).hexdigest())
            rec["y"] = y.numpy()
        logits = model(x) if kind == "fp32" else forward_emulated(model, x)
        loss = F.cross_entropy(logits, y)
        loss.backward()
        pre = "" if kind == "fp32" else "emu/"
        rec[pre + "logits"] = logits.detach().numpy()
        rec[pre + "loss"] = np.float64(float(loss))
        for nme, p in model.named_parameters():
            rec[pre + "gnorm/" + nme] = np.float64(float(p.grad.double().norm()))
        for nme in ("clf_head.bias", "backbone.bn1.weight"):
            rec[pre + "grad/" + nme] = dict(model.named_parameters())[nme].grad.detach().numpy()
        print(tag, kind, "loss %.6f" % float(loss))
        del model, logits, loss
    np.savez_compressed(os.path.join(OUT, tag + ".npz"), **rec)


def r3d50_full():
    """BASELINE config 3's per-GPU geometry at batch 2: R3D-50, 400 classes, 2x3x16x112x112 (Bottleneck kernels in the
    regimes of the bench: persistent 1x1x1 implicit GEMMs, wide-channel reductions, layer4 at T = 1)."""
    _whole_net(50, 400, 2, 16, 112, 1357, "r3d50_full")


def r3d50_long():
    """BASELINE config 4's geometry at batch 1: R3D-50, 400 classes, 1x3x64x224x224 (the largest the 8-core build
    container generates in about a minute)."""
    _whole_net(50, 400, 1, 64, 224, 9753, "r3d50_long")


def main():
    if not os.path.isdir(REF):
        raise SystemExit("make_golden.py runs in the build container only (needs %s)" % REF)
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    ref2d("BasicBlock", [2, 2, 2, 2], "ref2d_r18", 11)
    ref2d("Bottleneck", [3, 4, 6, 3], "ref2d_r50", 12)
    loss_ref()
    ops3d()
    r3d18_small()
    r3d18_full()
    r3d50_full()
    r3d50_long()


if __name__ == "__main__":
    main()
