"""CPU tests that pin the ORACLE (oracle/r3d_torch.py) before anything is compared with it.

1. Reference-anchored known answers: tests/golden/ref2d_r18.npz and
   ref2d_r50.npz hold outputs of the reference's own 2D BackBone
   (/root/reference/lib/models/backbone.py) run in the build container by
   oracle/make_golden.py.  The oracle's 3D network with each 2D kernel at the centre temporal
   tap, on the same input viewed as N x 3 x 1 x 224 x 224, must reproduce them: with T = 1 and
   temporal padding only the centre tap sees data, every temporal stride maps 1 -> 1, the
   3D max-pool pads with -inf, BN statistics are identical and the global average pool of a
   1 x 7 x 7 map equals AvgPool2d(7).  fp32 both sides: tolerance 2e-4 rel-L2 (conv3d and
   conv2d sum in different orders).
2. The reference's Loss (/root/reference/lib/core/loss.py) on three seeded cases, against the
   oracle's restatement AND the product's host-side Loss (pure torch, runs on CPU).
3. torch-version pins for the 3D arithmetic the reference does not contain ("parity
   unpinned" by the reference): ops3d.npz and r3d18_small.npz must be reproduced by the torch
   in this process, so the same fixtures can stand in for the oracle on the GPU box.
"""
import pytest
import torch
import torch.nn.functional as F

from helpers import load_golden, params_sha, rel_l2, seeded_2d_weights

TOL = 2e-4


def _t1_case(tag, depth):
    from oracle.r3d_torch import BackBone3D, resnet_spec
    g = load_golden(tag)
    block, layers = resnet_spec[depth]
    torch.manual_seed(0)
    model = BackBone3D(block, layers).train()
    flat2d = seeded_2d_weights(model, int(g["seed"]))
    if params_sha(flat2d) != g["w_sha"]:
        pytest.fail("torch's RNG stream differs from the one the fixture was generated with (a torch upgrade?): the "
                    "reference-anchored check cannot run -- regenerate the fixtures with oracle/make_golden.py; this is a "
                    "failure on purpose, a skip would silently switch the only reference-pinned checks off")
    gen = torch.Generator().manual_seed(int(g["seed"]) + 1)
    x = torch.randn(2, 3, 224, 224, generator=gen)
    assert params_sha([x]) == g["x_sha"]
    out = model(x.view(2, 3, 1, 224, 224))
    loss = out.pow(2).mean()
    loss.backward()
    assert rel_l2(out, g["out"]) <= TOL
    assert abs(float(loss.detach()) - g["loss"]) <= 1e-4 * abs(g["loss"])
    named = dict(model.named_parameters())
    bufs = dict(model.named_buffers())
    n_checked = 0
    for k, v in g.items():
        if k.startswith("gnorm/"):
            p = named[k[6:]]
            got = float(p.grad.double().norm())
            assert abs(got - v) <= 2e-3 * max(v, 1e-6), "%s: grad norm %.6e vs reference %.6e" % (k, got, v)
            if p.dim() == 5:  # off-centre temporal taps only ever see padding
                kt = p.shape[2]
                off = p.grad.clone()
                off[:, :, kt // 2] = 0
                assert float(off.abs().max()) == 0.0
            n_checked += 1
        elif k.startswith("grad/"):
            assert rel_l2(named[k[5:]].grad, v) <= 1e-2, k  # earliest layer of 53: fp32 order noise, ReLU-amplified
        elif k.startswith("buf/"):
            assert rel_l2(bufs[k[4:]], v) <= TOL, k
    assert n_checked == len(named)  # same parameter set as the reference (20 / 53 convs + BNs)
    return model


def test_t1_centre_tap_matches_reference_resnet18():
    model = _t1_case("ref2d_r18", 18)
    assert sum(isinstance(m, torch.nn.Conv3d) for m in model.modules()) == 20
    assert sum(isinstance(m, torch.nn.BatchNorm3d) for m in model.modules()) == 20


def test_t1_centre_tap_matches_reference_resnet50():
    model = _t1_case("ref2d_r50", 50)
    assert sum(isinstance(m, torch.nn.Conv3d) for m in model.modules()) == 53
    assert sum(isinstance(m, torch.nn.BatchNorm3d) for m in model.modules()) == 53


def test_oracle_parameter_counts():
    """R3D-18 = 33.26 M parameters (convs 33,194,304 + BN 9,600 + fc 51,813)."""
    from oracle.r3d_torch import R3DOracle
    m = R3DOracle(18, 101)
    convs = sum(p.numel() for p in m.parameters() if p.dim() == 5)
    bns = sum(p.numel() for n, p in m.named_parameters() if "bn" in n or "downsample.1" in n)
    fc = sum(p.numel() for n, p in m.named_parameters() if n.startswith("clf_head"))
    assert (convs, bns, fc) == (33194304, 9600, 51813)


def test_loss_matches_reference():
    import types
    from oracle.r3d_torch import loss_oracle
    from core.loss import Loss
    g = load_golden("loss_ref")
    for i in range(3):
        r_act, r_crt = (float(v) for v in g["r%d" % i])
        args = (g["score%d" % i], g["y%d" % i], g["log_prob%d" % i], g["adv%d" % i])
        want = g["loss%d" % i]
        assert abs(float(loss_oracle(*args, r_act=r_act, r_crt=r_crt)) - want) <= 1e-6 * abs(want)
        cfg = types.SimpleNamespace(TRAIN=types.SimpleNamespace(R_ACT=r_act, R_CRT=r_crt))
        assert abs(float(Loss(cfg)(*args)) - want) <= 1e-6 * abs(want)
    # all-zero actor/critic inputs (what the product's rollout supplies as None/None): plain CE
    args = (g["score2"], g["y2"])
    cfg = types.SimpleNamespace(TRAIN=types.SimpleNamespace(R_ACT=1.0, R_CRT=1.0))
    assert abs(float(Loss(cfg)(*args, None, None)) - g["loss2"]) <= 1e-6 * abs(g["loss2"])
    assert abs(float(F.cross_entropy(*args)) - g["loss2"]) <= 1e-6 * abs(g["loss2"])


def test_ops3d_fixture_is_reproduced_by_this_torch():
    g = load_golden("ops3d")
    for name in ("c3s1", "c3s2", "c1s2", "c1s1", "c3t1"):
        N, Ci, T, H, W, Co, k, s, p = (int(v) for v in g[name + "/geom"])
        x = g[name + "/x"].clone().requires_grad_()
        w = g[name + "/w"].clone().requires_grad_()
        y = F.conv3d(x, w, None, s, p)
        y.backward(g[name + "/dy"])
        assert rel_l2(y, g[name + "/y"]) <= 1e-6
        assert rel_l2(x.grad, g[name + "/dx"]) <= 1e-6
        assert rel_l2(w.grad, g[name + "/dw"]) <= 1e-6
    y = F.max_pool3d(g["pool/x"], 3, 2, 1)
    assert torch.equal(y, g["pool/y"])
    assert rel_l2(g["avg/x"].mean(dim=(2, 3, 4)), g["avg/y"]) <= 1e-6


def test_r3d18_small_fixture_is_reproduced_by_this_torch():
    from oracle.r3d_torch import R3DOracle, init_weights_, synthetic_batch
    g = load_golden("r3d18_small")
    model = init_weights_(R3DOracle(18, 101), 0).train()
    if params_sha(list(model.parameters())) != g["w_sha"]:
        pytest.fail("torch's RNG stream differs from the one the fixture was generated with (a torch upgrade?): the "
                    "reference-anchored check cannot run -- regenerate the fixtures with oracle/make_golden.py; this is a "
                    "failure on purpose, a skip would silently switch the only reference-pinned checks off")
    x, y = synthetic_batch(4, 8, 64, 64, 101, 4321)
    assert params_sha([x]) == g["x_sha"]
    logits = model(x)
    loss = F.cross_entropy(logits, y)
    loss.backward()
    assert rel_l2(logits, g["logits"]) <= 1e-5
    assert abs(float(loss) - g["loss"]) <= 1e-5 * g["loss"]
    for n, p in model.named_parameters():
        want = g["gnorm/" + n]
        assert abs(float(p.grad.double().norm()) - want) <= 1e-3 * max(want, 1e-9), n


def test_bf16_emulation_tracks_fp32_oracle():
    """forward_emulated (the oracle with the product's bf16 storage points) stays within bf16
    noise of the fp32 oracle on logits: the yardstick the end-to-end GPU test uses."""
    from oracle.r3d_torch import R3DOracle, init_weights_, synthetic_batch, forward_emulated
    model = init_weights_(R3DOracle(18, 11), 0).train()
    x, _ = synthetic_batch(4, 8, 64, 64, 11, 5)  # late BN layers need more than 2 values per channel
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    lo_e = forward_emulated(model, x)
    model.load_state_dict(sd)
    lo = model(x)
    assert rel_l2(lo_e, lo) <= 5e-2


def test_end_to_end_gradient_floor_of_bf16_operands():
    """Why no end-to-end gradient test carries north_star's 1e-2 figure.  The fp32 oracle is run twice on the same
    weights and clips; the second time ONLY the conv operands (inputs and weights) are rounded to bf16 -- storage,
    accumulation, BatchNorm and every gradient stay fp32.  That is the least any bf16-MFMA convolution does.  At
    random initialisation the whole-net gradient already moves by tens of percent (measured 0.34 at 2 x 16 x 112 x 112,
    0.2-0.4 here): about 0.3 % of the ReLU masks flip per layer, a flipped mask moves a whole gradient element, and
    at initialisation the per-position contributions to a weight gradient add incoherently.  The logits move by 1e-2.
    So the 1e-2 bar on gradients is applied per kernel and per layer instance on the product's own inputs
    (tests/test_ops_gpu.py, tests/test_teacher_forced_gpu.py), and end to end the product is held to the CPU
    bf16-storage emulation (tests/test_net_gpu.py) and to the fp32 training trajectory (tests/test_train_gpu.py)."""
    import torch.nn as nn
    import torch.nn.functional as F
    from oracle.r3d_torch import R3DOracle, init_weights_, synthetic_batch

    class _RoundFwd(torch.autograd.Function):
        @staticmethod
        def forward(ctx, t):
            return t.to(torch.bfloat16).float()

        @staticmethod
        def backward(ctx, g):
            return g

    def grads(round_operands):
        m = init_weights_(R3DOracle(18, 101), 0).train()
        if round_operands:
            for mod in m.modules():
                if isinstance(mod, nn.Conv3d):
                    mod.forward = (lambda c: lambda inp: F.conv3d(_RoundFwd.apply(inp), _RoundFwd.apply(c.weight), None,
                                                                  c.stride, c.padding))(mod)
        x, y = synthetic_batch(2, 8, 64, 64, 101, 2468)
        lo = m(x)
        F.cross_entropy(lo, y).backward()
        return lo.detach(), torch.cat([p.grad.flatten() for p in m.parameters()]).double()

    l0, g0 = grads(False)
    l1, g1 = grads(True)
    e_logit = float((l1 - l0).norm() / l0.norm())
    e_grad = float((g1 - g0).norm() / g0.norm())
    print("bf16 conv operands only: logits rel-L2 %.3e, whole-net gradient rel-L2 %.3e" % (e_logit, e_grad))
    assert e_logit <= 5e-2
    assert e_grad >= 3e-2, "the gradient floor of bf16 operands was expected to exceed north_star's 1e-2 figure"
