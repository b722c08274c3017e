"""The deterministic-reduction mode (SURVEY.md section 5, section 4 "bitwise for fp32 fixed order"): vc_set_deterministic /
VC_DETERMINISTIC=1 turns every BatchNorm sum -- the one place where the default path uses float atomics -- into exact
64-bit fixed-point integer accumulation.  Checked here: (1) per kernel, the mode reproduces the default mode's
statistics to float rounding and itself bit for bit; (2) whole training steps of both models are bit-identical from run
to run (with the allocator poisoned in between); (3) non-finite values are not swallowed."""
import pytest
import torch
import torch.nn.functional as F

from helpers import make_config

pytestmark = pytest.mark.gpu


def _bf(t):
    return t.to(torch.bfloat16).float()


def _fwd_case(kind):
    """returns run(stats, fin) -> y, Co, count for one producer of forward BatchNorm sums"""
    from vc3d import ops
    g = torch.Generator().manual_seed(21)
    if kind == "stem":
        N, T, H, W, Co = 3, 4, 32, 32, 64
        x = _bf(torch.randn(N, 3, T, H, W, generator=g) + 0.3)
        w = _bf(torch.randn(Co, 3, 7, 7, 7, generator=g) * 0.05)
        xs = ops.stem_pack_input(x.cuda())
        ws = ops.stem_pack_weight(w.cuda().contiguous(memory_format=torch.channels_last_3d))
        return (lambda st, fin: ops.stem_fwd(xs, ws, N, T, H, W, Co, stats=st, fin=fin)), Co, N * T * (H // 2) * (W // 2)
    N, Ci, T, H, W, Co, k, s, p = {"halo": (4, 64, 4, 14, 14, 64, 3, 1, 1), "halo_2wg": (8, 64, 8, 28, 28, 64, 3, 1, 1),
                                   "pointwise": (4, 64, 2, 14, 14, 256, 1, 1, 0), "strided": (4, 64, 4, 14, 14, 128, 3, 2, 1),
                                   "splitk": (8, 256, 2, 7, 7, 256, 3, 1, 1)}[kind]
    x = _bf(torch.randn(N, Ci, T, H, W, generator=g) + 0.2)
    w = _bf(torch.randn(Co, Ci, k, k, k, generator=g) * (2.0 / (Ci * k ** 3)) ** 0.5)
    d = ops.conv_desc(N, Ci, T, H, W, Co, k, s, p)
    xd = x.permute(0, 2, 3, 4, 1).contiguous().to(torch.bfloat16).cuda()
    wf, _ = ops.pack_weight(w.cuda().contiguous(memory_format=torch.channels_last_3d))
    To, Ho, Wo = d.out_dims()
    return (lambda st, fin: ops.conv3d_fwd(d, xd, wf, stats=st, fin=fin)), Co, N * To * Ho * Wo


def _run_fwd(run, Co, count, fused):
    from vc3d import ops
    gamma, beta = torch.ones(Co, device="cuda") * 1.5, torch.full((Co,), 0.1, device="cuda")
    rm, rv = torch.zeros(Co, device="cuda"), torch.ones(Co, device="cuda")
    st = ops.new_stats(Co, "cuda")
    if fused:
        ticket = torch.zeros(64, dtype=torch.int32, device="cuda")
        fin, prm = ops.bn_fin(ticket, count, gamma, beta, rm, rv, 0.1, 1e-5)
        y = run(st, fin)
    else:
        y = run(st, None)
        prm = ops.bn_finalize(st, count, gamma, beta, rm, rv, 0.1, 1e-5)
    torch.cuda.synchronize()
    return y, prm.clone(), rm, rv


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("kind", ["halo", "halo_2wg", "pointwise", "strided", "splitk", "stem"])
def test_forward_sums_deterministic_mode_matches_float_mode_and_repeats(kind, fused):
    from vc3d import ops
    run, Co, count = _fwd_case(kind)
    assert not ops.set_deterministic(False)
    y0, p0, rm0, rv0 = _run_fwd(run, Co, count, fused)
    ops.set_deterministic(True)
    try:
        y1, p1, rm1, rv1 = _run_fwd(run, Co, count, fused)
        y2, p2, rm2, rv2 = _run_fwd(run, Co, count, fused)
    finally:
        ops.set_deterministic(False)
    assert torch.equal(y0, y1) and torch.equal(y1, y2)
    assert torch.equal(p1, p2) and torch.equal(rm1, rm2) and torch.equal(rv1, rv2), "deterministic mode must repeat bit for bit"
    # against the float-atomic sums: the same numbers up to the rounding of the float additions
    assert torch.allclose(p0, p1, rtol=3e-5, atol=3e-6), float((p0 - p1).abs().max())
    assert torch.allclose(rv0, rv1, rtol=3e-5, atol=3e-6)


@pytest.mark.parametrize("mask", [0, 2, 3])
def test_backward_sums_deterministic_mode(mask, det_mode):
    from vc3d import ops
    C, rows = 128, 4 * 4 * 14 * 14
    g = torch.Generator().manual_seed(mask)
    y = (torch.randn(rows, C, generator=g) * 2 + 0.5).to(torch.bfloat16).cuda()
    dz = (torch.randn(rows, C, generator=g) * 1e-3).to(torch.bfloat16).cuda()
    prm = ops.bn_finalize(ops.bn_stats(y, C), rows, torch.ones(C, device="cuda"), torch.zeros(C, device="cuda"), None, None,
                          0.1, 1e-5)
    aux = None
    if mask == 3:
        aux = torch.randint(0, 256, (rows * C // 8,), generator=g, dtype=torch.uint8).cuda()
    outs = []
    for fused in (True, False, True):
        dg, db = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
        ticket = torch.zeros(64, dtype=torch.int32, device="cuda") if fused else None
        dy, _ = ops.bn_bwd(y, dz, prm, C, dg, db, mask=mask, aux=aux, ticket=ticket)
        torch.cuda.synchronize()
        outs.append((dy.clone(), dg.clone(), db.clone()))
    for a, b in zip(outs[0], outs[2]):
        assert torch.equal(a, b)
    for a, b in zip(outs[0], outs[1]):  # fused and stand-alone finalize read the same integers
        assert torch.equal(a, b)
    # reference: fp64 on the host
    yf, gf = y.double().cpu(), dz.double().cpu()
    if mask == 2:
        gf = gf * ((yf * prm[2].double().cpu() + prm[3].double().cpu()) > 0)
    elif mask == 3:
        bits = aux.cpu().view(rows, C // 8)
        m = torch.stack([(bits >> k) & 1 for k in range(8)], dim=2).reshape(rows, C).double()
        gf = gf * m
    xhat = (yf - prm[0].double().cpu()) * prm[1].double().cpu()
    assert torch.allclose(outs[0][2].double().cpu(), gf.sum(0), rtol=1e-5, atol=1e-7)
    assert torch.allclose(outs[0][1].double().cpu(), (gf * xhat).sum(0), rtol=1e-4, atol=1e-6)


def test_deterministic_sums_do_not_swallow_non_finite_values(det_mode):
    from vc3d import ops
    C, rows = 64, 1024
    y = torch.randn(rows, C).to(torch.bfloat16).cuda()
    y[17, 5] = float("inf")
    y[900, 9] = float("nan")
    prm = ops.bn_finalize(ops.bn_stats(y, C), rows, torch.ones(C, device="cuda"), torch.zeros(C, device="cuda"), None, None,
                          0.1, 1e-5)
    torch.cuda.synchronize()
    assert bool(torch.isnan(prm[0, 5])) and bool(torch.isnan(prm[0, 9]))
    ok = torch.ones(C, dtype=torch.bool)
    ok[5] = ok[9] = False
    assert bool(torch.isfinite(prm[:, ok.cuda()]).all())


def _poison():
    bufs = [torch.full((32 << 20,), float("nan"), dtype=torch.bfloat16, device="cuda") for _ in range(8)]
    small = [torch.full((n,), float("nan"), dtype=torch.float32, device="cuda") for n in (256, 4096, 65536, 1 << 20) for _ in range(8)]
    del bufs, small


def _two_steps(depth, classes, shape, opt_name):
    from core.loss import Loss
    from models.model import create_model
    from utils.utils import create_optimizer
    cfg = make_config(depth, classes, OPTIMIZER=opt_name, LR=0.01)
    torch.manual_seed(3)
    model = create_model(cfg, is_train=True).cuda().train()
    crit, opt = Loss(cfg).cuda(), create_optimizer(cfg, model)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(*shape, generator=g).to(torch.bfloat16).cuda()
    y = torch.randint(0, classes, (shape[0],), generator=g).cuda()
    losses = []
    for _ in range(2):
        _, score, _ = model(x)
        loss = crit(score, y, None, None)
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(loss.detach().clone())
    torch.cuda.synchronize()
    eng = model.backbone.engine
    bufs = torch.cat([b.detach().float().flatten() for b in model.buffers()])
    return torch.stack(losses), eng.flat_grad.clone(), eng.flat_param.clone(), bufs, score.detach().clone()


@pytest.mark.parametrize("depth,classes,shape,opt_name", [(18, 11, (6, 3, 8, 64, 64), "sgd"), (50, 17, (4, 3, 8, 64, 64), "adam"),
                                                          (18, 101, (16, 3, 16, 112, 112), "sgd")])
def test_training_steps_are_bit_identical_in_deterministic_mode(depth, classes, shape, opt_name, det_mode):
    """two optimizer steps (forward with batch statistics, backward on two streams, fused optimizer), three times from the
    same seed with the caching allocator poisoned with NaNs in between: losses, logits, every gradient, every parameter
    and every BatchNorm buffer are bit-identical.  (Default mode, same script: logits differ by ~1e-2 rel-L2 between runs
    at these sizes -- scripts/determinism.py.)"""
    ref = _two_steps(depth, classes, shape, opt_name)
    assert bool(torch.isfinite(ref[1]).all())
    for _ in range(2):
        _poison()
        got = _two_steps(depth, classes, shape, opt_name)
        for a, b, name in zip(ref, got, ("losses", "gradients", "parameters", "buffers", "logits")):
            assert torch.equal(a, b), "%s differ between two runs in deterministic mode: max |d| %g" % (
                name, float((a.float() - b.float()).abs().max()))


def _structured_batch(n, t, hw, classes, seed, amp=0.1):
    """clips ~ N(0,1) plus a class-dependent low-frequency pattern, so that training has something to learn (SURVEY 8(d)'s
    uniform labels on white noise can only be memorised, which drives the loss -- and every gradient -- to zero)."""
    g = torch.Generator().manual_seed(seed)
    y = torch.randint(0, classes, (n,), generator=g)
    pg = torch.Generator().manual_seed(99)  # the class patterns: the same for every batch
    pat = torch.randn(classes, 3, 2, 4, 4, generator=pg)
    pat = F.interpolate(pat, size=(t, hw, hw), mode="trilinear", align_corners=False)
    x = torch.randn(n, 3, t, hw, hw, generator=g) + amp * pat[y]
    return x, y


@pytest.mark.parametrize("depth", [18, 50])
def test_end_to_end_gradients_at_a_conditioned_operating_point(depth, det_mode):
    """north_star's rel-L2 <= 1e-2 on gradients is met per kernel and teacher-forced; end to end, at
    RANDOM initialisation on 2-4 clips, the network is chaotic (tests/test_oracle_cpu.py) and only a loose bar is
    possible.  Here the comparison is made where training operates: 16 clips per batch, weights after 100 SGD steps
    (trained by the product path itself on a learnable synthetic task -- class-dependent low-frequency patterns under
    unit noise -- in deterministic mode, so that the operating point is the same in every run), then ONE batch the net
    has not seen (loss ~0.6-2): fp32 oracle on the CPU against the product, same weights, same clips.
    What was measured (MI355X, printed below): logits 4e-3 / 1.2e-2, loss within 0.5 %; whole-gradient rel-L2 R3D-50 0.134
    with the oracle's own CPU bf16-STORAGE emulation at 0.141 (cosine 0.991 / 0.990), R3D-18 0.1791 against 0.1795
    (cosine 0.9839 / 0.9838).  So training does not take the end-to-end floor far below 0.1-0.2: it is the rounding of the
    stored activations (the emulation has it with fp32 arithmetic everywhere), and the product sits ON that floor.  That
    makes the sharp bar a relative one: err(product) <= 1.15 x err(emulation) + 0.01 (rounds 1-2 allowed 1.25 x + 0.02
    at random initialisation), cosine no more than 0.005 below the emulation's, logits <= 3e-2 and <= 1.25 x emulation
    + 5e-3, loss within 1 %."""
    from helpers import build_pair, rel_l2
    from oracle.r3d_torch import forward_emulated
    from utils.utils import create_optimizer
    classes = 16
    oracle, prod, cfg = build_pair(depth, classes)
    cfg.TRAIN.OPTIMIZER, cfg.TRAIN.LR, cfg.TRAIN.MOMENTUM, cfg.TRAIN.WD = "sgd", 0.005, 0.9, 1e-4
    opt = create_optimizer(cfg, prod)
    prod.train()
    batches = [_structured_batch(16, 8, 64, classes, 100 + i) for i in range(10)]
    dev_batches = [(x.to(torch.bfloat16).cuda(), y.cuda()) for x, y in batches]
    first = last = None
    for step in range(100):
        x, y = dev_batches[step % len(dev_batches)]
        _, score, _ = prod(x)
        loss = F.cross_entropy(score, y)
        opt.zero_grad()
        loss.backward()
        opt.step()
        if step < 10:
            first = float(loss.detach()) if first is None else first
        last = loss
    last = float(last)
    assert last < 0.7 * first, "training was expected to make progress (%.3f -> %.3f)" % (first, last)
    # the operating point, handed to the oracle
    sd = {k.replace("clf_haed.", "clf_head."): v.detach().float().cpu().clone() for k, v in prod.state_dict().items()}
    oracle.load_state_dict(sd)
    x, y = _structured_batch(16, 8, 64, classes, 777)
    x = x.to(torch.bfloat16).float()  # both paths see the same (bf16-representable) clips
    oracle.train()
    sd0 = {k: v.clone() for k, v in oracle.state_dict().items()}
    lo_emu = forward_emulated(oracle, x)
    F.cross_entropy(lo_emu, y).backward()
    g_emu = {n: p.grad.detach().clone() for n, p in oracle.named_parameters()}
    oracle.load_state_dict(sd0)
    oracle.zero_grad()
    lo_ref = oracle(x)
    loss_ref = F.cross_entropy(lo_ref, y)
    loss_ref.backward()
    g_ref = {n: p.grad.detach().clone() for n, p in oracle.named_parameters()}
    opt.zero_grad()
    _, lo, _ = prod(x.cuda())
    loss = F.cross_entropy(lo, y.cuda())
    loss.backward()
    torch.cuda.synchronize()
    g_prod = {n.replace("clf_haed.", "clf_head."): p.grad.detach().float().cpu() for n, p in prod.named_parameters()}
    keys = sorted(g_ref)
    cat = lambda d: torch.cat([d[k].flatten().double() for k in keys])
    a, b, c = cat(g_prod), cat(g_ref), cat(g_emu)
    e_all, e_emu = float((a - b).norm() / b.norm()), float((c - b).norm() / b.norm())
    cos = float(torch.dot(a, b) / (a.norm() * b.norm()))
    cos_emu = float(torch.dot(c, b) / (c.norm() * b.norm()))
    per = sorted(((rel_l2(g_prod[k], g_ref[k]), k) for k in keys if g_ref[k].dim() > 1), reverse=True)
    per_bn = sorted(((rel_l2(g_prod[k], g_ref[k]), k) for k in keys if g_ref[k].dim() == 1), reverse=True)
    e_logit = rel_l2(lo, lo_ref)
    print("R3D-%d after 100 steps (loss %.3f -> %.3f): logits rel-L2 %.3e, loss %.5f vs %.5f, whole gradient rel-L2 %.3e "
          "(emulation %.3e), cos %.5f (emulation %.5f)" % (depth, first, last, e_logit, float(loss), float(loss_ref), e_all, e_emu,
                                                          cos, cos_emu))
    print("  worst conv / fc weight gradients:", ["%s=%.3e" % (k, v) for v, k in per[:4]], "median %.3e" % per[len(per) // 2][0])
    print("  worst BatchNorm / bias gradients:", ["%s=%.3e" % (k, v) for v, k in per_bn[:4]], "median %.3e" % per_bn[len(per_bn) // 2][0])
    e_logit_emu = rel_l2(lo_emu, lo_ref)
    print("  logits: emulation rel-L2 %.3e" % e_logit_emu)
    assert e_logit <= 3e-2 and e_logit <= 1.25 * e_logit_emu + 5e-3
    assert abs(float(loss) - float(loss_ref)) <= 1e-2 * float(loss_ref)
    assert e_all <= 1.15 * e_emu + 1e-2
    assert cos >= cos_emu - 5e-3
