"""GPU parity tests, op level: every HIP kernel behind the C ABI against stock torch fp32 on
the CPU (the oracle's arithmetic), on the same seeded inputs.

Tolerances (stated per north_star: fp32 reference vs bf16 MFMA with fp32 accumulation):
inputs are rounded to bf16 before both paths, so the only differences are summation order
and the final bf16 rounding of the output (2^-9 relative per element).
  bf16 outputs : rel-L2 <= 4e-3, max |err| <= 2e-2 * max|ref|
  fp32 outputs : rel-L2 <= 1e-3
"""
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL_BF16 = 4e-3
TOL_F32 = 1e-3


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def rel_l2(a, b):
    a = a.double().flatten()
    b = b.double().flatten()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def bf16_round(t):
    return t.to(torch.bfloat16).to(torch.float32)


def to_nthwc_dev(x):
    return x.permute(0, 2, 3, 4, 1).contiguous().to(torch.bfloat16).cuda()


def from_nthwc(y):
    return y.float().cpu().permute(0, 4, 1, 2, 3).contiguous()


def cl3d_param(w):
    """fp32 weight on the GPU stored channels_last_3d, as the product's parameters are."""
    return w.cuda().contiguous(memory_format=torch.channels_last_3d)


CONV_CASES = [
    # N, Ci, T, H, W, Co, k, s, p
    (2, 64, 4, 6, 5, 64, 3, 1, 1),
    (2, 64, 5, 9, 7, 128, 3, 2, 1),
    (1, 128, 4, 8, 8, 128, 3, 1, 1),
    (3, 128, 3, 7, 7, 256, 3, 2, 1),
    (2, 64, 4, 8, 8, 128, 1, 2, 0),
    (2, 256, 2, 5, 5, 64, 1, 1, 0),
    (2, 64, 8, 14, 14, 64, 3, 1, 1),   # several 256-row tiles
    (2, 64, 1, 6, 6, 64, 3, 1, 1),     # T = 1 (centre-tap regime)
    (2, 64, 1, 7, 7, 128, 3, 2, 1),
    (1, 512, 1, 4, 4, 512, 3, 1, 1),
    (8, 256, 2, 7, 7, 256, 3, 1, 1),   # split-K forward / dgrad (few tiles, long K)
    (8, 128, 4, 14, 14, 256, 3, 2, 1), # split-K over stride-parity classes
    (16, 512, 1, 4, 4, 512, 3, 1, 1),
    (4, 256, 2, 7, 7, 512, 3, 2, 1),   # T 2 -> 1: the first temporal tap only sees padding (dead-tap pruning)
    (2, 64, 3, 1, 5, 64, 3, 1, 1),     # H = 1: six of nine (kh, kw) rows of taps are dead
    (2, 128, 1, 2, 2, 128, 3, 2, 1),   # 1 x 2 x 2 -> 1 x 1 x 1
]


@pytest.mark.parametrize("use_ws", [True, False])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv3d_fwd_dgrad_wgrad(case, use_ws):
    _need_gpu()
    from vc3d import ops
    ops.USE_WORKSPACE = use_ws  # False: no split-K slabs (direct kernels, atomics in wgrad)
    try:
        _conv_case(case)
    finally:
        ops.USE_WORKSPACE = True


def _conv_case(case):
    from vc3d import ops
    N, Ci, T, H, W, Co, k, s, p = case
    g = torch.Generator().manual_seed(hash(case) % (2 ** 31))
    x = bf16_round(torch.randn(N, Ci, T, H, W, generator=g))
    w = bf16_round(torch.randn(Co, Ci, k, k, k, generator=g) * (2.0 / (Ci * k ** 3)) ** 0.5)
    x.requires_grad_(True)
    w.requires_grad_(True)
    y_ref = F.conv3d(x, w, stride=s, padding=p)
    dy = bf16_round(torch.randn(y_ref.shape, generator=g))
    y_ref.backward(dy)

    d = ops.conv_desc(N, Ci, T, H, W, Co, k, s, p)
    wf, wd = ops.pack_weight(cl3d_param(w.detach()))
    xd = to_nthwc_dev(x.detach())
    stats = ops.new_stats(Co, "cuda")
    y = ops.conv3d_fwd(d, xd, wf, stats=stats)
    torch.cuda.synchronize()
    y_h = from_nthwc(y)
    assert rel_l2(y_h, y_ref.detach()) <= TOL_BF16
    assert float((y_h - y_ref.detach()).abs().max()) <= 2e-2 * float(y_ref.abs().max())
    # fused BN statistics (sum, sum of squares over N,T,H,W per channel)
    st = stats.sum(0).cpu()
    ref_s = y_ref.detach().sum(dim=(0, 2, 3, 4))
    ref_q = (y_ref.detach() ** 2).sum(dim=(0, 2, 3, 4))
    assert float((st[0] - ref_s).abs().max()) <= 1e-3 * float(ref_q.max().sqrt() * (y_ref[0, 0].numel() * N) ** 0.5)
    assert rel_l2(st[1], ref_q) <= TOL_F32
    _check_fused_finalize(lambda st2, fin: ops.conv3d_fwd(d, xd, wf, stats=st2, fin=fin), y, Co,
                          N * y_ref.shape[2] * y_ref.shape[3] * y_ref.shape[4])

    dyd = to_nthwc_dev(dy)
    dx = ops.conv3d_dgrad(d, dyd, wd)
    torch.cuda.synchronize()
    assert rel_l2(from_nthwc(dx), x.grad) <= TOL_BF16
    # accumulate form: dx2 = dx + dgrad
    dx2 = dx.clone()
    ops.conv3d_dgrad(d, dyd, wd, out=dx2, accumulate=True)
    torch.cuda.synchronize()
    assert rel_l2(from_nthwc(dx2), 2 * x.grad) <= 2 * TOL_BF16

    dw = torch.zeros(Co, k, k, k, Ci, dtype=torch.float32, device="cuda")
    ops.conv3d_wgrad(d, xd, dyd, dw)
    torch.cuda.synchronize()
    dw_h = dw.cpu().permute(0, 4, 1, 2, 3)
    assert rel_l2(dw_h, w.grad) <= TOL_F32


def _check_fused_finalize(run, y_plain, C, count):
    """BatchNorm finalize done by the producing launch's last workgroup == the separate vc_bn_finalize launch on the
    same sums (the sums themselves differ by the arrival order of float atomics only), twice in a row on ONE ticket
    word (the launch must leave it zero), running statistics included."""
    from vc3d import ops
    g = torch.Generator().manual_seed(C)
    gamma = (torch.rand(C, generator=g) + 0.5).cuda()
    beta = (torch.randn(C, generator=g) * 0.1).cuda()
    ticket = torch.zeros(64, dtype=torch.int32, device="cuda")
    rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    rm0, rv0 = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    for _ in range(2):
        st2 = ops.new_stats(C, "cuda")
        fin, prm = ops.bn_fin(ticket, count, gamma, beta, rm, rv, 0.1, 1e-5)
        y2 = run(st2, fin)
        prm0 = ops.bn_finalize(st2, count, gamma, beta, rm0, rv0, 0.1, 1e-5)
        torch.cuda.synchronize()
        assert int(ticket.abs().sum().item()) == 0
        assert torch.equal(y2, y_plain)
        assert torch.allclose(prm, prm0, rtol=1e-6, atol=1e-7), float((prm - prm0).abs().max())
        assert torch.allclose(rm, rm0, rtol=1e-6, atol=1e-7) and torch.allclose(rv, rv0, rtol=1e-6, atol=1e-7)


# the last shapes: ragged last row group + odd T (image-resident kernels: several units, frame pairs with an idle
# second frame), a wide image (3 rows per chunk, 4 output rows), and Co = 128 (the generic kernels)
@pytest.mark.parametrize("shape", [(2, 4, 16, 16), (1, 1, 12, 20), (3, 5, 8, 8), (1, 7, 80, 24), (2, 3, 8, 120),
                                   (1, 2, 16, 16, 128)])
def test_stem_fwd_wgrad(shape):
    _need_gpu()
    from vc3d import ops
    N, T, H, W = shape[:4]
    Co = shape[4] if len(shape) > 4 else 64
    g = torch.Generator().manual_seed(7 + N)
    x = bf16_round(torch.randn(N, 3, T, H, W, generator=g))
    w = bf16_round(torch.randn(Co, 3, 7, 7, 7, generator=g) * (2.0 / (64 * 343)) ** 0.5 * 4)
    w.requires_grad_(True)
    y_ref = F.conv3d(x, w, stride=(1, 2, 2), padding=3)
    dy = bf16_round(torch.randn(y_ref.shape, generator=g))
    y_ref.backward(dy)

    xs = ops.stem_pack_input(x.cuda())
    ws = ops.stem_pack_weight(cl3d_param(w.detach()))
    stats = ops.new_stats(Co, "cuda")
    y = ops.stem_fwd(xs, ws, N, T, H, W, Co, stats=stats)
    torch.cuda.synchronize()
    assert rel_l2(from_nthwc(y), y_ref.detach()) <= TOL_BF16
    st = stats.sum(0).cpu()
    assert rel_l2(st[1], (y_ref.detach() ** 2).sum(dim=(0, 2, 3, 4))) <= TOL_F32
    _check_fused_finalize(lambda st2, fin: ops.stem_fwd(xs, ws, N, T, H, W, Co, stats=st2, fin=fin), y, Co,
                          N * T * (H // 2) * (W // 2))

    dwm = torch.zeros(Co, 7, 7, 7, 3, dtype=torch.float32, device="cuda")
    ops.stem_wgrad(xs, to_nthwc_dev(dy), N, T, H, W, Co, dwm)
    torch.cuda.synchronize()
    assert rel_l2(dwm.cpu().permute(0, 4, 1, 2, 3), w.grad) <= TOL_F32
    # bf16 clip input gives the same packed source
    xs2 = ops.stem_pack_input(x.cuda().to(torch.bfloat16))
    assert torch.equal(xs, xs2)


@pytest.mark.parametrize("C,shape", [(64, (2, 4, 6, 6)), (128, (3, 2, 5, 5)), (512, (4, 1, 4, 4)), (2048, (2, 1, 2, 2))])
@pytest.mark.parametrize("mode", ["plain", "relu", "res_relu"])
def test_batchnorm_fwd_bwd(C, shape, mode):
    _need_gpu()
    from vc3d import ops
    N, T, H, W = shape
    g = torch.Generator().manual_seed(C + N)
    y = bf16_round(torch.randn(N, C, T, H, W, generator=g) * 2 + 0.5)
    res = bf16_round(torch.randn(N, C, T, H, W, generator=g))
    gamma = torch.rand(C, generator=g) + 0.5
    beta = torch.randn(C, generator=g) * 0.1
    y.requires_grad_(True)
    res.requires_grad_(True)
    bn = torch.nn.BatchNorm3d(C, momentum=0.1)
    with torch.no_grad():
        bn.weight.copy_(gamma)
        bn.bias.copy_(beta)
    bn.train()
    z_ref = bn(y)
    if mode == "res_relu":
        z_ref = z_ref + res
    if mode != "plain":
        z_ref = torch.relu(z_ref)
    dz = bf16_round(torch.randn(z_ref.shape, generator=g))
    z_ref.backward(dz)

    yd = to_nthwc_dev(y.detach())
    rows = N * T * H * W
    sums = ops.bn_stats(yd, C)
    rm = torch.zeros(C, device="cuda")
    rv = torch.ones(C, device="cuda")
    params = ops.bn_finalize(sums, rows, gamma.cuda(), beta.cuda(), rm, rv, 0.1, 1e-5)
    resd = to_nthwc_dev(res.detach()) if mode == "res_relu" else None
    z = ops.bn_apply(yd, params, C, residual=resd, relu=(mode != "plain"))
    torch.cuda.synchronize()
    assert rel_l2(from_nthwc(z), z_ref.detach()) <= TOL_BF16
    assert rel_l2(rm.cpu(), bn.running_mean) <= 1e-4
    assert rel_l2(rv.cpu(), bn.running_var) <= 1e-4

    dgamma = torch.zeros(C, device="cuda")
    dbeta = torch.zeros(C, device="cuda")
    mask = 0 if mode == "plain" else 1
    dy, gm = ops.bn_bwd(yd, to_nthwc_dev(dz), params, C, dgamma, dbeta, mask=mask, aux=z if mask else None,
                        want_g=True)
    torch.cuda.synchronize()
    assert rel_l2(from_nthwc(dy), y.grad) <= 2 * TOL_BF16
    assert rel_l2(dgamma.cpu(), bn.weight.grad) <= 5e-3
    assert rel_l2(dbeta.cpu(), bn.bias.grad) <= 5e-3
    if mode == "res_relu":
        assert rel_l2(from_nthwc(gm), res.grad) <= TOL_BF16
    # finalize folded into the reduction (its last workgroup): same result, ticket word left zero, twice in a row
    ticket = torch.zeros(64, dtype=torch.int32, device="cuda")
    for _ in range(2):
        dg3, db3 = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
        dy3, _ = ops.bn_bwd(yd, to_nthwc_dev(dz), params, C, dg3, db3, mask=mask, aux=z if mask else None, ticket=ticket)
        torch.cuda.synchronize()
        assert int(ticket.abs().sum().item()) == 0
        assert rel_l2(dy3.float(), dy.float()) <= 1e-3
        assert torch.allclose(dg3, dgamma, rtol=1e-4, atol=1e-5) and torch.allclose(db3, dbeta, rtol=1e-4, atol=1e-5)
    if mode != "plain":
        # the ReLU mask as bits (vc_bn_apply_mask, mask mode 3): same output, bit k of byte i = out[8 i + k] > 0, and a
        # backward identical to the one that reads the activation
        bits = torch.empty(yd.numel() // 8, dtype=torch.uint8, device="cuda")
        zb = ops.bn_apply(yd, params, C, residual=resd, relu=True, relu_bits=bits)
        assert torch.equal(zb, z)
        want = ((z.reshape(-1, 8) > 0).to(torch.int32) << torch.arange(8, device="cuda", dtype=torch.int32)).sum(1)
        assert torch.equal(bits.to(torch.int32), want)
        dg4, db4 = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
        dy4, gm4 = ops.bn_bwd(yd, to_nthwc_dev(dz), params, C, dg4, db4, mask=3, aux=bits, want_g=True)
        torch.cuda.synchronize()
        assert torch.equal(dy4, dy) and torch.equal(gm4, gm)
        assert torch.allclose(dg4, dgamma, rtol=1e-5, atol=1e-6) and torch.allclose(db4, dbeta, rtol=1e-5, atol=1e-6)
    if mode == "relu":
        # recomputed mask (y*scale+shift > 0) must agree with the saved-output mask
        dg2 = torch.zeros(C, device="cuda")
        db2 = torch.zeros(C, device="cuda")
        dy2, _ = ops.bn_bwd(yd, to_nthwc_dev(dz), params, C, dg2, db2, mask=2)
        torch.cuda.synchronize()
        assert rel_l2(from_nthwc(dy2), y.grad) <= 2 * TOL_BF16


def test_bn_eval_params():
    _need_gpu()
    from vc3d import ops
    C = 64
    g = torch.Generator().manual_seed(3)
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    rm, rv = torch.randn(C, generator=g), torch.rand(C, generator=g) + 0.5
    p = ops.bn_eval_params(gamma.cuda(), beta.cuda(), rm.cuda(), rv.cuda(), 1e-5).cpu()
    scale = gamma / torch.sqrt(rv + 1e-5)
    assert torch.allclose(p[2], scale, rtol=1e-5, atol=1e-6)
    assert torch.allclose(p[3], beta - rm * scale, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("shape", [(2, 64, 4, 8, 8), (1, 64, 5, 7, 9), (2, 64, 1, 6, 6), (1, 64, 3, 6, 70), (1, 128, 2, 4, 10)])
@pytest.mark.parametrize("fused_bn", [False, True])
def test_maxpool_fwd_bwd(shape, fused_bn):
    _need_gpu()
    from vc3d import ops
    N, C, T, H, W = shape
    g = torch.Generator().manual_seed(11)
    y = bf16_round(torch.randn(N, C, T, H, W, generator=g))
    d = ops.pool_desc(N, C, T, H, W, 3, 2, 1)
    yd = to_nthwc_dev(y)
    if fused_bn:
        scale = torch.rand(C, generator=g) + 0.5
        shift = torch.randn(C, generator=g) * 0.2
        params = torch.stack([torch.zeros(C), torch.ones(C), scale, shift]).cuda()
        a = bf16_round(torch.relu(y * scale.view(1, C, 1, 1, 1) + shift.view(1, C, 1, 1, 1)))
    else:
        params = None
        a = y
    a = a.clone().requires_grad_(True)
    p_ref = F.max_pool3d(a, 3, 2, 1)
    dp = bf16_round(torch.randn(p_ref.shape, generator=g))
    p_ref.backward(dp)
    out, idx = ops.maxpool3d_fwd(d, yd, params)
    din = ops.maxpool3d_bwd(d, to_nthwc_dev(dp), idx)
    torch.cuda.synchronize()
    if fused_bn and C != 64:  # the device transform is one fma, the reference a multiply and an add: a bf16 tie can fall either way
        assert rel_l2(from_nthwc(out), p_ref.detach()) <= TOL_BF16
    else:
        assert torch.equal(from_nthwc(out), p_ref.detach())
    if fused_bn:  # the tiled kernel (planes staged in LDS, integer keys) picks the thread-per-output kernel's arg-max
        os.environ["VC_POOL_FWD_GATHER"] = "1"
        try:
            out_g, idx_g = ops.maxpool3d_fwd(d, yd, params)
        finally:
            del os.environ["VC_POOL_FWD_GATHER"]
        assert torch.equal(out_g, out) and torch.equal(idx_g, idx)
    if fused_bn:
        # ties among equal (often zero) activations may pick another arg-max; where the activation is
        # positive and unique the gradient must match.  Compare after masking by (a > 0).
        m = (a.detach() > 0).float()
        assert rel_l2(from_nthwc(din) * m, a.grad * m) <= 2e-2
    else:
        assert rel_l2(from_nthwc(din), a.grad) <= TOL_BF16


@pytest.mark.parametrize("N,C,S,K", [(5, 512, 16, 101), (64, 2048, 1, 400), (3, 64, 2, 7), (9, 128, 4, 4)])
def test_avgpool_linear(N, C, S, K):
    """the head: ragged sample / class counts (the kernels work on groups of four of each), the R3D-18 and R3D-50 heads"""
    _need_gpu()
    from vc3d import ops
    g = torch.Generator().manual_seed(5)
    x = bf16_round(torch.randn(N, S, C, generator=g))
    feat = ops.gavgpool_fwd(x.to(torch.bfloat16).cuda(), N, S, C)
    assert torch.allclose(feat.cpu(), x.mean(1), rtol=1e-5, atol=1e-5)
    Wt = torch.randn(K, C, generator=g) * 0.05
    b = torch.randn(K, generator=g)
    f = feat.cpu().clone().requires_grad_(True)
    Wr = Wt.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True)
    lo_ref = F.linear(f, Wr, br)
    dl = torch.randn(N, K, generator=g)
    lo_ref.backward(dl)
    lo = ops.linear_fwd(feat, Wt.cuda(), b.cuda())
    assert torch.allclose(lo.cpu(), lo_ref.detach(), rtol=1e-4, atol=1e-4)
    dW = torch.zeros(K, C, device="cuda")
    db = torch.zeros(K, device="cuda")
    dfeat = ops.linear_bwd(dl.cuda(), feat, Wt.cuda(), dW, db)
    assert torch.allclose(dfeat.cpu(), f.grad, rtol=1e-4, atol=1e-4)
    assert torch.allclose(dW.cpu(), Wr.grad, rtol=1e-4, atol=1e-4)
    assert torch.allclose(db.cpu(), br.grad, rtol=1e-4, atol=1e-4)
    din = ops.gavgpool_bwd(dfeat, N, S, C, (N, S, C))
    ref = (f.grad / S).view(N, 1, C).expand(N, S, C)
    assert rel_l2(din.float().cpu(), ref) <= TOL_BF16


@pytest.mark.parametrize("N,K", [(64, 101), (64, 400), (3, 7), (5, 1000), (130, 64), (9, 1024)])
def test_cross_entropy_value_and_gradient_in_one_launch(N, K):
    """vc_softmax_ce (what Loss uses for the class scores on the GPU) against F.cross_entropy in fp64: the mean loss and
    d loss / d logits, through autograd with a non-unit upstream gradient; a label outside [0, K) gives NaN; Loss itself
    (reference loss.py:25-30) with and without the actor / critic terms equals its torch-only form."""
    _need_gpu()
    from vc3d.autograd import cross_entropy
    from core.loss import Loss
    import core.loss as loss_mod
    from helpers import make_config
    g = torch.Generator().manual_seed(N * K)
    logits = (torch.randn(N, K, generator=g) * 4).cuda().requires_grad_(True)
    y = torch.randint(0, K, (N,), generator=g).cuda()
    ref_in = logits.detach().double().requires_grad_(True)
    ref = F.cross_entropy(ref_in, y)
    (ref * 1.7).backward()
    out = cross_entropy(logits, y)
    (out * 1.7).backward()
    torch.cuda.synchronize()
    assert out.shape == () and abs(float(out) - float(ref)) <= 1e-5 * max(1.0, abs(float(ref)))
    assert rel_l2(logits.grad.cpu(), ref_in.grad.float().cpu()) <= 1e-5
    with torch.no_grad():
        assert abs(float(cross_entropy(logits, y)) - float(ref)) <= 1e-5 * max(1.0, abs(float(ref)))
        bad = y.clone()
        bad[0] = K
        assert torch.isnan(cross_entropy(logits, bad))
    cfg = make_config(18, K)
    crit = Loss(cfg).cuda()
    lp, adv = torch.randn(N, 1, generator=g).cuda().requires_grad_(True), torch.randn(N, 1, generator=g).cuda().requires_grad_(True)
    for extra in ((None, None), (lp, adv)):
        vals = []
        for fused in (True, False):
            loss_mod._FUSED_CE = fused
            try:
                l2 = logits.detach().clone().requires_grad_(True)
                v = crit(l2, y, *extra)
                v.backward()
                vals.append((float(v), l2.grad.clone()))
            finally:
                loss_mod._FUSED_CE = True
        assert abs(vals[0][0] - vals[1][0]) <= 1e-5 * max(1.0, abs(vals[1][0]))
        assert rel_l2(vals[0][1].cpu(), vals[1][1].cpu()) <= 1e-5


def test_layout_roundtrip_and_add():
    _need_gpu()
    from vc3d import ops
    g = torch.Generator().manual_seed(9)
    x = bf16_round(torch.randn(2, 64, 3, 4, 5, generator=g))
    xd = ops.ncthw_to_nthwc(x.cuda())
    assert torch.equal(xd.float().cpu(), x.permute(0, 2, 3, 4, 1))
    back = ops.nthwc_to_ncthw(xd)
    assert torch.equal(back.cpu(), x)
    s = ops.add(xd, xd)
    assert torch.equal(s.float().cpu(), bf16_round(2 * x).permute(0, 2, 3, 4, 1))


def test_shape_errors_are_loud():
    _need_gpu()
    from vc3d import ops, capi
    d = ops.conv_desc(1, 48, 2, 4, 4, 64, 3, 1, 1)  # Ci not a multiple of 64
    x = torch.zeros(1, 2, 4, 4, 48, dtype=torch.bfloat16, device="cuda")
    w = torch.zeros(64, 27 * 48, dtype=torch.bfloat16, device="cuda")
    with pytest.raises(capi.Vc3dError):
        ops.conv3d_fwd(d, x, w)
    with pytest.raises(capi.Vc3dError):
        ops.stem_pack_input(torch.zeros(1, 3, 2, 5, 4, device="cuda"))  # odd H
    with pytest.raises(capi.Vc3dError):
        ops.bn_apply(torch.zeros(8, 64), torch.zeros(4, 64), 64)  # CPU tensors: no fallback


@pytest.mark.parametrize("C,shape", [(64, (2, 4, 6, 6)), (256, (3, 2, 5, 5))])
@pytest.mark.parametrize("mode", ["plain", "relu"])
def test_batchnorm_backward_with_frozen_statistics(C, shape, mode):
    """eval-mode BatchNorm (running statistics) has no batch-mean terms in its backward: dy = gamma * invstd * g, dgamma
    and dbeta as in training mode but with the running statistics in x_hat.  ops.bn_bwd(frozen=True) against torch
    autograd on nn.BatchNorm3d.eval(), with and without the fused finalize."""
    _need_gpu()
    from vc3d import ops
    N, T, H, W = shape
    g = torch.Generator().manual_seed(3 * C + N)
    y = bf16_round(torch.randn(N, C, T, H, W, generator=g) * 2 + 0.5)
    y.requires_grad_(True)
    bn = torch.nn.BatchNorm3d(C)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(C, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(C, generator=g) * 0.1)
        bn.running_mean.copy_(torch.randn(C, generator=g) * 0.3 + 0.4)
        bn.running_var.copy_(torch.rand(C, generator=g) * 3 + 0.5)
    bn.eval()
    z_ref = bn(y)
    if mode == "relu":
        z_ref = torch.relu(z_ref)
    dz = bf16_round(torch.randn(z_ref.shape, generator=g))
    z_ref.backward(dz)
    yd = to_nthwc_dev(y.detach())
    params = ops.bn_eval_params(bn.weight.detach().cuda(), bn.bias.detach().cuda(), bn.running_mean.cuda(),
                                bn.running_var.cuda(), bn.eps)
    z = ops.bn_apply(yd, params, C, relu=(mode == "relu"))
    assert rel_l2(from_nthwc(z), z_ref.detach()) <= TOL_BF16
    for ticket in (None, torch.zeros(64, dtype=torch.int32, device="cuda")):
        dgamma, dbeta = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
        dy, _ = ops.bn_bwd(yd, to_nthwc_dev(dz), params, C, dgamma, dbeta, mask=2 if mode == "relu" else 0, ticket=ticket,
                           frozen=True)
        torch.cuda.synchronize()
        assert rel_l2(from_nthwc(dy), y.grad) <= TOL_BF16
        assert rel_l2(dgamma.cpu(), bn.weight.grad) <= 5e-3
        assert rel_l2(dbeta.cpu(), bn.bias.grad) <= 5e-3
        # and it is not the training-mode formula (which subtracts the batch means of g and g * x_hat)
        dg2, db2 = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
        dy_train, _ = ops.bn_bwd(yd, to_nthwc_dev(dz), params, C, dg2, db2, mask=2 if mode == "relu" else 0)
        assert rel_l2(from_nthwc(dy_train), y.grad) > 5e-2


@pytest.mark.parametrize("shape", [(2, 64, 4, 8, 8), (1, 64, 5, 7, 9), (2, 128, 1, 6, 6)])
def test_bn_bwd_through_pool_matches_unfused(shape):
    """stem backward: BN+ReLU backward fed by a gather from the pooled gradient == max-pool backward
    followed by BN+ReLU backward (mask recomputed), and == torch autograd through relu(bn(y)) -> max_pool3d."""
    _need_gpu()
    from vc3d import ops
    N, C, T, H, W = shape
    g = torch.Generator().manual_seed(11)
    y = bf16_round(torch.randn(N, C, T, H, W, generator=g) * 1.3 + 0.2)
    gamma = 1 + 0.2 * torch.randn(C, generator=g)
    beta = 0.2 * torch.randn(C, generator=g)
    gamma[3], beta[3] = 1e-3, 0.5  # |beta/gamma| large and gamma == 0: channels that keep the gather of y
    gamma[5], beta[5] = 0.0, 0.3
    yd = to_nthwc_dev(y)
    stats = ops.bn_stats(yd, C)
    rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    prm = ops.bn_finalize(stats, yd.numel() // C, gamma.cuda(), beta.cuda(), rm, rv, 0.1, 1e-5)
    pd = ops.pool_desc(N, C, T, H, W)
    pooled, idx = ops.maxpool3d_fwd(pd, yd, prm)
    dp = bf16_round(torch.randn(N, C, *pd.out_dims(), generator=g))
    dpd = to_nthwc_dev(dp)
    # unfused
    din = ops.maxpool3d_bwd(pd, dpd, idx)
    dg0, db0 = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    dy0, _ = ops.bn_bwd(yd, din, prm, C, dg0, db0, mask=2)
    # fused
    dg1, db1 = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    dy1 = ops.bn_bwd_pooled(pd, yd, dpd, idx, prm, C, dg1, db1)
    # the tiled apply kernel (the reference's 3/2/1 pool) adds the windows in the row-at-a-time kernel's order: same bits
    os.environ["VC_POOL_BWD_ROWWISE"] = "1"
    try:
        dy1r = ops.bn_bwd_pooled(pd, yd, dpd, idx, prm, C, torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda"))
    finally:
        del os.environ["VC_POOL_BWD_ROWWISE"]
    assert torch.equal(dy1r, dy1)
    # the same with y at the arg-max solved from the pool's output instead of gathered
    dg2, db2 = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    dy2 = ops.bn_bwd_pooled(pd, yd, dpd, idx, prm, C, dg2, db2, pooled=pooled)
    assert rel_l2(dy2.float().cpu(), dy1.float().cpu()) <= 2e-3
    assert rel_l2(dg2.cpu(), dg1.cpu()) <= 2e-3 and rel_l2(db2.cpu(), db1.cpu()) <= 1e-4
    # and with the finalize folded into the reduction
    ticket = torch.zeros(64, dtype=torch.int32, device="cuda")
    dg3, db3 = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    dy3 = ops.bn_bwd_pooled(pd, yd, dpd, idx, prm, C, dg3, db3, pooled=pooled, ticket=ticket)
    torch.cuda.synchronize()
    assert int(ticket.abs().sum().item()) == 0
    assert rel_l2(dy3.float().cpu(), dy2.float().cpu()) <= 1e-3
    assert torch.allclose(dg3, dg2, rtol=1e-4, atol=1e-5) and torch.allclose(db3, db2, rtol=1e-4, atol=1e-5)
    # the unfused path rounds the pool-input gradient (a sum over up to 8 windows) to bf16, the fused one keeps it in fp32
    assert rel_l2(dy1.float().cpu(), dy0.float().cpu()) <= TOL_BF16
    assert rel_l2(dg1.cpu(), dg0.cpu()) <= 2e-3 and rel_l2(db1.cpu(), db0.cpu()) <= 2e-3
    # torch reference of the whole chain
    yr = y.clone().requires_grad_()
    bn = torch.nn.BatchNorm3d(C)
    with torch.no_grad():
        bn.weight.copy_(gamma)
        bn.bias.copy_(beta)
    F.max_pool3d(bf16_round(torch.relu(bn(yr)).detach()) + (torch.relu(bn(yr)) - torch.relu(bn(yr)).detach()), 3, 2, 1).backward(dp)
    assert rel_l2(from_nthwc(dy1), yr.grad) <= 2e-2


def _clip_pack_reference(frames, crops, T, H, W, tstep, mean, std):
    """torch restatement of vc_clip_pack_u8 up to (not including) the stem packing: N x 3 x T x H x W fp32,
    v = fma(u8, 1/(255 std), -mean/std) in fp32."""
    N = frames.shape[0]
    out = torch.empty(N, 3, T, H, W)
    a = 1.0 / (255.0 * torch.tensor(std, dtype=torch.float32))
    b = -torch.tensor(mean, dtype=torch.float32) / torch.tensor(std, dtype=torch.float32)
    for n in range(N):
        t0, y0, x0, flip = [int(v) for v in crops[n]]
        c = frames[n, t0:t0 + (T - 1) * tstep + 1:tstep, y0:y0 + H, x0:x0 + W, :].float()  # T H W 3
        if flip:
            c = c.flip(2)
        out[n] = torch.addcmul(b.view(1, 1, 1, 3), c, a.view(1, 1, 1, 3)).permute(3, 0, 1, 2)
    return out


@pytest.mark.parametrize("geom", [
    # N, Ts, Hs, Ws, T, H, W, tstep
    (2, 9, 20, 28, 4, 16, 16, 2),
    (3, 5, 12, 20, 5, 12, 20, 1),    # crop = whole frame
    (1, 33, 128, 171, 16, 112, 112, 2),  # the usual 128x171 -> 112x112 training crop, every other frame
])
def test_clip_pack_u8_matches_torch_and_float_path(geom):
    """uint8 frames -> sampled / cropped / flipped / normalised bf16 in the stem's packed form.
    Bar: the packed source equals stem_pack_input of the torch restatement to <= 1 bf16 ulp (the kernel's single
    fp32 fma vs torch's addcmul may round the last fp32 bit differently before the bf16 rounding), padding exactly 0."""
    _need_gpu()
    from vc3d import ops
    N, Ts, Hs, Ws, T, H, W, tstep = geom
    g = torch.Generator().manual_seed(N * 100 + Ts)
    frames = torch.randint(0, 256, (N, Ts, Hs, Ws, 3), generator=g, dtype=torch.uint8)
    crops = torch.zeros(N, 4, dtype=torch.int32)
    for n in range(N):
        crops[n, 0] = int(torch.randint(0, Ts - (T - 1) * tstep, (1,), generator=g))
        crops[n, 1] = int(torch.randint(0, Hs - H + 1, (1,), generator=g))
        crops[n, 2] = int(torch.randint(0, Ws - W + 1, (1,), generator=g))
        crops[n, 3] = n % 2
    if N > 1:
        crops[1, 3] = 1
    mean, std = (0.43, 0.40, 0.37), (0.23, 0.22, 0.25)
    pc = ops.clip_pack_u8(frames.cuda(), crops, T, H, W, tstep, mean, std)
    ref = _clip_pack_reference(frames, crops, T, H, W, tstep, mean, std)
    xs_ref = ops.stem_pack_input(ref.cuda())
    torch.cuda.synchronize()
    assert pc.shape == (N, 3, T, H, W)
    a, b = pc.xs.float().cpu(), xs_ref.float().cpu()
    assert torch.equal(a == 0, b == 0) or float((a - b).abs().max()) < 1e-2  # same zero padding
    ulp = torch.maximum(b.abs(), torch.tensor(1e-30)) * 2.0 ** -7
    assert bool(((a - b).abs() <= ulp).all()), float(((a - b).abs() / ulp).max())
    assert float((a != b).float().mean()) < 1e-3  # and all but a handful are identical


def test_clip_pack_u8_rejects_bad_tables():
    _need_gpu()
    from vc3d import ops
    from vc3d.capi import Vc3dError
    frames = torch.zeros(1, 8, 20, 20, 3, dtype=torch.uint8, device="cuda")
    ok = torch.tensor([[0, 2, 2, 0]], dtype=torch.int32)
    ops.clip_pack_u8(frames, ok, 4, 16, 16, 2)
    for bad in ([2, 0, 0, 0], [0, 5, 0, 0], [0, 0, 5, 0], [0, 0, 0, 2], [-1, 0, 0, 0]):
        with pytest.raises(Vc3dError):
            ops.clip_pack_u8(frames, torch.tensor([bad], dtype=torch.int32), 4, 16, 16, 2)
    with pytest.raises(Vc3dError):
        ops.clip_pack_u8(frames.cpu(), ok, 4, 16, 16, 2)
    with pytest.raises(Vc3dError):
        ops.clip_pack_u8(frames, ok, 4, 16, 16, 2, std=(1.0, 0.0, 1.0))


@pytest.mark.parametrize("case", [(2, 64, 4, 12, 12, 64, 3, 1, 1), (2, 64, 4, 8, 8, 128, 1, 1, 0), (2, 128, 4, 8, 8, 128, 3, 2, 1),
                                  (8, 256, 2, 7, 7, 256, 3, 1, 1)])
def test_conv3d_fwd_inference_epilogue(case):
    """vc_conv3d_fwd_act: y = relu(scale * conv(x) + shift) in the conv epilogue == conv then vc_bn_apply, up to the one
    bf16 rounding the fused form saves (bar: 1.5 bf16 ulp of the larger operand per element; vs fp32 torch TOL_BF16).
    Covers the halo kernel, the implicit GEMM (1x1x1 and strided) and a split-K geometry (applied after its finalize)."""
    _need_gpu()
    from vc3d import ops
    N, Ci, T, H, W, Co, k, s, p = case
    g = torch.Generator().manual_seed(sum(case))
    x = bf16_round(torch.randn(N, Ci, T, H, W, generator=g))
    w = bf16_round(torch.randn(Co, Ci, k, k, k, generator=g) * (2.0 / (Ci * k ** 3)) ** 0.5)
    gamma, beta = torch.rand(Co, generator=g) + 0.5, torch.randn(Co, generator=g) * 0.2
    rm, rv = torch.randn(Co, generator=g) * 0.1, torch.rand(Co, generator=g) + 0.5
    ref = torch.relu(F.batch_norm(F.conv3d(x, w, stride=s, padding=p), rm, rv, gamma, beta, False, 0.0, 1e-5))
    d = ops.conv_desc(N, Ci, T, H, W, Co, k, s, p)
    wf, _ = ops.pack_weight(cl3d_param(w))
    xd = to_nthwc_dev(x)
    prm = ops.bn_eval_params(gamma.cuda(), beta.cuda(), rm.cuda(), rv.cuda(), 1e-5)
    z_fused = ops.conv3d_fwd(d, xd, wf, act_params=prm, relu=True)
    z_two = ops.bn_apply(ops.conv3d_fwd(d, xd, wf), prm, Co, relu=True)
    torch.cuda.synchronize()
    assert rel_l2(from_nthwc(z_fused), ref) <= TOL_BF16
    a, b = z_fused.float().cpu(), z_two.float().cpu()
    assert rel_l2(a, b) <= TOL_BF16
    with pytest.raises(Exception):
        ops.conv3d_fwd(d, xd, wf, stats=ops.new_stats(Co, "cuda"), act_params=prm)
    # residual form: relu(scale * conv + shift + res); in the implicit-GEMM epilogue for 1x1x1 / strided convs,
    # conv + in-place vc_bn_apply behind a 3x3x3 halo conv or a split-K geometry
    res = bf16_round(torch.randn(ref.shape, generator=g))
    ref_r = torch.relu(F.batch_norm(F.conv3d(x, w, stride=s, padding=p), rm, rv, gamma, beta, False, 0.0, 1e-5) + res)
    z_r = ops.conv3d_fwd(d, xd, wf, act_params=prm, relu=True, residual=to_nthwc_dev(res))
    torch.cuda.synchronize()
    assert rel_l2(from_nthwc(z_r), ref_r) <= TOL_BF16


def test_concurrent_host_threads_on_distinct_streams():
    """include/vc3d.h: calls from distinct host threads on distinct streams are safe.  The per-device set-up of a kernel
    (raising its dynamic-LDS limit) used to sit behind a plain static flag; it is a std::call_once now.  Four threads
    launch the halo conv (the kernel with the largest LDS request) at the same time, each on its own stream with its
    own workspace, first call in this process for two of the geometries; every result equals the single-threaded one."""
    import threading
    from vc3d import capi, ops
    geoms = [(4, 64, 4, 12, 12, 64), (2, 128, 2, 10, 10, 128), (4, 64, 3, 9, 11, 128), (2, 128, 4, 6, 6, 64)]
    cases = []
    for i, (N, Ci, T, H, W, Co) in enumerate(geoms):
        d = ops.conv_desc(N, Ci, T, H, W, Co, 3, 1, 1)
        g = torch.Generator(device="cuda").manual_seed(40 + i)
        x = torch.randn(N, T, H, W, Ci, device="cuda", generator=g).to(torch.bfloat16)
        w = (torch.randn(Co, 27 * Ci, device="cuda", generator=g) * 0.02).to(torch.bfloat16)
        need = int(capi.lib.vc_conv3d_workspace_bytes(d, 0))
        ws = torch.empty(max(need, 16), dtype=torch.uint8, device="cuda")
        y = torch.empty(N, T, H, W, Co, dtype=torch.bfloat16, device="cuda")
        cases.append((d, x, w, ws, need, y, torch.cuda.Stream()))
    torch.cuda.synchronize()
    errs = []

    def work(c):
        d, x, w, ws, need, y, st = c
        for _ in range(20):
            rc = capi.lib.vc_conv3d_fwd(d, x.data_ptr(), w.data_ptr(), y.data_ptr(), None, ws.data_ptr() if need else None,
                                        need, st.cuda_stream)
            if rc != 0:
                errs.append(capi.lib.vc_last_error().decode())
                return

    threads = [threading.Thread(target=work, args=(c,)) for c in cases]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    torch.cuda.synchronize()
    assert not errs, errs
    for d, x, w, ws, need, y, st in cases:
        ref = ops.conv3d_fwd(d, x, w)
        torch.cuda.synchronize()
        assert torch.equal(ref.view(torch.int16), y.view(torch.int16))


def test_kernels_beyond_two_giga_elements():
    """BASELINE config 4 (64 x 224 x 224 clips) at batch 64 puts 3.3 G elements into one tensor.  The bandwidth kernels
    take 64-bit element counts; the conv entry points run sub-batches (capi.hip conv_max_batch).  Checked here on a
    2.3 G-element tensor (36 M rows x 64 channels, 4.6 GB): each op on the whole tensor equals the same op on its last
    slice alone (the part whose offsets do not fit 32 bits) -- exact equality for the element-wise ops, 1e-5 on sums."""
    from vc3d import ops
    C = 64
    N, T, H, W = 46, 16, 224, 224  # N*T*H*W*C = 2.36e9 > 2^31; one sample = 51 M elements
    g = torch.Generator(device="cuda").manual_seed(3)
    y = torch.empty(N, T, H, W, C, dtype=torch.bfloat16, device="cuda")
    for n in range(N):
        y[n] = torch.randn(T, H, W, C, device="cuda", generator=g).to(torch.bfloat16)
    assert y.numel() > 2 ** 31
    prm = torch.stack([0.1 * torch.randn(C, device="cuda", generator=g), 1 + 0.1 * torch.rand(C, device="cuda", generator=g),
                       1 + 0.1 * torch.rand(C, device="cuda", generator=g), 0.1 * torch.randn(C, device="cuda", generator=g)])
    tail = slice(N - 4, N)
    # BatchNorm apply (+ReLU)
    z = ops.bn_apply(y, prm, C, relu=True)
    z_t = ops.bn_apply(y[tail].contiguous(), prm, C, relu=True)
    assert torch.equal(z[tail], z_t)
    # statistics: whole = sum of the per-sample-group sums
    st = ops.bn_stats(y, C).sum(0)
    st_parts = sum(ops.bn_stats(y[a:a + 8].contiguous(), C).sum(0) for a in range(0, N, 8))
    assert torch.allclose(st, st_parts, rtol=1e-5, atol=1e-2)
    # BatchNorm backward apply (mask recomputed from y) on the tail
    dz = z  # any tensor of the same shape serves as the incoming gradient
    coef = torch.stack([0.01 * torch.randn(C, device="cuda", generator=g), 0.01 * torch.randn(C, device="cuda", generator=g)])
    from vc3d import capi
    dy = torch.empty_like(y)
    capi.check(capi.lib.vc_bn_bwd_apply(y.data_ptr(), dz.data_ptr(), None, prm.data_ptr(), coef.data_ptr(), dy.data_ptr(), None,
                                        y.numel(), C, 2, capi.cur_stream()))
    yt, dzt = y[tail].contiguous(), dz[tail].contiguous()
    dy_t = torch.empty_like(yt)
    capi.check(capi.lib.vc_bn_bwd_apply(yt.data_ptr(), dzt.data_ptr(), None, prm.data_ptr(), coef.data_ptr(), dy_t.data_ptr(), None,
                                        yt.numel(), C, 2, capi.cur_stream()))
    assert torch.equal(dy[tail], dy_t)
    del dy, dy_t, dz, z_t
    # max pool with the fused BatchNorm + ReLU, and the BatchNorm backward through the pool
    pd = ops.pool_desc(N, C, T, H, W)
    out, idx = ops.maxpool3d_fwd(pd, y, prm)
    pd_t = ops.pool_desc(4, C, T, H, W)
    out_t, idx_t = ops.maxpool3d_fwd(pd_t, yt, prm)
    assert torch.equal(out[tail], out_t) and torch.equal(idx[tail], idx_t)
    dgam, dbet = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    dpool = out  # gradient stand-in
    dy0 = ops.bn_bwd_pooled(pd, y, dpool, idx, prm, C, dgam, dbet, pooled=out)
    assert bool(torch.isfinite(dy0[tail].float()).all()) and float(dy0[tail].float().abs().sum()) > 0
    # conv through the C ABI: 1x1x1, whole batch (sub-batches inside) == last samples alone
    d = ops.conv_desc(N, C, T, H, W, 64, 1, 1, 0)
    wf = (0.1 * torch.randn(64, C, device="cuda", generator=g)).to(torch.bfloat16)
    yo = ops.conv3d_fwd(d, y, wf)
    yo_t = ops.conv3d_fwd(ops.conv_desc(4, C, T, H, W, 64, 1, 1, 0), yt, wf)
    assert torch.equal(yo[tail], yo_t)


@pytest.mark.parametrize("kind", ["halo", "pointwise", "strided", "halo_splitk", "stem"])
def test_sub_batched_conv_keeps_the_shift_of_the_batchnorm_sums(kind):
    """A conv whose tensor passes the per-launch element limit runs as sub-batches (capi.hip conv_max_batch /
    stem_max_batch; BASELINE config 4 at batch 64).  The fused finalize of such a call must see sums taken relative to
    the SAME shift it adds back: with a non-zero shift (every step after the first: the engine passes the previous
    batch mean) the sub-launches used to accumulate raw sums while the finalize added the shift, doubling the mean.
    VC_MAX_TENSOR_ELEMS lowers the limit so that a small batch splits three ways; the result must equal the unsplit
    call's (same output bits; mean / invstd / running statistics / next shift to the order of the float atomics), on
    two consecutive calls that hand the shift on."""
    _need_gpu()
    from vc3d import ops
    g = torch.Generator().manual_seed(11)
    N = 5
    if kind == "stem":
        T, H, W, Co = 4, 16, 16, 64
        x = bf16_round(torch.randn(N, 3, T, H, W, generator=g) + 0.7)
        w = bf16_round(torch.randn(Co, 3, 7, 7, 7, generator=g) * 0.05 + 0.01)
        xs = ops.stem_pack_input(x.cuda())
        ws = ops.stem_pack_weight(cl3d_param(w))
        count = N * T * (H // 2) * (W // 2)
        per_clip = max(xs.numel() // N, count // N * Co)
        run = lambda st, fin: ops.stem_fwd(xs, ws, N, T, H, W, Co, stats=st, fin=fin)
    else:
        Ci, T, H, W, Co, k, s, p = {"halo": (64, 4, 6, 6, 64, 3, 1, 1), "pointwise": (64, 2, 6, 6, 128, 1, 1, 0),
                                    "strided": (64, 4, 8, 8, 128, 3, 2, 1), "halo_splitk": (256, 2, 7, 7, 256, 3, 1, 1)}[kind]
        x = bf16_round(torch.randn(N, Ci, T, H, W, generator=g) + 0.5)
        w = bf16_round(torch.randn(Co, Ci, k, k, k, generator=g) * (2.0 / (Ci * k ** 3)) ** 0.5 + 0.02)
        d = ops.conv_desc(N, Ci, T, H, W, Co, k, s, p)
        xd = to_nthwc_dev(x)
        wf, _ = ops.pack_weight(cl3d_param(w))
        To, Ho, Wo = d.out_dims()
        count = N * To * Ho * Wo
        per_clip = max(Ci * T * H * W, Co * To * Ho * Wo)
        run = lambda st, fin: ops.conv3d_fwd(d, xd, wf, stats=st, fin=fin)
    gamma = (torch.rand(Co, generator=g) + 0.5).cuda()
    beta = (torch.randn(Co, generator=g) * 0.1).cuda()

    def two_steps():
        ticket = torch.zeros(64, dtype=torch.int32, device="cuda")
        rm, rv = torch.zeros(Co, device="cuda"), torch.ones(Co, device="cuda")
        shift = torch.zeros(Co, device="cuda")
        out = []
        for _ in range(2):
            st = ops.new_stats(Co, "cuda")
            fin, prm = ops.bn_fin(ticket, count, gamma, beta, rm, rv, 0.1, 1e-5, shift=shift)
            y = run(st, fin)
            torch.cuda.synchronize()
            out.append((y.clone(), prm.clone(), shift.clone()))
        return out, rm, rv

    ref, rm0, rv0 = two_steps()
    assert float(ref[0][2].abs().min()) > 1e-3, "the test needs a non-zero shift in the second call"
    os.environ["VC_MAX_TENSOR_ELEMS"] = str(2 * per_clip)  # sub-batches of 2, 2 and 1 clips
    try:
        got, rm1, rv1 = two_steps()
    finally:
        del os.environ["VC_MAX_TENSOR_ELEMS"]
    for (y0, p0, s0), (y1, p1, s1) in zip(ref, got):
        assert torch.equal(y0, y1)
        assert torch.allclose(p0, p1, rtol=2e-5, atol=2e-6), float((p0 - p1).abs().max())
        assert torch.allclose(s0, s1, rtol=2e-5, atol=2e-6)
    assert torch.allclose(rm0, rm1, rtol=2e-5, atol=2e-6) and torch.allclose(rv0, rv1, rtol=2e-5, atol=2e-6)
    # and the mean is the mean: against fp64 on the host
    yh = got[1][0].double().view(-1, Co)
    assert torch.allclose(got[1][1][0].double(), yh.mean(0), rtol=0, atol=2e-2 * float(yh.std()))


def test_batchnorm_variance_survives_a_large_mean():
    """SURVEY section 7 flagged var = sum x^2 / n - mean^2 in fp32 as cancellation-prone.  The sums are shifted now
    (bn_fin.h): stand-alone statistics relative to row 0 of the tensor, fused conv-epilogue statistics relative to a
    caller-held shift that the finalize refreshes with the batch mean.  Stress: |mean| / sigma = 100.
      (a) vc_bn_stats + vc_bn_finalize on 64 channels x 3.2 M values (the stem's size): variance within 1e-3 of the
          fp64 variance of the same bf16 values, mean within 1e-6 relative; the unshifted form is shown beside it;
      (b) the fused path (1x1x1 conv, statistics from the fp32 accumulators, finalize by the last workgroup): first
          call with a zero shift (unshifted, may be off), second call with the shift the first one left: <= 1e-3."""
    from vc3d import capi, ops
    C, rows = 64, 3_211_264  # 64 x 16 x 56 x 56
    g = torch.Generator(device="cuda").manual_seed(8)
    sigma = 0.5 + torch.rand(C, device="cuda", generator=g)
    y = torch.empty(rows, C, dtype=torch.bfloat16, device="cuda")
    for a in range(0, rows, 401408):
        y[a:a + 401408] = (100.0 * sigma + sigma * torch.randn(401408, C, device="cuda", generator=g)).to(torch.bfloat16)
    yd = y.double()
    mean64, var64 = yd.mean(0), yd.var(0, unbiased=False)
    del yd
    gamma, beta = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    res = {}
    for tag, use_shift in (("shifted", True), ("unshifted", False)):
        shift = torch.zeros(C, device="cuda") if use_shift else None
        sums = ops.bn_stats(y, C, shift=shift)
        prm = ops.bn_finalize(sums, rows, gamma, beta, None, None, 0.1, 0.0, shift=shift)
        var = 1.0 / prm[1].double() ** 2
        res[tag] = (float(((prm[0].double() - mean64).abs() / mean64.abs()).max()), float(((var - var64).abs() / var64).max()))
    print("stand-alone statistics at |mean|/sigma = 100: (mean rel err, var rel err)", res)
    assert res["shifted"][0] <= 1e-6 and res["shifted"][1] <= 1e-3
    # (b) fused: y = conv1x1x1(x), x = 1 + 0.08 r, w = 1/64 (exact in bf16)  ->  mean 1, sigma 0.01
    N, T, H, W = 16, 8, 28, 28
    x = (1.0 + 0.08 * torch.randn(N, T, H, W, 64, device="cuda", generator=g)).to(torch.bfloat16)
    wf = torch.full((64, 64), 1.0 / 64, dtype=torch.bfloat16, device="cuda")
    yref = x.double().sum(-1, keepdim=True) / 64.0  # every output channel is the same exact mean of the 64 inputs
    m64, v64 = float(yref.mean()), float(yref.var(unbiased=False))
    assert abs(m64) / v64 ** 0.5 > 80
    d = ops.conv_desc(N, 64, T, H, W, 64, 1, 1, 0)
    shift = torch.zeros(64, device="cuda")
    ticket = torch.zeros(capi.VC_TICKET_WORDS, dtype=torch.int32, device="cuda")
    errs = []
    for call in range(2):
        stats = ops.new_stats(64, x.device)
        fin, prm = ops.bn_fin(ticket, N * T * H * W, gamma, beta, None, None, 0.1, 0.0, shift=shift)
        ops.conv3d_fwd(d, x, wf, stats=stats, fin=fin)
        torch.cuda.synchronize()
        var = 1.0 / prm[1].double() ** 2
        errs.append((float((prm[0].double() - m64).abs().max() / abs(m64)), float(((var - v64).abs() / v64).max())))
    print("fused conv statistics at |mean|/sigma = %.0f: call 1 (zero shift) %s, call 2 (shift = last mean) %s"
          % (abs(m64) / v64 ** 0.5, errs[0], errs[1]))
    assert float((shift.double() - m64).abs().max()) <= 1e-5 * abs(m64)
    assert errs[1][0] <= 1e-6 and errs[1][1] <= 1e-3


@pytest.mark.parametrize("geom", [(64, 256, 8, 28, 28, 64), (64, 64, 8, 28, 28, 256), (2, 2048, 1, 2, 2, 512),
                                  (3, 256, 2, 7, 7, 1024), (5, 128, 4, 14, 14, 512), (1, 64, 1, 3, 5, 64)])
def test_pointwise_kernel_matches_general_kernel(geom):
    """pw_conv.hip (the dense 1x1x1 GEMM: three workgroups per CU, BatchNorm finalize per co block) against the general
    implicit-GEMM kernel on the same inputs (VC_NO_PW_KERNEL is read per call): bit-equal outputs in forward, dgrad and
    accumulating dgrad -- same tile, same summation order -- and BatchNorm parameters / running statistics equal to the
    order of the float atomics.  The first two geometries give a workgroup several tiles of several K chunks."""
    _need_gpu()
    from vc3d import ops
    N, Ci, T, H, W, Co = geom
    d = ops.conv_desc(N, Ci, T, H, W, Co, 1, 1, 0)
    M = N * T * H * W
    g = torch.Generator().manual_seed(Ci + Co)
    x = torch.randn(N, T, H, W, Ci, generator=g).to(torch.bfloat16).cuda()
    dy = torch.randn(N, T, H, W, Co, generator=g).to(torch.bfloat16).cuda()
    w = (torch.randn(Co, Ci, 1, 1, 1, generator=g) * 0.05).cuda().contiguous(memory_format=torch.channels_last_3d)
    wf, wd = ops.pack_weight(w)
    gamma, beta = (torch.rand(Co, generator=g) + 0.5).cuda(), torch.randn(Co, generator=g).cuda()
    base = torch.randn(N, T, H, W, Ci, generator=g).to(torch.bfloat16).cuda()

    def run():
        stats = ops.new_stats(Co, x.device)
        rm, rv = torch.zeros(Co, device="cuda"), torch.ones(Co, device="cuda")
        ticket = torch.zeros(64, dtype=torch.int32, device="cuda")
        fin, params = ops.bn_fin(ticket, M, gamma, beta, rm, rv, 0.1, 1e-5)
        y = ops.conv3d_fwd(d, x, wf, stats=stats, fin=fin)
        dx = ops.conv3d_dgrad(d, dy, wd)
        acc = base.clone()
        ops.conv3d_dgrad(d, dy, wd, out=acc, accumulate=True)
        torch.cuda.synchronize()
        assert int(ticket.abs().sum()) == 0
        return y, dx, acc, params.clone(), rm, rv

    os.environ["VC_NO_PW_KERNEL"] = "1"
    try:
        ref = run()
    finally:
        del os.environ["VC_NO_PW_KERNEL"]
    got = run()
    for a, b in zip(got[:3], ref[:3]):
        assert torch.equal(a, b)
    for a, b in zip(got[3:], ref[3:]):
        assert rel_l2(a.cpu(), b.cpu()) <= 1e-5


@pytest.mark.parametrize("geom", [(64, 64, 8, 28, 28, 256, 1), (4, 128, 4, 14, 14, 512, 1), (3, 256, 2, 7, 7, 1024, 1),
                                  (2, 512, 1, 4, 4, 2048, 1), (1, 64, 1, 3, 5, 64, 1), (2, 64, 2, 9, 9, 64, 3),
                                  (8, 64, 8, 28, 28, 64, 3), (6, 128, 4, 14, 14, 128, 3), (16, 256, 2, 7, 7, 256, 3),
                                  (3, 128, 3, 10, 6, 64, 3)])
@pytest.mark.parametrize("accumulate,use_bits", [(False, True), (True, True), (False, False), ("masked", True)])
def test_dgrad_with_folded_bn_backward_reduction(geom, accumulate, use_bits):
    """vc_conv3d_dgrad_fused's reduction (dense geometries: the BatchNorm backward sums of the tensor the data-gradient GEMM
    completes are taken in its epilogue, pw_conv.hip) against the two launches it replaces, vc_conv3d_dgrad +
    vc_bn_bwd_reduce_fin with the ReLU bits (mask mode 3) or with the mask recomputed from y (mode 2, no residual): the
    same dx bit for bit, and coef / dgamma / dbeta equal to the order of the float additions (both reduce the ROUNDED
    dx).  `geom` = the conv (Co -> its input Ci = the BatchNorm's channels).  The 3x3x3 ones take the halo kernel
    (plain and split-K plans): the entry point runs the plain plans as the two launches itself and, with split-K slabs,
    takes the reduction in the slab finalize launch (splitk_finalize_kernel<2|3>).  accumulate == "masked": the
    existing dx counts only where its own ReLU bit is set (acc_relu_bits: the identity shortcut's gradient), against
    masking it first with torch."""
    _need_gpu()
    from vc3d import ops
    N, Co, T, H, W, Ci, k = geom
    d = ops.conv_desc(N, Ci, T, H, W, Co, k, 1, k // 2)
    M = N * T * H * W
    g = torch.Generator().manual_seed(Ci + Co + k)
    dy = torch.randn(N, T, H, W, Co, generator=g).to(torch.bfloat16).cuda()
    w = (torch.randn(Co, Ci, k, k, k, generator=g) * 0.05).cuda().contiguous(memory_format=torch.channels_last_3d)
    _, wd = ops.pack_weight(w)
    base = torch.randn(N, T, H, W, Ci, generator=g).to(torch.bfloat16).cuda()
    # the BatchNorm whose backward consumes dx: its conv output y, parameters, residual and the bits of its output
    y = (torch.randn(N, T, H, W, Ci, generator=g) * 1.5 + 0.3).to(torch.bfloat16).cuda()
    res = torch.randn(N, T, H, W, Ci, generator=g).to(torch.bfloat16).cuda()
    gamma, beta = (torch.rand(Ci, generator=g) + 0.5).cuda(), torch.randn(Ci, generator=g).cuda()
    params = ops.bn_finalize(ops.bn_stats(y, Ci), M, gamma, beta, None, None, 0.1, 1e-5)
    bits = None
    if use_bits:
        bits = torch.empty(M * Ci // 8, dtype=torch.uint8, device="cuda")
        ops.bn_apply(y, params, Ci, residual=res, relu=True, relu_bits=bits)
    mask = 3 if use_bits else 2

    acc_bits = None
    if accumulate == "masked":
        keep = torch.rand(M * Ci, generator=g).view(-1, 8) > 0.4
        acc_bits = (keep.to(torch.uint8) << torch.arange(8, dtype=torch.uint8)).sum(1).to(torch.uint8).cuda()
        masked_base = torch.where(keep.view(base.shape).cuda(), base, torch.zeros_like(base))

    def run(folded):
        dx = base.clone() if accumulate else torch.empty_like(base)
        if acc_bits is not None and not folded:
            dx = masked_base.clone()
        dgamma, dbeta = torch.full((Ci,), 0.25, device="cuda"), torch.full((Ci,), -0.5, device="cuda")  # += targets
        coef = torch.empty(2, Ci, device="cuda")
        ticket = torch.zeros(64, dtype=torch.int32, device="cuda")
        sums = ops.new_stats(Ci, "cuda")
        if folded:
            ops.conv3d_dgrad(d, dy, wd, out=dx, accumulate=bool(accumulate), acc_bits=acc_bits,
                             bnred=(y, params, bits, sums, ticket, coef, dgamma, dbeta))
            dyb, gm = ops.bn_bwd(y, dx, params, Ci, dgamma, dbeta, mask=mask, aux=bits, want_g=True, coef=coef, reduced=True)
        else:
            ops.conv3d_dgrad(d, dy, wd, out=dx, accumulate=bool(accumulate))
            dyb, gm = ops.bn_bwd(y, dx, params, Ci, dgamma, dbeta, mask=mask, aux=bits, want_g=True, sums=sums, coef=coef,
                                 ticket=ticket)
        torch.cuda.synchronize()
        assert int(ticket.abs().sum()) == 0
        return dx, coef, dgamma, dbeta, dyb, gm

    ref = run(False)
    got = run(True)
    assert torch.equal(got[0], ref[0])
    for a, b in zip(got[1:4], ref[1:4]):
        assert torch.allclose(a, b, rtol=2e-4, atol=2e-4 * float(b.abs().max())), float((a - b).abs().max())
    assert torch.equal(got[5], ref[5])
    assert rel_l2(got[4].float().cpu(), ref[4].float().cpu()) <= 2e-3  # bf16 outputs of coef that differ in the last bits
