"""GPU parity, whole network, end to end: product path (HIP kernels, bf16 storage, bf16 MFMA
with fp32 accumulation) against the fp32 torch oracle on identical weights and inputs.

Stated tolerance.  Forward: logits rel-L2 <= 3e-2 with top-1 agreement, loss within 2 %,
running statistics rel-L2 <= 3e-2.  Gradients: at random initialisation the fp32 oracle's
own gradient is chaotic (a 1e-4 relative input perturbation moves it by 5 % rel-L2, 1e-3 by
15 %: ReLU masks flip), so no bf16 path can meet rel-L2 <= 1e-2 end to end; north_star's
1e-2 figure is met per kernel (tests/test_ops_gpu.py) and per layer instance inside a real
pass (tests/test_teacher_forced_gpu.py).  End to end we require that the product is no
further from the fp32 oracle than the oracle's own CPU bf16-storage emulation
(oracle.r3d_torch.forward_emulated) is: err(product) <= 1.25 * err(emulation) + 0.02, and
cosine similarity of the full gradient >= 0.85.
"""
import pytest
import torch
import torch.nn.functional as F

from helpers import build_pair, rel_l2

pytestmark = pytest.mark.gpu


def _flat_grads(named, rename=None):
    return {(rename(n) if rename else n): p.grad.detach().float().cpu().clone() for n, p in named}


def _run_pair(resnet_type, classes, n, t, hw, seed=1234):
    from oracle.r3d_torch import synthetic_batch, forward_emulated
    oracle, prod, cfg = build_pair(resnet_type, classes)
    x, y = synthetic_batch(n, t, hw, hw, classes, seed)
    oracle.train()
    sd0 = {k: v.clone() for k, v in oracle.state_dict().items()}
    lo_emu = forward_emulated(oracle, x)
    F.cross_entropy(lo_emu, y).backward()
    g_emu = _flat_grads(oracle.named_parameters())
    oracle.load_state_dict(sd0)  # undo the emulation pass's running-stat update
    oracle.zero_grad()
    lo_ref = oracle(x)
    loss_ref = F.cross_entropy(lo_ref, y)
    loss_ref.backward()
    g_ref = _flat_grads(oracle.named_parameters())
    prod.train()
    _, lo, _ = prod(x.cuda())
    loss = F.cross_entropy(lo, y.cuda())
    loss.backward()
    torch.cuda.synchronize()
    g_prod = _flat_grads(prod.named_parameters(), lambda k: k.replace("clf_haed.", "clf_head."))
    return oracle, prod, lo_ref.detach(), lo.detach(), lo_emu.detach(), loss_ref, loss, g_ref, g_prod, g_emu


def _cat(d, keys):
    return torch.cat([d[k].flatten() for k in keys]).double()


def _check(res, tol_logit=3e-2):
    oracle, prod, lo_ref, lo, lo_emu, loss_ref, loss, g_ref, g_prod, g_emu = res
    e_logit = rel_l2(lo, lo_ref)
    e_logit_emu = rel_l2(lo_emu, lo_ref)
    keys = sorted(g_ref)
    assert sorted(g_prod) == keys
    a, b, c = _cat(g_prod, keys), _cat(g_ref, keys), _cat(g_emu, keys)
    e_grad = float((a - b).norm() / b.norm())
    e_grad_emu = float((c - b).norm() / b.norm())
    cos = float(torch.dot(a, b) / (a.norm() * b.norm()))
    print("logits rel-L2 %.3e (emulation %.3e) | all-grad rel-L2 %.3e (emulation %.3e) cos %.4f"
          % (e_logit, e_logit_emu, e_grad, e_grad_emu, cos))
    assert e_logit <= tol_logit, "logits rel-L2 %.3e" % e_logit
    assert torch.equal(lo.argmax(1).cpu(), lo_ref.argmax(1)), "top-1 disagreement"
    assert abs(float(loss) - float(loss_ref)) <= 2e-2 * abs(float(loss_ref))
    assert e_grad <= 1.25 * e_grad_emu + 0.02, "grad error %.3e vs emulation %.3e" % (e_grad, e_grad_emu)
    assert cos >= 0.85
    rb = dict(oracle.named_buffers())
    for name, buf in prod.named_buffers():
        if name.endswith("running_mean") or name.endswith("running_var"):
            e = rel_l2(buf, rb[name])
            assert e <= 3e-2, "%s rel-L2 %.3e" % (name, e)


def test_r3d18_small_clip():
    _check(_run_pair(18, 101, n=4, t=8, hw=64), tol_logit=5e-2)


def test_r3d18_odd_crop_gradients():
    """33 x 33 crops (the engine adds a zero row and column): logits, loss, every gradient and the running statistics"""
    _check(_run_pair(18, 101, n=4, t=5, hw=33), tol_logit=5e-2)


def test_r3d18_full_clip_16x112x112():
    _check(_run_pair(18, 101, n=2, t=16, hw=112))


def test_eval_mode_matches_oracle():
    from oracle.r3d_torch import synthetic_batch
    oracle, prod, cfg = build_pair(18, 101)
    x, y = synthetic_batch(3, 8, 64, 64, 101, 99)
    # make the running statistics non-trivial first
    oracle.train()
    prod.train()
    with torch.no_grad():
        oracle(x)
        prod(x.cuda())
    oracle.eval()
    prod.eval()
    with torch.no_grad():
        lo_ref = oracle(x)
        _, lo, _ = prod(x.cuda())
    assert rel_l2(lo, lo_ref) <= 5e-2
    assert torch.equal(lo.argmax(1).cpu(), lo_ref.argmax(1))


@pytest.mark.parametrize("depth", [34, 101, 152])
def test_other_depths_of_resnet_spec(depth):
    """backbone.py:188-192 lists five depths; 18 and 50 carry the benchmarks, the other three share their kernels
    (BasicBlock [3, 4, 6, 3]; Bottleneck [3, 4, 23, 3] and [3, 8, 36, 3]).  One training step: logits against the fp32
    oracle with the oracle's bf16-storage emulation as the yardstick for what 100+ layers of bf16 activations cost at
    random initialisation (as test_golden_gpu does for R3D-50), loss, finite gradients for every parameter."""
    from oracle.r3d_torch import synthetic_batch, forward_emulated
    oracle, prod, cfg = build_pair(depth, 101)
    x, y = synthetic_batch(3, 8, 64, 64, 101, 7 + depth)
    oracle.train()
    prod.train()
    sd0 = {k: v.clone() for k, v in oracle.state_dict().items()}
    with torch.no_grad():
        lo_emu = forward_emulated(oracle, x)
    oracle.load_state_dict(sd0)
    with torch.no_grad():
        lo_ref = oracle(x)
    loss_ref = F.cross_entropy(lo_ref, y)
    _, lo, _ = prod(x.cuda())
    loss = F.cross_entropy(lo, y.cuda())
    loss.backward()
    torch.cuda.synchronize()
    e, e_emu = rel_l2(lo.detach(), lo_ref), rel_l2(lo_emu, lo_ref)
    print("R3D-%d train-mode logits rel-L2 %.3e (bf16-storage emulation %.3e)" % (depth, e, e_emu))
    assert e <= max(5e-2, 1.25 * e_emu + 5e-3)
    assert abs(float(loss.detach()) - float(loss_ref)) <= 3e-2 * abs(float(loss_ref)), (float(loss.detach()), float(loss_ref))
    names = [n for n, _ in oracle.named_parameters()]
    got = {n.replace("clf_haed.", "clf_head."): p for n, p in prod.named_parameters()}
    assert sorted(got) == sorted(names)
    for n, p in got.items():
        assert p.grad is not None and bool(torch.isfinite(p.grad).all()), n
        assert float(p.grad.abs().max()) > 0, n


def test_grad_accumulation_and_zero_grad():
    from oracle.r3d_torch import synthetic_batch
    _, prod, _ = build_pair(18, 101)
    x, y = synthetic_batch(2, 4, 32, 32, 101, 5)
    xc, yc = x.cuda(), y.cuda()
    prod.train()
    eng = prod.backbone.engine
    eng.first_stats = False

    def step():
        # the same forward three times: the BatchNorm sums must be taken relative to the same shift each time (the
        # engine moves it to the last batch mean), or they differ in their last bits and this 2-clip net amplifies that
        for cb in eng.all_convbn():
            if cb.shift is not None:
                cb.shift.zero_()
        _, lo, _ = prod(xc)
        F.cross_entropy(lo, yc).backward()

    step()
    g1 = {n: p.grad.detach().clone() for n, p in prod.named_parameters()}
    step()  # accumulates
    for n, p in prod.named_parameters():
        if n.endswith("conv2.weight") or n.startswith("clf_haed"):
            assert rel_l2(p.grad, 2 * g1[n]) <= 2e-2, n
    for p in prod.parameters():
        p.grad = None
    step()
    for n, p in prod.named_parameters():
        if n.endswith("conv2.weight") or n.startswith("clf_haed"):
            assert rel_l2(p.grad, g1[n]) <= 2e-2, n


def test_clip_gradient_is_refused_loudly():
    from vc3d.capi import Vc3dError
    _, prod, _ = build_pair(18, 101)
    prod.train()
    x = torch.randn(2, 3, 4, 32, 32, device="cuda", requires_grad=True)
    with pytest.raises(Vc3dError, match="input clip"):
        prod(x)
    with torch.no_grad():  # no graph: fine
        prod(x)


def test_eval_mode_backward_with_frozen_batchnorm_statistics():
    """model.eval() with gradients on -- fine-tuning with frozen BatchNorm statistics: the forward normalises with the
    running statistics (which must not move), the backward has no batch-mean terms (dy = scale * g) and still makes
    dgamma / dbeta.  Against torch autograd on the oracle in eval mode; the logits equal the no_grad inference path's
    up to the rounding points (that path rounds once, after BatchNorm; this one keeps the conv output in bf16)."""
    from oracle.r3d_torch import synthetic_batch
    # R3D-50 at random initialisation is the noisy case (eval logits ~12 % from fp32, gradients ~50 %, norms within 13 %:
    # bf16 storage through 50 unnormalised layers); it is here for the Bottleneck plumbing, R3D-18 carries the tight bars
    for depth, tol_lo, tol_g, tol_cos, tol_n in ((18, 5e-2, 0.25, 0.95, 0.10), (50, 0.15, 0.70, 0.75, 0.20)):
        oracle, prod, _ = build_pair(depth, 101)
        x, y = synthetic_batch(4, 4, 64, 64, 101, 21)
        oracle.train()
        prod.train()
        with torch.no_grad():  # one training step's worth of running statistics
            oracle(x)
            prod(x.cuda())
        oracle.eval()
        prod.eval()
        rs0 = {n: b.clone() for n, b in prod.named_buffers()}
        lo_ref = oracle(x)
        F.cross_entropy(lo_ref, y).backward()
        _, lo, _ = prod(x.cuda())
        F.cross_entropy(lo, y.cuda()).backward()
        torch.cuda.synchronize()
        assert rel_l2(lo.detach(), lo_ref.detach()) <= tol_lo
        with torch.no_grad():
            assert rel_l2(prod(x.cuda())[1], lo.detach()) <= tol_lo
        for n, b in prod.named_buffers():
            assert torch.equal(b, rs0[n]), n  # eval mode: no running-statistics update, no step count
        g_ref = {n: p.grad for n, p in oracle.named_parameters()}
        # The formula itself is pinned per layer at bf16 tolerance (test_ops_gpu: ...frozen_statistics); here the plumbing:
        # bf16 rounding accumulates along the backward chain (R3D-18: 1 % at the head, ~15 % at the stem, measured), a
        # wrong term anywhere moves a whole layer's gradient norm, so the norms are held to a few percent
        worst_ratio = 0.0
        for n, p in prod.named_parameters():
            r = g_ref[n.replace("clf_haed.", "clf_head.")]
            worst_ratio = max(worst_ratio, abs(float(p.grad.float().norm().cpu() / r.norm()) - 1.0))
        a = torch.cat([p.grad.flatten().float().cpu() for _, p in prod.named_parameters()]).double()
        b = torch.cat([g_ref[n.replace("clf_haed.", "clf_head.")].flatten() for n, _ in prod.named_parameters()]).double()
        e_all = float((a - b).norm() / b.norm())
        cos = float(torch.dot(a, b) / (a.norm() * b.norm()))
        print("R3D-%d frozen-statistics backward: all-gradient rel-L2 %.3e cos %.4f, worst norm deviation %.3e"
              % (depth, e_all, cos, worst_ratio))
        assert e_all <= tol_g and cos >= tol_cos and worst_ratio <= tol_n


def test_mixed_batchnorm_modes_are_refused():
    from vc3d.capi import Vc3dError
    _, prod, _ = build_pair(18, 101)
    prod.train()
    prod.backbone.layer2[0].bn1.eval()
    with pytest.raises(Vc3dError, match="different train / eval mode"):
        prod(torch.randn(2, 3, 4, 32, 32, device="cuda"))
    prod.train()
    prod(torch.randn(2, 3, 4, 32, 32, device="cuda"))


def test_cpu_input_raises():
    from vc3d.capi import Vc3dError
    _, prod, _ = build_pair(18, 101)
    with pytest.raises(Vc3dError):
        prod(torch.zeros(1, 3, 4, 32, 32))


@pytest.mark.parametrize("n,t,h,w", [(1, 3, 34, 50), (2, 7, 98, 130), (5, 2, 66, 34), (7, 4, 48, 80), (1, 16, 112, 112),
                                     (2, 5, 33, 47), (3, 4, 65, 64)])
def test_unusual_clip_shapes(n, t, h, w):
    """one clip, odd frame counts, non-square, not-multiple-of-16 and odd crops (ragged tiles in every kernel; an odd
    extent gets a zero row / column, which the reference's padded conv sees there as well): a training step
    and the eval-mode logits against the oracle (tests/shape_sweep.py: measured <= 3.5e-2 / 2.1e-2)"""
    from oracle.r3d_torch import synthetic_batch
    oracle, prod, _ = build_pair(18, 101)
    x, y = synthetic_batch(n, t, h, w, 101, 11)
    oracle.train()
    prod.train()
    with torch.no_grad():
        lo_ref = oracle(x)
    loss_ref = F.cross_entropy(lo_ref, y)
    _, lo, _ = prod(x.cuda())
    loss = F.cross_entropy(lo, y.cuda())
    loss.backward()
    torch.cuda.synchronize()
    assert rel_l2(lo.detach(), lo_ref) <= 6e-2
    assert abs(float(loss.detach()) - float(loss_ref)) <= 2e-2 * float(loss_ref)
    assert all(bool(torch.isfinite(p.grad).all()) for p in prod.parameters())
    oracle.eval()
    prod.eval()
    with torch.no_grad():
        assert rel_l2(prod(x.cuda())[1], oracle(x)) <= 4e-2


def test_one_value_per_channel_is_refused_like_torch_batchnorm():
    """a 1 x 1 x 32 x 32 clip leaves layer4 one value per channel: nn.BatchNorm3d (the reference's layers) raises in
    training mode and computes in eval mode; so does the product"""
    from helpers import make_config
    from models.model import create_model
    m = create_model(make_config(18, 101), is_train=True).cuda()
    x = torch.randn(1, 3, 1, 32, 32).cuda()
    m.train()
    with pytest.raises(ValueError, match="Expected more than 1 value per channel when training"):
        m(x)
    m.eval()
    with torch.no_grad():
        _, lo, _ = m(x)
    assert bool(torch.isfinite(lo).all())
    m.train()  # the refused step left the engine usable
    _, lo, _ = m(torch.randn(2, 3, 4, 32, 32).cuda())
    lo.sum().backward()
    torch.cuda.synchronize()
    assert all(bool(torch.isfinite(p.grad).all()) for p in m.parameters())


def test_packed_uint8_clips_give_the_float_path_logits():
    """the model takes ops.clip_pack_u8's PackedClips in place of the float clip tensor: same logits as feeding the
    torch-normalised float clip (both round the same values to bf16; <= 1 ulp differences in a handful of inputs)."""
    if not torch.cuda.is_available():
        pytest.skip("needs the GPU")
    from vc3d import ops
    _, prod, _ = build_pair(18, 11, device="cuda:0")
    prod.eval()
    g = torch.Generator().manual_seed(5)
    frames = torch.randint(0, 256, (2, 12, 40, 48, 3), generator=g, dtype=torch.uint8)
    crops = torch.tensor([[1, 3, 5, 0], [0, 8, 16, 1]], dtype=torch.int32)
    mean, std = (0.45, 0.45, 0.45), (0.225, 0.225, 0.225)
    pc = ops.clip_pack_u8(frames.cuda(), crops, 8, 32, 32, 1, mean, std)
    x = torch.stack([((frames[n, c[0]:c[0] + 8, c[1]:c[1] + 32, c[2]:c[2] + 32].float() / 255.0
                       - torch.tensor(mean)) / torch.tensor(std)).permute(3, 0, 1, 2) for n, c in enumerate(crops.tolist())])
    x[1] = x[1].flip(-1)
    with torch.no_grad():
        _, lo_u8, _ = prod(pc)
        _, lo_f, _ = prod(x.cuda())
    torch.cuda.synchronize()
    assert rel_l2(lo_u8, lo_f) <= 2e-3


def test_full_bench_size_batch_split_property():
    """BASELINE config 2's full size (64 x 3 x 16 x 112 x 112, R3D-18, 101 classes), where the CPU oracle is too slow to
    be the checker: size-independent properties instead.  Eval mode (running statistics, so samples do not interact):
    the logits of the batch of 64 equal those of its two halves run separately (tilings, split-K plans and
    persistent-grid shapes all differ between M = 64 and M = 32 clips, and with them where the inference epilogue rounds:
    once when fused, twice behind a split-K conv; bar: rel-L2 <= 1e-2, i.e. bf16 rounding noise, and the same top-1).  Train mode:
    one step at the full batch gives a finite loss near ln(101) (within 0.05 ln(101) + 0.5: random initialisation)
    and finite gradients in every parameter."""
    if not torch.cuda.is_available():
        pytest.skip("needs the GPU")
    import math
    _, prod, _ = build_pair(18, 101, device="cuda:0")
    g = torch.Generator().manual_seed(11)
    x = torch.randn(64, 3, 16, 112, 112, generator=g).to(torch.bfloat16).cuda()
    y = torch.randint(0, 101, (64,), generator=g).cuda()
    prod.eval()
    with torch.no_grad():
        full = prod(x)[1].float()
        halves = torch.cat([prod(x[:32])[1], prod(x[32:])[1]]).float()
    torch.cuda.synchronize()
    assert rel_l2(full, halves) <= 1e-2
    assert bool((full.argmax(1) == halves.argmax(1)).all())
    prod.train()
    loss = F.cross_entropy(prod(x)[1], y)
    loss.backward()
    torch.cuda.synchronize()
    assert math.isfinite(float(loss)) and abs(float(loss) - math.log(101)) <= 0.05 * math.log(101) + 0.5
    assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in prod.parameters())


@pytest.mark.parametrize("depth,classes,batch,t,hw", [(50, 400, 64, 16, 112), (50, 400, 32, 64, 224), (50, 400, 64, 64, 224)])
def test_full_size_r3d50_batch_split_property(depth, classes, batch, t, hw):
    """BASELINE config 3's per-GPU batch (64 x 3 x 16 x 112 x 112, R3D-50, 400 classes) and config 4 (64 x 224 x 224
    clips) at batch 32 and at batch 64, where the stem output has 3.3 G elements: beyond the 32-bit element offsets
    of the conv kernels, so the C ABI runs the convs as sub-batches (capi.hip: conv_max_batch) while the halves of the
    comparison run unsplit.  The same size-independent properties as the R3D-18 test above.
    Eval mode: the logits of the whole batch equal those of its two halves run separately (rel-L2 <= 1e-2, same top-1
    wherever the top-2 margin exceeds the noise).  Train mode: one step gives a finite loss near ln(classes) and
    finite gradients in every parameter."""
    if not torch.cuda.is_available():
        pytest.skip("needs the GPU")
    import math
    _, prod, _ = build_pair(depth, classes, device="cuda:0")
    g = torch.Generator().manual_seed(13)
    x = torch.empty(batch, 3, t, hw, hw, dtype=torch.bfloat16)
    for i in range(batch):  # clip by clip: a float32 staging copy of config 4's batch would be 38 GB
        x[i] = torch.randn(3, t, hw, hw, generator=g).to(torch.bfloat16)
    x = x.cuda()
    y = torch.randint(0, classes, (batch,), generator=g).cuda()
    prod.eval()
    h = batch // 2
    with torch.no_grad():
        full = prod(x)[1].float()
        halves = torch.cat([prod(x[:h])[1], prod(x[h:])[1]]).float()
    torch.cuda.synchronize()
    assert rel_l2(full, halves) <= 1e-2
    top2 = full.topk(2, dim=1).values
    sure = (top2[:, 0] - top2[:, 1]) > 2e-2 * full.abs().max()
    assert bool((full.argmax(1) == halves.argmax(1))[sure].all())
    prod.train()
    loss = F.cross_entropy(prod(x)[1], y)
    loss.backward()
    torch.cuda.synchronize()
    assert math.isfinite(float(loss)) and abs(float(loss) - math.log(classes)) <= 0.05 * math.log(classes) + 0.5
    assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in prod.parameters())


def test_fused_and_separate_batchnorm_finalize_agree():
    """the engine's default (BatchNorm finalize done by the last workgroup of the producing launch) against separate
    finalize launches (VC_NO_BN_FIN): same arithmetic on sums that differ only by the arrival order of float atomics,
    so logits and running statistics agree to 2e-3 and gradients to 5e-2 (round-off amplified through ReLU masks)."""
    if not torch.cuda.is_available():
        pytest.skip("needs the GPU")
    from vc3d import engine as eng_mod
    from oracle.r3d_torch import synthetic_batch
    x, y = synthetic_batch(2, 8, 64, 64, 7, 3)
    out = {}
    keep = eng_mod.FUSED_BN_FIN
    try:
        for fused in (True, False):
            eng_mod.FUSED_BN_FIN = fused
            _, prod, _ = build_pair(18, 7, device="cuda:0")
            prod.train()
            lo = prod(x.cuda())[1]
            F.cross_entropy(lo, y.cuda()).backward()
            torch.cuda.synchronize()
            out[fused] = (lo.detach().float().cpu(), prod.backbone.bn1.running_var.detach().cpu().clone(),
                          torch.cat([p.grad.detach().float().flatten().cpu() for p in prod.parameters()]))
    finally:
        eng_mod.FUSED_BN_FIN = keep
    # logits and running statistics: forward only, last-bit differences stay small; gradients: a ReLU mask that flips
    # on a last-bit difference moves a whole element, which this tiny net amplifies (the sharp comparison of the two
    # finalize forms on identical sums is tests/test_ops_gpu.py::_check_fused_finalize)
    for (a, b), tol in zip(zip(out[True], out[False]), (2e-3, 2e-3, 5e-2)):
        assert rel_l2(a, b) <= tol


def test_production_order_matches_the_traced_order(monkeypatch):
    """The per-layer parity test (tests/test_teacher_forced_gpu.py) runs with engine.trace set, which keeps the weight
    gradients on the main stream and takes the stem's BatchNorm backward through a materialised pool-input gradient.
    Here the production order (weight gradients on the side stream behind events, the pooled shortcut, ReLU masks as
    bits) runs the same step: every gradient of the residual stages must agree with the traced run to within the
    rerun noise of this configuration (4x), the stem's (whose pool-input gradient stays fp32 in the shortcut instead of
    being rounded to bf16) to that or 5e-3.  A missing event between the two streams shows as an O(1) deviation."""
    if not torch.cuda.is_available():
        pytest.skip("needs the GPU")
    from oracle.r3d_torch import synthetic_batch
    monkeypatch.setenv("VC_HALO_NO_PAD", "1")  # one summation order in the BatchNorm atomics of the halo kernels
    x, y = synthetic_batch(8, 8, 64, 64, 7, 5)

    def run(trace):
        _, prod, _ = build_pair(18, 7, device="cuda:0")
        eng = prod.backbone.engine
        if trace:
            eng.trace = {}
        prod.train()
        lo = prod(x.cuda())[1]
        F.cross_entropy(lo, y.cuda()).backward()
        torch.cuda.synchronize()
        return {n: p.grad.detach().float().cpu().clone() for n, p in prod.named_parameters()}, lo.detach().float().cpu()

    g_a, lo_a = run(False)
    g_b, lo_b = run(False)
    g_t, lo_t = run(True)
    noise = max(rel_l2(g_b[n], g_a[n]) for n in g_a)
    print("rerun noise %.2e; logits traced vs production %.2e" % (noise, rel_l2(lo_t, lo_a)))
    assert rel_l2(lo_t, lo_a) <= max(4 * rel_l2(lo_b, lo_a), 1e-6)
    worst = {}
    for n in g_a:
        stem = n.startswith("backbone.conv1") or n.startswith("backbone.bn1")
        d = rel_l2(g_t[n], g_a[n])
        worst["stem" if stem else "stages"] = max(worst.get("stem" if stem else "stages", 0.0), d)
        assert d <= max(4 * noise, 5e-3 if stem else 1e-5), (n, d, noise)
    print("traced vs production: %s" % {k: "%.2e" % v for k, v in worst.items()})
