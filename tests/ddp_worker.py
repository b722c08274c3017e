"""Worker of tests/test_ddp_gpu.py (launched by torch.distributed.run).

  identity : one rank, backend nccl (RCCL), VC_DDP_FORCE_COMM=1.  The same training step plain (twice: the
             rerun noise), wrapped in DataParallel3D (an AVG all-reduce over one rank is the identity) and wrapped
             with an INJECTED collective that doubles each bucket on the communication stream: the final gradients
             must be 2x the plain ones, so a bucket launched before its last gradient was final -- which the identity
             cannot show -- is off by O(1) in that parameter.  Prints the deviation per parameter.
  shards   : two ranks (backend gloo, both on cuda:0), each with its own half of the batch, per-rank BatchNorm
             statistics, gradients averaged by the bucket reducer -- the data-parallel semantics of the reference's
             torch.nn.DataParallel (train.py:42-43,65).  Rank 0 then repeats the step in one process as two sequential
             micro-shards with per-shard BatchNorm and averaged gradients, and compares gradients and parameters.
"""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-classification_amd"))
import _init_paths  # noqa: F401,E402
from core.config import config  # noqa: E402
from core.loss import Loss  # noqa: E402
from models.model import create_model  # noqa: E402
from utils.utils import create_optimizer  # noqa: E402
from vc3d.ddp import DataParallel3D  # noqa: E402


def build(lr=0.01, lstm=False):
    cfg = config.clone()
    cfg.MODEL.RESNET_TYPE = 18
    cfg.MODEL.CLFDIM = 11
    cfg.MODEL.INIT_WEIGHTS = False
    if lstm:
        cfg.MODEL.USE_LSTM, cfg.MODEL.LSTM_OUTDIM = True, 128
    cfg.TRAIN.OPTIMIZER = "sgd"
    cfg.TRAIN.LR = lr
    torch.manual_seed(3)
    model = create_model(cfg, is_train=True).cuda()
    return cfg, model, Loss(cfg).cuda(), create_optimizer(cfg, model)


def batch(n=8):
    g = torch.Generator().manual_seed(5)
    x = torch.randn(n, 3, 8, 64, 64, generator=g).to(torch.bfloat16).cuda()
    y = torch.randint(0, 11, (n,), generator=g).cuda()
    return x, y


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


def per_param(eng, model, a, b, top=4):
    """largest relative deviations of flat buffer `a` from `b`, per parameter (name, rel-L2)."""
    names = {id(p): n for n, p in model.named_parameters()}
    out = []
    for p, off, n in eng.grad_ranges:
        out.append((rel(a[off:off + n], b[off:off + n]), names.get(id(p), "?")))
    out.sort(reverse=True)
    return out[:top], out[len(out) // 2]


def run_identity(wrap, inject=False):
    cfg, model, crit, opt = build()
    net = DataParallel3D(model, bucket_mb=4, tail_mb=0.25, grad_dtype="fp32") if wrap else model
    x, y = batch()
    net.train()
    _, score, _ = net(x)
    loss = crit(score, y, None, None)
    opt.zero_grad()
    if inject:
        net._ensure().collective = lambda seg: seg.mul_(2)
    loss.backward()
    opt.step()
    torch.cuda.synchronize()
    eng = model.backbone.engine
    nb = len(net.reducer.buckets) if wrap else 0
    return eng, model, eng.flat_grad.clone(), eng.flat_param.clone(), float(loss), nb


def main_identity():
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    eng, model, g0, p0, l0, _ = run_identity(False)
    _, _, g2, p2, l2, _ = run_identity(False)
    _, _, g1, p1, l1, nb = run_identity(True)
    _, _, g3, _, _, _ = run_identity(True, inject=True)
    for tag, a, b in (("rerun", g2, g0), ("wrapped", g1, g0), ("injected_x2", g3, 2 * g0)):
        worst, med = per_param(eng, model, a, b)
        print("DDP_PARAMS %s worst %s median %.3g" % (tag, ["%s=%.3g" % (n, v) for v, n in worst], med[0]), flush=True)
    print("DDP_RESULT grad_rel=%g param_rel=%g rerun_grad_rel=%g rerun_param_rel=%g inject_rel=%g loss=%g/%g buckets=%d "
          "finite=%d" % (rel(g1, g0), rel(p1, p0), rel(g2, g0), rel(p2, p0), rel(g3, 2 * g0), l0, l1, nb,
                         int(torch.isfinite(g1).all())), flush=True)
    dist.destroy_process_group()


def main_shards():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)  # both ranks share the one GPU of the box; gradients travel through gloo
    dist.init_process_group("gloo", rank=rank, world_size=world)
    x, y = batch(16)  # 8 clips per rank: the size at which a plain step repeats itself (see DDP_CHECK rerun below)
    per = x.shape[0] // world
    res = {}
    for dtype in ("fp32", "bf16"):
        cfg, model, crit, opt = build()
        net = DataParallel3D(model, bucket_mb=4, tail_mb=0.25, grad_dtype=dtype)
        net.train()
        _, score, _ = net(x[rank * per:(rank + 1) * per])
        loss = crit(score, y[rank * per:(rank + 1) * per], None, None)
        opt.zero_grad()
        loss.backward()
        opt.step()
        torch.cuda.synchronize()
        eng = model.backbone.engine
        res[dtype] = (eng.flat_grad.clone(), eng.flat_param.clone(), len(net.reducer.buckets))
        # every rank must hold the same averaged gradients
        mine = eng.flat_grad.cpu()
        both = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(both, mine)
        assert all(torch.equal(both[0], t) for t in both), "ranks disagree on the averaged gradients"
    if rank == 0:
        cfg, model, crit, opt = build()
        model.train()
        opt.zero_grad()
        eng = model.backbone.engine
        for r in range(world):  # sequential micro-shards: per-shard BatchNorm statistics, gradients accumulate
            _, score, _ = model(x[r * per:(r + 1) * per])
            crit(score, y[r * per:(r + 1) * per], None, None).backward()
            for cb in eng.all_convbn():  # a rank's first step takes its BatchNorm sums relative to a zero shift: the
                cb.shift.zero_()         # emulation must too, or the sums differ in their last bits (and then see above)
        eng.flat_grad.div_(world)
        opt.step()
        torch.cuda.synchronize()
        # third form: each micro-shard in a fresh model, gradients averaged on the host side of the comparison
        singles = []
        for r in range(world):
            _, m2, crit2, opt2 = build()
            m2.train()
            opt2.zero_grad()
            _, score, _ = m2(x[r * per:(r + 1) * per])
            crit2(score, y[r * per:(r + 1) * per], None, None).backward()
            torch.cuda.synchronize()
            singles.append(m2.backbone.engine.flat_grad.clone())
        avg = sum(singles) / world
        _, m3, crit3, opt3 = build()  # shard 0 once more: the run-to-run noise of this configuration
        m3.train()
        opt3.zero_grad()
        crit3(m3(x[:per])[1], y[:per], None, None).backward()
        torch.cuda.synchronize()
        print("DDP_CHECK rerun=%g accumulate_vs_separate=%g ddp_fp32_vs_separate=%g shard0_vs_shard1=%g" % (
            rel(m3.backbone.engine.flat_grad, singles[0]), rel(eng.flat_grad, avg), rel(res["fp32"][0], avg),
            rel(singles[0], singles[1])), flush=True)
        for dtype, (g, p, nb) in res.items():
            worst, med = per_param(eng, model, g, eng.flat_grad)
            print("DDP_PARAMS shards_%s worst %s median %.3g" % (dtype, ["%s=%.3g" % (n, v) for v, n in worst], med[0]),
                  flush=True)
            print("DDP_SHARDS dtype=%s grad_rel=%g param_rel=%g buckets=%d" % (dtype, rel(g, eng.flat_grad),
                                                                                rel(p, eng.flat_param), nb), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def main_lstm():
    """The recurrent head (MODEL.USE_LSTM) under data parallelism: its parameters live outside the engine's flat
    buffer; after backward every rank must hold the SAME gradients for them, equal to the average of the two shards'
    own gradients (computed by each rank before the wrapped run, exchanged through gloo)."""
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    x, y = batch(16)
    per = x.shape[0] // world
    xs, ys = x[rank * per:(rank + 1) * per], y[rank * per:(rank + 1) * per]

    def step(wrap):
        cfg, model, crit, opt = build(lstm=True)
        net = DataParallel3D(model, bucket_mb=4, tail_mb=0.25) if wrap else model
        assert not wrap or net.grad_dtype == "fp32", "fp32 on the wire is the default"
        net.train()
        model.lstm.reset(per)
        action, score, value = net(xs)
        g = torch.Generator().manual_seed(9 + rank)
        log_prob = torch.log(action.clamp_min(1e-6))
        adv = torch.randn(per, 1, generator=g).cuda() + value
        loss = crit(score, ys, log_prob, adv)
        opt.zero_grad()
        loss.backward()
        torch.cuda.synchronize()
        extras = net.extra_params() if wrap else DataParallel3D.extra_params(type("W", (), {"module": model})())
        names = {id(p): n for n, p in model.named_parameters()}
        return {names[id(p)]: p.grad.detach().float().cpu().clone() for p in extras}, model.backbone.engine.flat_grad.cpu().clone()

    own, own_flat = step(False)
    got, got_flat = step(True)
    assert {"lstm.lstm.weight_ih", "lstm.lstm.weight_hh", "act_head.0.weight", "clf_haed.weight", "crit_head.bias"} <= set(got)
    worst_avg, worst_ranks, self_dev = 0.0, 0.0, 0.0
    for name in sorted(own):
        parts = [torch.zeros_like(own[name]) for _ in range(world)]
        dist.all_gather(parts, own[name])
        avg = sum(parts) / world
        theirs = [torch.zeros_like(got[name]) for _ in range(world)]
        dist.all_gather(theirs, got[name])
        assert all(torch.equal(theirs[0], t) for t in theirs), "ranks disagree on %s" % name
        worst_avg = max(worst_avg, rel(got[name], avg))
        self_dev = max(self_dev, rel(parts[0], parts[1]))
    parts = [torch.zeros_like(own_flat) for _ in range(world)]
    dist.all_gather(parts, own_flat)
    flat_rel = rel(got_flat, sum(parts) / world)
    if rank == 0:
        print("DDP_LSTM extras=%d worst_vs_average=%g backbone_vs_average=%g shard0_vs_shard1=%g" % (
            len(got), worst_avg, flat_rel, self_dev), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def main_steps():
    """Several training steps through the wrapper with the step RECORDED (vc3d/replay.py): the data-parallel backward
    replays in segments that end where a gradient became final, and the wrapper launches its buckets between them.
    One rank, RCCL, the injected doubling collective (a bucket launched before its last gradient was final misses that
    gradient's doubling): the replayed run must end with the same parameters as the eager run, bit for bit, and both
    with exactly what the doubling implies; the AVG collective likewise against the plain, unwrapped run."""
    import vc3d.engine as eng_mod
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))

    def run(wrap, inject, replay, steps=6, arm=False):
        eng_mod.REPLAY = replay
        cfg, model, crit, opt = build(lr=0.02)
        net = DataParallel3D(model, bucket_mb=4, tail_mb=0.25, grad_dtype="fp32") if wrap else model
        net.train()
        g = torch.Generator().manual_seed(11)
        info = None
        for i in range(steps):
            x = torch.randn(8, 3, 8, 64, 64, generator=g).to(torch.bfloat16).cuda()
            y = torch.randint(0, 11, (8,), generator=g).cuda()
            _, score, _ = net(x)
            loss = crit(score, y, None, None)
            opt.zero_grad()
            if inject and wrap:
                net._ensure().collective = lambda seg: seg.mul_(2)
            if arm and i > 0:  # the optimizer steps each bucket right behind its all-reduce
                assert opt.arm_backward(data_parallel=True)
            loss.backward()
            if arm and i > 0:
                assert sum(b - a for a, b in opt._applied) >= 24 << 20, opt._applied  # (layer4 at least: 25 of 33 M elements)
            opt.step()
            t = model.backbone.engine._tape
            if t is not None and t["bwd"] is not None:
                info = (t["serial"], t["bwd"].n, len(t["bwd"].marks), sum(1 for m in t["bwd"].marks if m[2]))
        torch.cuda.synchronize()
        eng = model.backbone.engine
        return eng.flat_param.clone(), eng.flat_grad.clone(), info

    plain_p, plain_g, plain_info = run(False, False, True)
    avg_e_p, avg_e_g, _ = run(True, False, False)
    avg_r_p, avg_r_g, avg_info = run(True, False, True)
    inj_e_p, inj_e_g, _ = run(True, True, False)
    inj_r_p, inj_r_g, inj_info = run(True, True, True)
    # the optimizer stepping bucket by bucket inside the backward (optim.arm_backward under the wrapper): same result
    arm_r_p, arm_r_g, _ = run(True, False, True, arm=True)
    arm_i_p, arm_i_g, _ = run(True, True, False, arm=True)
    eng_mod.REPLAY = True
    print("DDP_STEPS avg_replay_vs_eager=%g avg_vs_plain=%g inject_replay_vs_eager=%g inject_vs_plain_grad=%g "
          "plain=%s wrapped=%s injected=%s armed_vs_not=%g" % (
              rel(avg_r_p, avg_e_p) + rel(avg_r_g, avg_e_g), rel(avg_r_p, plain_p), rel(inj_r_p, inj_e_p) + rel(inj_r_g, inj_e_g),
              rel(inj_r_p, plain_p), plain_info, avg_info, inj_info,
              rel(arm_r_p, avg_r_p) + rel(arm_r_g, avg_r_g) + rel(arm_i_p, inj_e_p) + rel(arm_i_g, inj_e_g)), flush=True)
    dist.destroy_process_group()


def main_steps2():
    """Two ranks (gloo, both on cuda:0, buckets staged through the host), five training steps each on its own shard,
    with the step recorded and with it eager: the ranks must end with identical parameters in both runs (the buckets
    of the segmented replay line up across ranks), and the replayed run must equal the eager one bit for bit."""
    import vc3d.engine as eng_mod
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)

    def run(replay, steps=5, arm=False):
        eng_mod.REPLAY = replay
        cfg, model, crit, opt = build(lr=0.02)
        net = DataParallel3D(model, bucket_mb=4, tail_mb=0.25, grad_dtype="fp32")
        net.train()
        g = torch.Generator().manual_seed(21 + rank)
        info = None
        for i in range(steps):
            x = torch.randn(8, 3, 8, 64, 64, generator=g).to(torch.bfloat16).cuda()
            y = torch.randint(0, 11, (8,), generator=g).cuda()
            loss = crit(net(x)[1], y, None, None)
            opt.zero_grad()
            if arm and i > 0:
                assert opt.arm_backward(data_parallel=True)
            loss.backward()
            opt.step()
            t = model.backbone.engine._tape
            if t is not None and t["bwd"] is not None:
                info = (t["serial"], len(t["bwd"].marks))
        torch.cuda.synchronize()
        p = model.backbone.engine.flat_param.cpu()
        both = [torch.zeros_like(p) for _ in range(world)]
        dist.all_gather(both, p)
        return p, all(torch.equal(both[0], t) for t in both), info

    pe, same_e, _ = run(False)
    pr, same_r, info = run(True)
    pa, same_a, _ = run(True, arm=True)  # the optimizer stepping each bucket behind its all-reduce
    eng_mod.REPLAY = True
    if rank == 0:
        print("DDP_STEPS2 ranks_agree_eager=%d ranks_agree_replay=%d replay_vs_eager=%g recorded=%s armed_agree=%d armed_vs_eager=%g" % (
            int(same_e), int(same_r), rel(pr, pe), info, int(same_a), rel(pa, pe)), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    import faulthandler
    faulthandler.dump_traceback_later(100, exit=True)  # a stuck collective names itself instead of hanging the box
    if len(sys.argv) > 1 and sys.argv[1] == "shards":
        main_shards()
    elif len(sys.argv) > 1 and sys.argv[1] == "lstm":
        main_lstm()
    elif len(sys.argv) > 1 and sys.argv[1] == "steps":
        main_steps()
    elif len(sys.argv) > 1 and sys.argv[1] == "steps2":
        main_steps2()
    else:
        main_identity()
