"""Recorded steps (vc3d/replay.py, csrc/replay.hip: one C call per forward and per backward instead of ~200 calls from
Python) against the eager path.  In deterministic mode both issue the same launches on the same data, so everything
must be bit-identical: losses, gradients, parameters, BatchNorm buffers."""
import pytest
import torch
import torch.nn.functional as F

from helpers import make_config

pytestmark = pytest.mark.gpu


def _train(depth, classes, shapes, replay, opt_name="sgd", double_forward_at=None, single_stream_at=None):
    import vc3d.engine as eng_mod
    from core.loss import Loss
    from models.model import create_model
    from utils.utils import create_optimizer
    keep, eng_mod.REPLAY = eng_mod.REPLAY, replay
    try:
        cfg = make_config(depth, classes, OPTIMIZER=opt_name, LR=0.01)
        torch.manual_seed(3)
        model = create_model(cfg, is_train=True).cuda().train()
        crit, opt = Loss(cfg).cuda(), create_optimizer(cfg, model)
        g = torch.Generator().manual_seed(5)
        losses, used = [], []
        for i, shape in enumerate(shapes):
            x = torch.randn(*shape, generator=g).to(torch.bfloat16 if (i // 4) % 2 == 0 else torch.float32).cuda()  # a new tensor every step
            y = torch.randint(0, classes, (shape[0],), generator=g).cuda()
            # (bench.py's timing step: everything on one stream, weight gradients and downsample branches in line)
            if single_stream_at == i:
                from vc3d import ops
                ops.TIMER, eng_mod.SIDE_WGRAD = ops.KernelTimer(), False
            if double_forward_at == i:  # a second forward while the first one's activations wait for their backward
                _, s2, _ = model(x)
                l2 = crit(s2, y, None, None)
                _, score, _ = model(x.clone())
                loss = crit(score, y, None, None) + l2
            else:
                _, score, _ = model(x)
                loss = crit(score, y, None, None)
            opt.zero_grad()
            loss.backward()
            opt.step()
            losses.append(loss.detach().clone())
            if single_stream_at == i:
                torch.cuda.synchronize()
                assert len(ops.TIMER.summary()) > 10
                ops.TIMER, eng_mod.SIDE_WGRAD = None, True
            t = model.backbone.engine._tape
            used.append(None if t is None else (t["fwd"].n, None if t["bwd"] is None else t["bwd"].n, t["serial"]))
        torch.cuda.synchronize()
        eng = model.backbone.engine
        bufs = torch.cat([b.detach().double().flatten() for b in model.buffers()])
        nbt = [int(m.num_batches_tracked) for m in model.modules() if isinstance(m, torch.nn.BatchNorm3d)]
        return torch.stack(losses), eng.flat_grad.clone(), eng.flat_param.clone(), bufs, used, nbt
    finally:
        eng_mod.REPLAY = keep
        eng_mod.SIDE_WGRAD = True
        from vc3d import ops as _ops
        _ops.TIMER = None


def _same(a, b):
    for u, v, name in zip(a[:4], b[:4], ("losses", "gradients", "parameters", "buffers")):
        assert torch.equal(u, v), "%s differ: max |d| %g" % (name, float((u.double() - v.double()).abs().max()))


@pytest.mark.parametrize("depth,classes,opt_name", [(18, 11, "sgd"), (50, 7, "adam")])
def test_replayed_steps_equal_eager_steps_bit_for_bit(depth, classes, opt_name, det_mode):
    shapes = [(4, 3, 8, 64, 64)] * 8
    eager = _train(depth, classes, shapes, False, opt_name)
    rep = _train(depth, classes, shapes, True, opt_name)
    _same(eager, rep)
    assert eager[4][-1] is None
    used = rep[4]
    # step 0 runs eagerly (the fused optimizer has not built the packed weights yet), step 1 records, steps 2-3 replay;
    # then the clips change from bf16 to fp32: step 4 records again, steps 5-7 replay
    assert [u and u[2] for u in used] == [None, 0, 1, 2, 0, 1, 2, 3], used
    assert used[-1][0] > 40 and used[-1][1] is not None and used[-1][1] > 60, used
    print("commands per forward / backward list:", used[-1][:2])


def test_replay_with_one_input_format_replays_and_counts_batches(det_mode):
    import vc3d.engine as eng_mod
    from core.loss import Loss
    from models.model import create_model
    from utils.utils import create_optimizer
    cfg = make_config(18, 5, OPTIMIZER="sgd", LR=0.01)
    torch.manual_seed(0)
    model = create_model(cfg, is_train=True).cuda().train()
    crit, opt = Loss(cfg).cuda(), create_optimizer(cfg, model)
    serials = []
    for i in range(6):
        x = torch.randn(4, 3, 4, 32, 32).cuda()
        y = torch.randint(0, 5, (4,)).cuda()
        loss = crit(model(x)[1], y, None, None)
        opt.zero_grad()
        loss.backward()
        opt.step()
        t = model.backbone.engine._tape
        serials.append(None if t is None else t["serial"])
    torch.cuda.synchronize()
    assert serials == [None, 0, 1, 2, 3, 4], serials
    assert int(model.backbone.bn1.num_batches_tracked) == 6 and int(model.backbone.layer4[1].bn2.num_batches_tracked) == 6
    # an evaluation pass in between, a different batch size, and back: re-recorded, no stale state
    model.eval()
    with torch.no_grad():
        model(torch.randn(2, 3, 4, 32, 32).cuda())
    model.train()
    for n in (2, 4, 4):
        loss = crit(model(torch.randn(n, 3, 4, 32, 32).cuda())[1], torch.randint(0, 5, (n,)).cuda(), None, None)
        opt.zero_grad()
        loss.backward()
        opt.step()
    torch.cuda.synchronize()
    assert bool(torch.isfinite(loss)) and int(model.backbone.bn1.num_batches_tracked) == 9
    assert model.backbone.engine._tape["serial"] == 1


def test_a_single_stream_step_between_replays_leaves_no_stale_state(det_mode):
    """bench.py times every launch in one warm-up step with the side streams off; that step meets the BatchNorm layers
    in another order (downsample branches in line), which once replaced the table of step counters that the recorded
    forward names by address -- the next replay then read freed memory.  Steps before, at and after such a step must
    equal the all-eager run with the same schedule, and every counter must have advanced once per step."""
    shapes = [(4, 3, 4, 32, 32)] * 8
    eager = _train(18, 6, shapes, False, single_stream_at=2)
    rep = _train(18, 6, shapes, True, single_stream_at=2)
    _same(eager, rep)
    # step 1 records, step 2 is the eager single-stream step (the recording stays), step 3 REPLAYS the recording of step 1
    assert [u and u[2] for u in rep[4]] == [None, 0, 0, 1, 0, 1, 2, 3], rep[4]
    assert len(rep[5]) == 20 and set(rep[5]) == {8} and eager[5] == rep[5], rep[5]


def test_second_forward_before_backward_runs_eagerly_and_matches(det_mode):
    shapes = [(4, 3, 4, 32, 32)] * 5
    eager = _train(18, 6, shapes, False, double_forward_at=3)
    rep = _train(18, 6, shapes, True, double_forward_at=3)
    _same(eager, rep)


def test_backward_of_an_overwritten_recorded_step_is_refused(det_mode):
    from core.loss import Loss
    from models.model import create_model
    from utils.utils import create_optimizer
    from vc3d.capi import Vc3dError
    cfg = make_config(18, 5, OPTIMIZER="sgd", LR=0.01)
    model = create_model(cfg, is_train=True).cuda().train()
    crit, opt = Loss(cfg).cuda(), create_optimizer(cfg, model)
    x, y = torch.randn(4, 3, 4, 32, 32).cuda(), torch.randint(0, 5, (4,)).cuda()
    for _ in range(3):
        loss = crit(model(x)[1], y, None, None)
        opt.zero_grad()
        loss.backward()
        opt.step()
    assert model.backbone.engine._tape is not None and model.backbone.engine._tape["bwd"] is not None
    stale = crit(model(x)[1], y, None, None)  # recorded step, backward pending
    # the pending step keeps its activations: a second forward runs eagerly and both backward passes work
    other = crit(model(x)[1], y, None, None)
    opt.zero_grad()
    other.backward()
    stale.backward()
    torch.cuda.synchronize()
    assert bool(torch.isfinite(model.backbone.engine.flat_grad).all())
