"""the reference's recurrent head -- nn.LSTMCell with the state in the module (lstm.py:12-27), the
action / class / critic heads (model.py:18-25,35-41) and the actor / critic loss terms (loss.py:25-27) -- on the vc3d
kernels.  Oracle: torch.nn.LSTMCell, fp32, CPU (the reference's own operator).  Tolerances: the gates cross the bf16
MFMA GEMM (bf16 operands and a bf16 gate tensor), so outputs agree to rel-L2 <= 1e-2 and gradients to <= 2e-2."""
import pytest
import torch
import torch.nn.functional as F

from helpers import make_config, rel_l2

pytestmark = pytest.mark.gpu


def rbf(t):
    return t.to(torch.bfloat16).float()


@pytest.mark.parametrize("N,fin,H", [(16, 512, 256), (64, 2048, 2048), (5, 64, 64)])
def test_lstm_cell_matches_torch_lstmcell(N, fin, H):
    from vc3d.lstm import _PackedLSTMWeights, lstm_cell
    g = torch.Generator().manual_seed(N + fin)
    ref = torch.nn.LSTMCell(fin, H)
    with torch.no_grad():
        for p in ref.parameters():
            p.copy_(rbf(p))
    x = rbf(torch.randn(N, fin, generator=g)).requires_grad_()
    h0 = rbf(0.5 * torch.randn(N, H, generator=g)).requires_grad_()
    c0 = (0.5 * torch.randn(N, H, generator=g)).requires_grad_()
    gh, gc = torch.randn(N, H, generator=g), torch.randn(N, H, generator=g)
    h1, c1 = ref(x, (h0, c0))
    (h1 * gh).sum().add((c1 * gc).sum()).backward()
    dev = [t.detach().cuda().requires_grad_() for t in (x, h0, c0, ref.weight_ih, ref.weight_hh, ref.bias_ih, ref.bias_hh)]
    h, c = lstm_cell(*dev, _PackedLSTMWeights())
    (h * gh.cuda()).sum().add((c * gc.cuda()).sum()).backward()
    torch.cuda.synchronize()
    errs = {"h": rel_l2(h, h1), "c": rel_l2(c, c1)}
    for name, a, b in zip(("dx", "dh0", "dc0", "dW_ih", "dW_hh", "db_ih", "db_hh"), dev,
                          (x, h0, c0, ref.weight_ih, ref.weight_hh, ref.bias_ih, ref.bias_hh)):
        errs[name] = rel_l2(a.grad, b.grad)
    print({k: "%.2e" % v for k, v in errs.items()})
    assert errs["h"] <= 1e-2 and errs["c"] <= 1e-2
    assert all(v <= 2e-2 for k, v in errs.items() if k not in ("h", "c")), errs


def test_recurrent_head_model_trains_with_actor_critic_terms(monkeypatch):
    """create_model with MODEL.USE_LSTM: the reference's forward contract (model.py:27-41: action N x 1 in (0, 1), score
    N x classes, value N x 1), its state_dict keys, the stateful h / c across inner steps, and the reference's step loop
    (function.py:58-85) with rollout's log_prob / advantages feeding Loss: the loss falls on a fixed batch."""
    from core.config import config as gcfg
    from core.function import train
    from core.loss import Loss
    from models.model import create_model
    from utils.utils import create_optimizer, rollout
    cfg = make_config(18, 6, OPTIMIZER="adam", LR=1e-3, TRAIN_STEP=6, PRINT_EVERY=100)
    cfg.MODEL.USE_LSTM, cfg.MODEL.LSTM_OUTDIM = True, 256
    torch.manual_seed(0)
    model = create_model(cfg, True).cuda()
    keys = set(model.state_dict())
    assert {"lstm.lstm.weight_ih", "lstm.lstm.weight_hh", "lstm.lstm.bias_ih", "lstm.lstm.bias_hh", "act_head.0.weight",
            "act_head.0.bias", "clf_haed.weight", "crit_head.weight", "backbone.conv1.weight"} <= keys
    assert model.lstm.lstm.weight_ih.shape == (4 * 256, 512)  # input width = the backbone's feature width
    opt = create_optimizer(cfg, model)
    assert type(opt) is torch.optim.Adam
    g = torch.Generator().manual_seed(4)
    x = torch.randn(4, 3, 4, 32, 32, generator=g)
    y = torch.randint(0, 6, (4,), generator=g)
    model.train()
    model.lstm.reset(4)
    action, score, value = model(x.cuda())
    assert action.shape == (4, 1) and value.shape == (4, 1) and score.shape == (4, 6)
    assert bool(((action > 0) & (action < 1)).all())
    h_after_1 = model.lstm.h.clone()
    monkeypatch.setattr(gcfg.TRAIN, "ROLLOUT_STEP", 1, raising=False)
    log_prob, adv = rollout(x.cuda(), model, action, value)
    assert log_prob.shape == (4, 1) and adv.shape == (4, 1) and bool((log_prob < 0).all())
    assert torch.equal(model.lstm.h, h_after_1), "rollout must restore the LSTM state"
    crit = Loss(cfg)
    loss = crit(score, y.cuda(), log_prob, adv)
    want = F.cross_entropy(score, y.cuda()) - cfg.TRAIN.R_ACT * (adv.detach() * log_prob).mean() + cfg.TRAIN.R_CRT * (adv ** 2).mean()
    assert abs(float(loss) - float(want)) <= 1e-6 * abs(float(want)) + 1e-7
    loss.backward()
    for n, p in model.named_parameters():
        assert p.grad is not None and bool(torch.isfinite(p.grad).all()), n
    assert float(model.crit_head.weight.grad.abs().sum()) > 0 and float(model.act_head[0].weight.grad.abs().sum()) > 0
    opt.zero_grad()
    # the reference's loop: TRAIN_STEP inner steps per batch, state carried from step to step
    losses = []
    for ep in range(3):
        model.lstm.reset(4)
        with torch.no_grad():
            model.eval()
            losses.append(float(F.cross_entropy(model(x.cuda())[1], y.cuda())))
        model.lstm.reset(4)
        train(cfg, [(x, y)], model, crit, opt, ep)
    print("cross-entropy before each epoch of 6 inner steps:", ["%.3f" % v for v in losses])
    assert losses[-1] < losses[0]
