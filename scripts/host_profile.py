#!/usr/bin/env python3
"""cProfile of the host side of training steps (no synchronisation inside the profiled region)."""
import cProfile, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "video-classification_amd"))
import _init_paths  # noqa
import torch
from core.config import config
from core.loss import Loss
from models.model import create_model
from utils.utils import create_optimizer
cfg = config.clone(); cfg.MODEL.RESNET_TYPE = 18; cfg.MODEL.CLFDIM = 101; cfg.MODEL.INIT_WEIGHTS = False
cfg.TRAIN.OPTIMIZER = "sgd"; cfg.TRAIN.LR = 0.001
model = create_model(cfg, is_train=True).cuda(); crit = Loss(cfg).cuda(); opt = create_optimizer(cfg, model)
x = torch.randn(64, 3, 16, 112, 112).to(torch.bfloat16).cuda(); y = torch.randint(0, 101, (64,)).cuda()
model.train()
def step():
    _, s, _ = model(x); loss = crit(s, y, None, None); opt.zero_grad(); loss.backward(); opt.step()
for _ in range(5): step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(10): step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
