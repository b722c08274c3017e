import os, sys
sys.path.insert(0, "/root/repo/video-classification_amd")
import _init_paths, torch
from core.config import config
from core.loss import Loss
from models.model import create_model
from utils.utils import create_optimizer
cfg = config.clone(); cfg.MODEL.RESNET_TYPE = 18; cfg.MODEL.CLFDIM = 101; cfg.MODEL.INIT_WEIGHTS = False
cfg.TRAIN.OPTIMIZER = "sgd"; cfg.TRAIN.LR = 0.001
model = create_model(cfg, is_train=True).cuda(); crit = Loss(cfg).cuda(); opt = create_optimizer(cfg, model)
x = torch.randn(16, 3, 16, 112, 112).to(torch.bfloat16).cuda(); y = torch.randint(0, 101, (16,)).cuda()
model.train()
def step():
    _, s, _ = model(x); loss = crit(s, y, None, None); opt.zero_grad(); loss.backward(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], record_shapes=True, with_stack=True) as prof:
    step()
torch.cuda.synchronize()
ka = prof.key_averages(group_by_input_shape=True, group_by_stack_n=4)
rows = [e for e in ka if e.key in ("aten::copy_", "aten::fill_", "aten::zero_", "aten::zeros", "aten::clone", "aten::contiguous", "aten::to", "aten::_to_copy")]
for e in sorted(rows, key=lambda e: -e.count):
    st = [s for s in e.stack if "site-packages" not in s and "dist-packages" not in s][:2]
    print(e.key, e.count, e.input_shapes[:2], " | ".join(s.split("/")[-1][:70] for s in st))
