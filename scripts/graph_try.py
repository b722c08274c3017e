"""Experiment: whole training step captured in a HIP graph vs eager (scripts/, not a test)."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "video-classification_amd"), os.path.join(ROOT, "video-classification_amd", "lib")):
    sys.path.insert(0, p)
import torch
from core.config import config
from core.loss import Loss
from models.model import create_model
from utils.utils import create_optimizer

depth = int(sys.argv[1]) if len(sys.argv) > 1 else 18
cfg = config.clone()
cfg.MODEL.RESNET_TYPE = depth
cfg.MODEL.CLFDIM = 101 if depth == 18 else 400
cfg.MODEL.INIT_WEIGHTS = False
cfg.TRAIN.OPTIMIZER = "sgd"
cfg.TRAIN.LR = 0.001
torch.manual_seed(0)
dev = torch.device("cuda", 0)
model = create_model(cfg, True).to(dev).train()
crit = Loss(cfg).to(dev)
opt = create_optimizer(cfg, model)
g = torch.Generator().manual_seed(1234)
x = torch.randn(64, 3, 16, 112, 112, generator=g).to(torch.bfloat16).to(dev)
y = torch.randint(0, cfg.MODEL.CLFDIM, (64,), generator=g).to(dev)
loss_out = torch.zeros((), device=dev)

def step():
    _, score, _ = model(x)
    loss = crit(score, y, None, None)
    opt.zero_grad()
    loss.backward()
    opt.step()
    loss_out.copy_(loss.detach())

def timeit(fn, n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

for _ in range(5): step()
print("eager %.3f ms/step" % timeit(step, 30), flush=True)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    step()
torch.cuda.synchronize()
print("captured", flush=True)
for _ in range(3): gr.replay()
print("graph %.3f ms/step  loss %.4f" % (timeit(gr.replay, 30), float(loss_out)), flush=True)
print("eager again %.3f ms/step" % timeit(step, 30), flush=True)
