"""Which Python lines issue the small fill / copy launches of a training step?  (torch.profiler with stacks.)
usage (GPU box, repo root): python scripts/find_fills.py [18|50] [eval]"""
import collections
import os
import sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "video-classification_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "video-classification_amd", "lib"))


def main():
    from core.config import config
    from core.loss import Loss
    from models.model import create_model
    from utils.utils import create_optimizer
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 18
    dev = torch.device("cuda", 0)
    cfg = config.clone()
    cfg.MODEL.RESNET_TYPE = n
    cfg.MODEL.CLFDIM = 101 if n == 18 else 400
    cfg.MODEL.INIT_WEIGHTS = False
    cfg.TRAIN.OPTIMIZER = "sgd"
    model = create_model(cfg, is_train=True).to(dev)
    criterion = Loss(cfg).to(dev)
    optimizer = create_optimizer(cfg, model)
    x = torch.randn(16, 3, 16, 112, 112).to(torch.bfloat16).to(dev)
    y = torch.randint(0, cfg.MODEL.CLFDIM, (16,)).to(dev)
    model.train()

    def step():
        _, score, _ = model(x)
        loss = criterion(score, y, None, None)
        optimizer.zero_grad()
        loss.backward()
        optimizer.step()

    if len(sys.argv) > 2 and sys.argv[2] == "eval":
        model.eval()

        def step():  # noqa: F811
            with torch.no_grad():
                model(x)

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    import traceback
    cnt = collections.Counter()

    def wrap(owner, name):
        orig = getattr(owner, name)

        def f(*a, **k):
            fr = [("%s:%d" % (os.path.basename(t.filename), t.lineno)) for t in traceback.extract_stack()[:-1]
                  if "/torch/" not in t.filename and "find_fills" not in t.filename][-3:]
            cnt[(name, " <- ".join(reversed(fr)))] += 1
            return orig(*a, **k)
        setattr(owner, name, f)

    for owner, name in ((torch.Tensor, "zero_"), (torch.Tensor, "fill_"), (torch.Tensor, "copy_"), (torch, "zeros"),
                        (torch, "zeros_like"), (torch, "full"), (torch, "ones"), (torch, "tensor"), (torch.Tensor, "to"),
                        (torch.Tensor, "cuda"), (torch.Tensor, "contiguous"), (torch.Tensor, "clone"), (torch.Tensor, "float"),
                        (torch, "as_tensor"), (torch, "empty"), (torch, "empty_like")):
        wrap(owner, name)
    step()
    torch.cuda.synchronize()
    for (name, where), k in cnt.most_common():
        print(k, name, where)


main()
