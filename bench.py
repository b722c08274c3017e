#!/usr/bin/env python3
"""Headline benchmark: train clips/sec of the 3D-ResNet at 16x112x112 on N MI355X GPUs.

  python bench.py --gpus N --steps K --warmup W

N > 1: when the command is not already a rank of a torch.distributed.run launch (WORLD_SIZE unset), bench.py itself
starts N rank processes, one per GPU (video-classification_amd/launch.py: fresh children, before this process touches
the GPU), relays rank 0's JSON line and exits with the ranks' status -- the counterpart of the reference fanning out
over config.GPUS inside one command (/root/reference/train.py:42-43,65).  Under torch.distributed.run it is one rank.
`n_gpus` in the line is the world size the process group reports; a mismatch with --gpus is an error, not a warning.

A step = forward + cross-entropy + backward + optimizer step on one synthetic batch that is
already resident in HBM.  Rank 0 prints ONE JSON line (contract in the task statement) with
two extra objects:
  roofline     : the dominant kernel (largest share of conv time), timed live with HIP events
                 on the launch stream inside the timed region; achieved = algorithmic FLOPs
                 per launch / average launch duration, against the 2.5 PFLOP/s dense bf16 peak.
  cpu_baseline : the oracle (stock torch.nn fp32 3D-ResNet-18, oracle/r3d_torch.py) timed on
                 this box's host cores on a bounded sample (config 1: batch 8), rank 0, N=1 only.
"""
import os as _os

# more hardware queues than HIP's default of 4: backward stream + weight-gradient stream + bucket stream + RCCL
# (vc3d/ddp.py); must be set before the HIP runtime initialises
if int(_os.environ.get("GPU_MAX_HW_QUEUES", "4") or 4) < 8 and _os.environ.get("VC_KEEP_HW_QUEUES") != "1":
    _os.environ["GPU_MAX_HW_QUEUES"] = "8"

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "video-classification_amd")
for p in (ROOT, PKG, os.path.join(PKG, "lib")):
    if p not in sys.path:
        sys.path.insert(0, p)

import launch  # noqa: E402  (stdlib only: no torch, no HIP)


PEAK_BF16_TFLOPS = 2500.0  # MI355X dense bf16 MFMA (guide: ~2.5 PF dense)
PEAK_HBM_GBS = 8000.0
# fwd+bwd GFLOP per clip (contract figure, SURVEY.md 8(d) / BASELINE.md)
GFLOP_PER_CLIP = {(18, 16, 112): 43.25, (50, 16, 112): 53.81}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--model", type=int, default=None,
                    help="ResNet depth; default 18 at every N (BASELINE config 2, weak-scaled: one workload along the "
                         "1/2/4/8 curve, so that value(N) / (N * value(1)) is a scaling efficiency); 50 = config 3")
    ap.add_argument("--batch", type=int, default=64, help="clips per GPU")
    ap.add_argument("--clip-len", type=int, default=16)
    ap.add_argument("--crop", type=int, default=112)
    ap.add_argument("--classes", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=0, help="timed oracle steps of the cpu_baseline leg (0: as many as fit ~12 s)")
    ap.add_argument("--profile-kernels", action="store_true", help="print the per-kernel table to stderr")
    ap.add_argument("--eval", action="store_true", help="forward-only (valid.py path) throughput")
    ap.add_argument("--optimizer", default="sgd", choices=("sgd", "adam"), help="TRAIN.OPTIMIZER (reference default: adam)")
    ap.add_argument("--views", type=int, default=0,
                    help="with --eval: BASELINE config 5's protocol -- every video is VIEWS clips (10 clips x 3 crops = 30) "
                         "through valid.py's loader and validate(), logits averaged per video; --batch = clips per forward")
    ap.add_argument("--dry-launch", action="store_true",
                    help="start the rank processes, have each report its environment and stop before touching the GPU")
    return ap.parse_args()


def cpu_baseline(steps):
    """oracle fp32 fwd+bwd+SGD step, 8x3x16x112x112, on the host cores (kind: port)."""
    from oracle.r3d_torch import R3DOracle, init_weights_, synthetic_batch
    model = init_weights_(R3DOracle(18, 101), 0).train()
    opt = torch.optim.SGD(model.parameters(), lr=0.001, momentum=0.9, weight_decay=1e-4)
    x, y = synthetic_batch(8, 16, 112, 112, 101, 1234)

    def step():
        opt.zero_grad()
        F.cross_entropy(model(x), y).backward()
        opt.step()

    step()  # warm-up
    t0 = time.time()
    step()
    if steps <= 0:  # a bounded sample: about 12 s of CPU work
        steps = max(2, min(200, int(12.0 / max(time.time() - t0, 1e-3))))
    t0 = time.time()
    for _ in range(steps):
        step()
    dt = (time.time() - t0) / steps
    return {"value": 8.0 / dt, "unit": "clips/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%d timed steps (2 warm-up) of oracle R3D-18 fp32 fwd+bwd+SGD on 8x3x16x112x112, %s" %
                      (steps, _cpu_model())}


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def pmc_traffic(key):
    """HBM bytes per launch of the dominant kernel, from the committed rocprofv3 --pmc passes
    (profiles/pmc_traffic.json, made by scripts/pmc_check.sh + pmc_summary.py; bench.py cannot run the
    profiler on itself).  None when that launch (kernel AND geometry) has no measurement on file."""
    try:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_traffic.json")) as f:
            rec = json.load(f).get(key[0])
        if rec is None or rec.get("launch") != "x".join(str(v) for v in key[1:]):
            return None
        return rec["hbm_bytes"]
    except (OSError, ValueError, KeyError):
        return None


def pointwise_bytes(key):
    """algorithmic HBM bytes of a 1x1x1 conv launch: the activation read once and the result written once, bf16
    (bandwidth kernels are priced at (elements read + elements written) x 2 B); weights are noise."""
    kind, N, Ci, T, H, W, Co, kt, kh, kw, st, sh, sw = key
    To, Ho, Wo = (T - 1) // st + 1, (H - 1) // sh + 1, (W - 1) // sw + 1
    m_in = N * To * Ho * Wo if kind != "dgrad" else N * T * H * W  # a strided 1x1x1 forward only reads the rows it keeps
    return 2.0 * (m_in * Ci + N * To * Ho * Wo * Co) if kind != "wgrad" else 2.0 * N * To * Ho * Wo * (Ci + Co)


def launch_roofline(key, ms, flops):
    """roofline object of one conv launch: the 1x1x1 convs of the Bottleneck (K = 64..2048 per output row) are bound
    by HBM bandwidth, everything else by the MFMA pipe."""
    name = "conv3d %s %s" % (key[0], "x".join(str(v) for v in key[1:]))
    if len(key) == 13 and key[7] * key[8] * key[9] == 1:
        gbs = pointwise_bytes(key) / (ms * 1e-3) / 1e9
        return {"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                "frac": round(gbs / PEAK_HBM_GBS, 4), "kernel": name}
    ach = flops / (ms * 1e-3) / 1e12
    return {"bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
            "frac": round(ach / PEAK_BF16_TFLOPS, 4), "kernel": name}


def family_rooflines(table):
    """conv launches of one step grouped by the kernel that serves them; returns the family with the largest share."""
    fam = {}
    for key, (n, ms, fl) in table.items():
        if key[0] in ("stem_fwd", "stem_wgrad"):
            name = key[0] + "_kernel"
        else:
            kind, k, s = key[0], key[7] * key[8] * key[9], key[10] * key[11] * key[12]
            if k == 27 and s == 1:
                name = "wgrad9_kernel (3x3x3 stride-1 wgrad)" if kind == "wgrad" else "halo_conv_kernel (3x3x3 stride-1 fwd+dgrad)"
            elif k == 1:
                name = "1x1x1 convs (pw_kernel / igemm_kernel / wgrad_kernel, HBM-bound)"
            else:
                name = "strided convs (igemm_kernel / wgrad_kernel)"
        f = fam.setdefault(name, [0, 0.0, 0.0, 0.0])
        f[0] += n
        f[1] += n * ms
        f[2] += n * fl
        f[3] += n * (pointwise_bytes(key) if len(key) == 13 and key[7] * key[8] * key[9] == 1 else 0.0)
    if not fam:
        return None
    tot = sum(f[1] for f in fam.values())

    def entry(name, n, ms, fl, by):
        out = {"kernel": name, "launches_per_step": n, "ms_per_step": round(ms, 4), "share_of_conv_time": round(ms / tot, 4)}
        if by > 0:
            gbs = by / (ms * 1e-3) / 1e9
            out.update({"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                        "frac": round(gbs / PEAK_HBM_GBS, 4)})
        else:
            ach = fl / (ms * 1e-3) / 1e12
            out.update({"bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(ach / PEAK_BF16_TFLOPS, 4)})
        return out

    ranked = sorted(fam.items(), key=lambda kv: -kv[1][1])
    top = entry(ranked[0][0], *ranked[0][1])
    # every family, largest first: the headline `roofline` is the best of the big kernels, these are all of them
    top["families"] = [entry(name, *f) for name, f in ranked]
    return top


def bench_multi_view(args, cfg, net, criterion, dev, rank, world, ddp):
    """BASELINE config 5 as specified: every video is VIEWS = 10 clips x 3 crops views of 3 x 16 x 112 x 112, scored by
    lib/core/function.py:validate (reference function.py:98-156, valid.py:28-61): eval-mode BatchNorm folded into the
    conv epilogues, TEST.BATCH_SIZE clips per forward, the views' logits averaged per video, top-1 / top-5 counted on
    the device, one host read at the end.  A step = one loader batch of `vids` videos (vids x VIEWS clips, resident in
    HBM); value = clips (views) per second, videos per second beside it."""
    from core.function import validate
    cfg.TEST.BATCH_SIZE = args.batch
    vids = max(1, args.batch // 2)  # 32 videos x 30 views = 960 clips = 15 forwards of 64
    g = torch.Generator(device=dev).manual_seed(4321 + rank)
    base = torch.randn(vids, args.views, 3, args.clip_len, args.crop, args.crop, generator=g, device=dev,
                       dtype=torch.float32).to(torch.bfloat16)
    y = torch.randint(0, cfg.MODEL.CLFDIM, (vids,), generator=g, device=dev)

    class Resident:  # a loader whose batches already sit in HBM (the bench contract: inputs resident when the clock starts)
        def __init__(self, n):
            self.n = n

        def __len__(self):
            return self.n

        def __iter__(self):
            for _ in range(self.n):
                yield base, y

    validate(cfg, Resident(max(1, args.warmup // 4)), net, criterion, 0)
    if ddp:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    validate(cfg, Resident(args.steps), net, criterion, 0)
    torch.cuda.synchronize()
    if ddp:
        dist.barrier()
    dt = time.perf_counter() - t0
    if ddp:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    if rank == 0:
        clips = vids * args.views * world * args.steps
        print(json.dumps({
            "metric": "inference clips/sec (whole node) 3D-ResNet multi-view validation", "value": round(clips / dt, 2),
            "unit": "clips/s", "videos_per_s": round(clips / args.views / dt, 2), "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "3D-ResNet-%d validate(): %d views (10 clips x 3 crops) of %dx%dx%d per video, %d videos "
                                   "per step, %d clips per forward, %d classes, logits averaged per video"
                                   % (args.model, args.views, args.clip_len, args.crop, args.crop, vids, args.batch,
                                      cfg.MODEL.CLFDIM),
                       "global_batch": vids * args.views * world, "parallelism": "dp%d" % world},
            "validate": validate.last}), flush=True)
    if ddp:
        dist.destroy_process_group()


def main():
    args = parse()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if not launch.launched() and (args.gpus > 1 or args.dry_launch):
        # one command, N GPUs: N fresh rank processes; this parent never initialises HIP
        raise SystemExit(launch.fan_out(os.path.abspath(__file__), sys.argv[1:], args.gpus, dry=args.dry_launch))
    if os.environ.get(launch.DRY_ENV) == "1":
        rec = launch.dry_child_report()
        if rec["RANK"] == "0":
            print(json.dumps({"dry_launch": True, "n_gpus": int(rec["WORLD_SIZE"]), "gpus_requested": args.gpus}), flush=True)
        return
    global torch, dist, F
    import torch
    import torch.distributed as dist
    import torch.nn.functional as F
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = launch.local_device_index()
    if args.gpus != world:
        raise SystemExit("bench.py: --gpus %d but the launch has WORLD_SIZE %d; refusing to report a line for the wrong "
                         "job size (run `python bench.py --gpus %d` by itself, or torch.distributed.run with "
                         "--nproc-per-node %d)" % (args.gpus, world, args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # VC_DDP_FORCE_COMM=1 under a one-rank torchrun: keep the RCCL bucket path in the step (a rehearsal of N > 1)
    ddp = world > 1 or (os.environ.get("VC_DDP_FORCE_COMM") == "1" and "MASTER_PORT" in os.environ)
    if ddp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
    if args.model is None:
        args.model = 18  # the same workload at every N: a scaling curve compares like with like
    if ddp and dist.get_world_size() != world:
        raise SystemExit("bench.py: the process group has %d ranks, WORLD_SIZE says %d" % (dist.get_world_size(), world))

    import vc3d
    from vc3d import ops
    from vc3d.ddp import DataParallel3D
    from core.config import config
    from core.loss import Loss
    from models.model import create_model
    from utils.utils import create_optimizer

    cfg = config.clone()
    cfg.MODEL.RESNET_TYPE = args.model
    cfg.MODEL.CLFDIM = args.classes or (101 if args.model == 18 else 400)
    cfg.MODEL.INIT_WEIGHTS = False
    cfg.TRAIN.OPTIMIZER = args.optimizer
    cfg.TRAIN.LR = 0.001
    torch.manual_seed(0)
    model = create_model(cfg, is_train=True).to(dev)
    criterion = Loss(cfg).to(dev)
    optimizer = create_optimizer(cfg, model)
    net = DataParallel3D(model) if ddp else model
    arm = getattr(optimizer, "arm_backward", lambda: False)

    g = torch.Generator().manual_seed(1234 + rank)
    x = torch.randn(args.batch, 3, args.clip_len, args.crop, args.crop, generator=g).to(torch.bfloat16).to(dev)
    y = torch.randint(0, cfg.MODEL.CLFDIM, (args.batch,), generator=g).to(dev)

    if args.eval and args.views > 0:
        return bench_multi_view(args, cfg, net, criterion, dev, rank, world, ddp)
    if args.eval:
        net.eval()

        def step():
            with torch.no_grad():
                net(x)
    else:
        net.train()

        def step():
            _, score, _ = net(x)
            loss = criterion(score, y, None, None)
            optimizer.zero_grad()
            arm()  # (this backward makes the step's complete gradients: the fused optimizer may step inside it)
            loss.backward()
            optimizer.step()

    # ---- warm-up; the last warm-up step times every conv launch to find the dominant kernel
    # (that one step runs with the weight gradients on the main stream: event-bracketed times are exclusive kernel
    # durations only when nothing runs beside the kernel; the timed region below keeps the two-stream overlap)
    import vc3d.engine as _eng
    for i in range(max(args.warmup, 1)):
        last = i == max(args.warmup, 1) - 1
        if last:
            torch.cuda.synchronize()
            ops.TIMER = ops.KernelTimer()
            side, _eng.SIDE_WGRAD = _eng.SIDE_WGRAD, False
        step()
        if last:
            _eng.SIDE_WGRAD = side
    torch.cuda.synchronize()
    table = ops.TIMER.summary()
    ops.TIMER = None
    # the launch geometry with the largest launches x exclusive time, among the FORWARD launches: those have the GPU to
    # themselves in the timed region as well, so their live event-bracketed duration is a kernel time.  A backward
    # launch shares the GPU with the weight gradients of the side stream (layer1's 3x3x3 weight gradient, 4 x 0.126 ms
    # exclusive, is level with the stem forward's 0.51 ms and reads 0.24 ms per launch live): its live duration would
    # price the overlap, not the kernel.  The backward kernels are in roofline_by_time (exclusive times) and profiles/.
    fwd_only = {k: v for k, v in table.items() if k[0] in ("fwd", "stem_fwd")}
    dominant = max(fwd_only or table, key=lambda k: table[k][0] * table[k][1]) if table else None
    by_time = family_rooflines(table)
    if args.profile_kernels and rank == 0:
        tot = sum(v[0] * v[1] for v in table.values())
        for k, (n, ms, fl) in sorted(table.items(), key=lambda kv: -kv[1][0] * kv[1][1]):
            print("%-70s x%d %8.3f ms  %7.1f TFLOP/s  %5.1f%%" % (k, n, ms, fl / ms / 1e9, 100 * n * ms / tot),
                  file=sys.stderr)
        print("conv total %.3f ms per step" % tot, file=sys.stderr)

    # ---- timed region
    # one launch of the dominant geometry per step is timed (its first): an event pair costs the stream a few
    # microseconds, which a geometry launched several times per step (R3D-50's pointwise convs) would turn into a
    # visible share of a forward-only step
    timed = [False]

    def pick(k):
        if k != dominant or timed[0]:
            return False
        timed[0] = True
        return True
    ops.TIMER = ops.KernelTimer(select=pick, key=dominant) if dominant is not None else None
    if ddp:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        timed[0] = False
        step()
    torch.cuda.synchronize()
    if ddp:
        dist.barrier()
    dt = time.perf_counter() - t0
    if ddp:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    roof = None
    if ops.TIMER is not None:
        summ = ops.TIMER.summary()
        ops.TIMER = None
        if dominant in summ:
            n, ms, fl = summ[dominant]
            roof = launch_roofline(dominant, ms, fl)
            roof.update({"traffic": pmc_traffic(dominant), "launches_timed": n, "avg_ms": round(ms, 4),
                         "flops_per_launch": fl})

    if rank == 0:
        clips = args.batch * world * args.steps
        value = clips / dt
        gf = GFLOP_PER_CLIP.get((args.model, args.clip_len, args.crop))
        out = {
            "metric": "train clips/sec (whole node) 3D-ResNet 16x112x112" if not args.eval
            else "inference clips/sec (whole node) 3D-ResNet forward-only",
            "value": round(value, 2), "unit": "clips/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "3D-ResNet-%d %s, synthetic %dx%dx%d clips, %d classes, batch %d per GPU, "
                                   "%s" % (args.model, "forward-only" if args.eval else "bf16 training",
                                           args.clip_len, args.crop, args.crop, cfg.MODEL.CLFDIM, args.batch,
                                           "SGD momentum" if args.optimizer == "sgd" else "Adam"),
                       "global_batch": args.batch * world, "parallelism": "dp%d" % world},
            "roofline": roof,
            # the kernel FAMILY with the largest aggregate share of the step's conv time (one warm-up step with every
            # launch on one stream, so the event-bracketed times are exclusive), beside the single dominant launch above
            "roofline_by_time": by_time,
        }
        if gf is not None and not args.eval:
            out["mfma_frac_end_to_end"] = round(value * gf / 1e3 / PEAK_BF16_TFLOPS, 4)
        if world == 1 and not args.no_cpu_baseline and not args.eval:
            out["cpu_baseline"] = cpu_baseline(args.cpu_steps)
        print(json.dumps(out), flush=True)
    if ddp:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
