"""CPU tests of the drop-in boundary and the host logic (no GPU, no compute launches).

 - libvc3d.so loads and exports every symbol include/vc3d.h declares; shape logic that runs
   on the host (output dims, workspace sizes, argument validation) answers without a GPU;
 - the reference's Python surface is mirrored: config keys, create_model / getBackbone /
   BackBone attribute names and state_dict keys, Loss, create_optimizer, AverageMeter,
   init_weights behaviour (reference file:line in each test);
 - the product path refuses CPU tensors loudly (there is no fallback);
 - data-parallel gradient bucketing: 2 ranks over gloo.
"""
import ctypes
import os
import re
import sys

import pytest
import torch
import torch.nn as nn

from helpers import make_config

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ------------------------------------------------------------------ C ABI
def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "vc3d.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vc_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from vc3d import capi
    syms = _declared_symbols()
    assert len(syms) >= 30
    raw = ctypes.CDLL(capi.LIB_PATH)
    for s in syms:
        assert hasattr(raw, s), "libvc3d.so does not export %s" % s
    assert set(capi.PROTOTYPES) == set(syms), "ctypes binding and header disagree: %s" % (
        set(capi.PROTOTYPES) ^ set(syms))
    assert capi.lib.vc_abi_version() == 2


def test_host_side_shape_logic():
    from vc3d import capi, ops
    d = ops.conv_desc(64, 64, 8, 28, 28, 128, 3, 2, 1)
    to, ho, wo = (ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32())
    assert capi.lib.vc_conv3d_out_dims(d, to, ho, wo) == 0
    assert (to.value, ho.value, wo.value) == (4, 14, 14) == d.out_dims()
    # same arithmetic as torch's Conv3d for every (k, s, p) of the layer table
    for (T, H, W, k, s, p) in [(16, 56, 56, 3, 1, 1), (1, 7, 7, 3, 2, 1), (8, 28, 28, 1, 2, 0), (5, 9, 7, 3, 2, 1)]:
        dd = ops.conv_desc(1, 64, T, H, W, 64, k, s, p)
        want = tuple(nn.Conv3d(1, 1, k, s, p)(torch.zeros(1, 1, T, H, W)).shape[2:])
        assert dd.out_dims() == want
    pd = ops.pool_desc(2, 64, 16, 56, 56)
    assert capi.lib.vc_pool3d_out_dims(pd, to, ho, wo) == 0
    assert (to.value, ho.value, wo.value) == (8, 28, 28)
    assert capi.lib.vc_stem_packed_input_elems(2, 16, 112, 112) == 2 * 22 * 59 * 60 * 12
    # workspace query is pure host arithmetic; a big grid needs none
    assert capi.lib.vc_conv3d_workspace_bytes(ops.conv_desc(64, 64, 8, 28, 28, 64, 3, 1, 1), 0) == 0


def test_bad_descriptors_are_rejected_on_the_host():
    from vc3d import capi, ops
    to, ho, wo = (ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32())
    bad = ops.conv_desc(1, 64, 2, 2, 2, 64, 7, 1, 0)  # kernel larger than the padded input
    assert capi.lib.vc_conv3d_out_dims(bad, to, ho, wo) != 0
    assert b"conv3d" in capi.lib.vc_last_error()
    # channel counts the MFMA tiling cannot take: rejected before any launch
    d = ops.conv_desc(1, 48, 2, 4, 4, 64, 3, 1, 1)
    rc = capi.lib.vc_conv3d_fwd(d, None, None, None, None, None, 0, None)
    assert rc != 0 and b"multiples of 64" in capi.lib.vc_last_error()
    d = ops.conv_desc(1, 64, 2, 4, 4, 64, 3, 1, 1)
    rc = capi.lib.vc_conv3d_fwd(d, None, None, None, None, None, 0, None)
    assert rc != 0 and b"null pointer" in capi.lib.vc_last_error()
    with pytest.raises(capi.Vc3dError):
        capi.check(rc, "vc_conv3d_fwd")


def test_no_cpu_fallback():
    """a CPU tensor must raise, not silently run somewhere else."""
    from vc3d.capi import Vc3dError
    from models.model import create_model
    model = create_model(make_config(18, 11), is_train=True)
    with pytest.raises(Vc3dError, match="no CPU fallback|GPU only"):
        model(torch.zeros(1, 3, 4, 16, 16))
    from vc3d import ops
    with pytest.raises(Vc3dError):
        ops.bn_apply(torch.zeros(1, 1, 1, 1, 64, dtype=torch.bfloat16), torch.zeros(4, 64), 64)


# ------------------------------------------------------------------ reference API mirror
def test_config_keys_match_reference():
    """reference lib/core/config.py:12-69 (key names and defaults)."""
    from core.config import config
    assert (config.GPUS, config.WORKERS, config.DATASET, config.OUTPUT_DIR) == ('0', 0, 'FCVID', 'experiments')
    assert config.MODEL.LSTM_INDIM == 2048 and config.MODEL.LSTM_OUTDIM == 2048 and config.MODEL.CLFDIM == 100
    assert config.MODEL.RESNET_TYPE == 18 and config.MODEL.INIT_WEIGHTS is True
    t = config.TRAIN
    assert (t.LR, t.LR_DECAY_RATE, t.LR_MILESTONES, t.OPTIMIZER, t.MOMENTUM, t.WD, t.NESTEROV) == (
        0.001, 1, [], 'adam', 0.9, 0.0001, False)
    assert (t.R_ACT, t.R_CRT, t.TRAIN_STEP, t.ROLLOUT_STEP, t.BEGIN_EPOCH, t.END_EPOCH, t.BATCH_SIZE, t.SHUFFLE,
            t.PRINT_EVERY) == (1, 1, 10, 1, 0, 100, 32, True, 100)
    assert (config.TEST.TEST_STEP, config.TEST.BATCH_SIZE, config.TEST.PRINT_EVERY) == (10, 32, 100)
    c2 = config.clone()
    c2.TRAIN.LR = 1.0
    assert config.TRAIN.LR == 0.001


@pytest.mark.parametrize("depth,convs,feat", [(18, 20, 512), (50, 53, 2048)])
def test_backbone_names_and_counts(depth, convs, feat):
    """attribute names = checkpoint keys (reference backbone.py:108-125), conv/BN counts of the 2D
    reference, state_dict key set identical to the oracle's."""
    from models.backbone import getBackbone, resnet_spec, BackBone
    from oracle.r3d_torch import BackBone3D, resnet_spec as ospec
    cfg = make_config(depth, 101)
    bb = getBackbone(cfg, is_train=True)
    assert isinstance(bb, BackBone) and bb.out_features == feat
    for name in ("conv1", "bn1", "relu", "maxpool", "layer1", "layer2", "layer3", "layer4", "final_layer"):
        assert hasattr(bb, name)
    assert sum(isinstance(m, nn.Conv3d) for m in bb.modules()) == convs
    assert sum(isinstance(m, nn.BatchNorm3d) for m in bb.modules()) == convs
    assert all(m.momentum == 0.1 for m in bb.modules() if isinstance(m, nn.BatchNorm3d))  # backbone.py:13
    assert all(m.bias is None for m in bb.modules() if isinstance(m, nn.Conv3d))
    ob = BackBone3D(*ospec[depth])
    assert set(bb.state_dict()) == set(ob.state_dict())
    assert {k: tuple(v.shape) for k, v in bb.state_dict().items()} == {k: tuple(v.shape) for k, v in
                                                                        ob.state_dict().items()}
    assert sorted(resnet_spec) == [18, 34, 50, 101, 152]  # backbone.py:188-192
    # Bottleneck carries the stride on the 3x3 (backbone.py:72)
    if depth == 50:
        blk = bb.layer2[0]
        assert blk.conv1.stride == (1, 1, 1) and blk.conv2.stride == (2, 2, 2)


def test_create_model_returns_reference_triple_shape_contract():
    from models.model import create_model, VedioClfNet
    m = create_model(make_config(18, 101), is_train=True)
    assert isinstance(m, VedioClfNet)
    assert isinstance(m.clf_haed, nn.Linear) and m.clf_haed.in_features == 512 and m.clf_haed.out_features == 101


def test_init_weights_reference_behaviour(tmp_path):
    """backbone.py:159-185: missing file -> ValueError; {'state_dict': ...} accepted; 'module.'
    prefix stripped; strict=False; 2D kernels are inflated along T (repeat / kT)."""
    from models.backbone import getBackbone
    cfg = make_config(18, 101)
    cfg.MODEL.INIT_WEIGHTS = True
    cfg.MODEL.PRETRAINED_PATH = str(tmp_path / "nope.pth")
    with pytest.raises(ValueError):
        getBackbone(cfg, is_train=True)
    getBackbone(cfg, is_train=False)  # not loaded when is_train is False (backbone.py:210)
    bb = getBackbone(make_config(18, 101), True)
    w2d = torch.randn(64, 64, 3, 3)
    sd = {"module.layer1.0.conv1.weight": w2d, "module.bn1.weight": torch.full((64,), 2.0), "unknown.key": torch.ones(1)}
    path = str(tmp_path / "ck.pth")
    torch.save({"state_dict": sd}, path)
    bb.init_weights(path)
    w = bb.layer1[0].conv1.weight
    assert torch.allclose(w.sum(2), w2d, atol=1e-6) and torch.allclose(w[:, :, 0], w2d / 3)
    assert float(bb.bn1.weight[0]) == 2.0
    assert w.is_contiguous(memory_format=torch.channels_last_3d)


def test_optimizer_meter_logger(tmp_path):
    """reference utils.py:13-36 (create_optimizer), :39-54 (AverageMeter), :84-104 (create_logger)."""
    from utils.utils import create_optimizer, AverageMeter, create_logger, ChooseInput, rollout
    model = nn.Linear(4, 2)
    cfg = make_config(18, 10, OPTIMIZER="sgd", NESTEROV=True)
    opt = create_optimizer(cfg, model)
    g = opt.param_groups[0]
    assert isinstance(opt, torch.optim.SGD) and (g["lr"], g["momentum"], g["weight_decay"], g["nesterov"]) == (
        0.001, 0.9, 0.0001, True)
    cfg.TRAIN.OPTIMIZER = "adam"
    assert isinstance(create_optimizer(cfg, model), torch.optim.Adam)
    cfg.TRAIN.OPTIMIZER = "lamb"
    assert create_optimizer(cfg, model) is None  # utils.py:21,36
    m = AverageMeter()
    assert m.avg == 0 and m.count == 0
    m.update(2.0, 3)
    m.update(4.0, 1)
    assert (m.val, m.sum, m.count, m.avg) == (4.0, 10.0, 4, 2.5)
    cfg.OUTPUT_DIR = str(tmp_path / "out")
    lg = create_logger(cfg, "train")
    lg.info("hello")
    files = os.listdir(cfg.OUTPUT_DIR)
    assert len(files) == 1 and files[0].startswith("FCVID_") and files[0].endswith("_train.log")
    for h in list(lg.handlers):
        lg.removeHandler(h)
    v = torch.zeros(1)
    assert ChooseInput(v, None) is v and rollout(v, None, None, None) == (None, None)


def test_synthetic_dataset_shapes():
    from dataset.dataset import get_dataset
    cfg = make_config(18, 7)
    cfg.MODEL.CLIP_LEN, cfg.MODEL.CROP_SIZE = 4, 16
    ds = get_dataset(True, cfg, length=3)
    x, y = ds[1]
    assert len(ds) == 3 and tuple(x.shape) == (3, 4, 16, 16) and 0 <= y < 7
    x2, y2 = ds[1]
    assert torch.equal(x, x2) and y == y2


# ------------------------------------------------------------------ data parallel (gloo, 2 ranks)
def _ddp_worker(rank, world, port, out, grad_dtype="fp32"):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from vc3d.ddp import GradBucketReducer, broadcast_module
        torch.manual_seed(100 + rank)
        mod = nn.Sequential(nn.Linear(5, 3), nn.BatchNorm1d(3))
        broadcast_module(mod, 0)
        w0 = mod[0].weight.detach().clone()
        sizes = [7, 130, 5, 64, 9]  # several parameters per bucket, one bucket boundary inside
        flat = torch.zeros(sum(sizes))
        ranges, off = [], 0
        for i, n in enumerate(sizes):
            ranges.append((i, off, n))
            off += n
        red = GradBucketReducer(flat, ranges, bucket_bytes=4 * 140, tail_bytes=4 * 9, grad_dtype=grad_dtype)
        nb = len(red.buckets)
        for step in range(2):  # the reducer is reused across steps
            flat.copy_(torch.arange(flat.numel(), dtype=torch.float32) * (rank + 1) + step)
            for i in range(len(sizes) - 1):  # the last parameter never reports (unused): finish() flushes it
                red.on_ready(i)
            red.finish()
            want = torch.arange(flat.numel(), dtype=torch.float32) * (sum(range(1, world + 1)) / world) + step
            # bf16 buckets: each rank's slice is rounded once on the way out and the average once on the way back
            assert torch.allclose(flat, want, rtol=1e-2 if grad_dtype == "bf16" else 1e-5), (rank, step)
        gathered = [torch.zeros_like(w0) for _ in range(world)]
        dist.all_gather(gathered, w0)
        assert all(torch.equal(gathered[0], t) for t in gathered)
        out.put((rank, nb))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("grad_dtype", ["fp32", "bf16"])
def test_grad_bucket_reducer_two_ranks_gloo(grad_dtype):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = 29500 + (os.getpid() % 2000) + (7 if grad_dtype == "bf16" else 0)
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q, grad_dtype)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = sorted(q.get() for _ in range(2))
    assert [r for r, _ in res] == [0, 1] and res[0][1] >= 2


def test_bucket_layout_single_process():
    from vc3d.ddp import GradBucketReducer
    flat = torch.zeros(100)
    # cut from the end: the bucket that finishes last holds at most tail_bytes, the others bucket_bytes
    red = GradBucketReducer(flat, [("a", 0, 30), ("b", 30, 30), ("c", 60, 40)], bucket_bytes=4 * 50, tail_bytes=4 * 40)
    assert [b[:2] for b in red.buckets] == [[0, 60], [60, 100]]
    assert red.key2bucket == {"a": 0, "b": 0, "c": 1}
    red.on_ready("a")
    assert red.launched == [False, False]
    red.on_ready("b")
    assert red.launched == [True, False]
    red.finish()
    assert red.launched == [False, False]  # reset for the next step
    # a small tail: the last (exposed) bucket is only the last parameter
    red = GradBucketReducer(flat, [("a", 0, 30), ("b", 30, 30), ("c", 60, 20), ("d", 80, 20)], bucket_bytes=4 * 60,
                            tail_bytes=4 * 10)
    assert [b[:2] for b in red.buckets] == [[0, 80], [80, 100]]
    # injected collective (test hook) and no_sync-style disabling
    calls = []
    red.collective = lambda seg: calls.append(seg.numel())
    red.enabled = False
    for k in "abcd":
        red.on_ready(k)
    red.finish()
    assert calls == []
    red.enabled = True
    for k in "abcd":
        red.on_ready(k)
    red.finish()
    assert calls == [80, 20]


def test_c_abi_host_code_under_sanitizers():
    """The library's host side (planners, tap-table builders, split and sub-batch logic, argument checks) compiled with
    AddressSanitizer + UndefinedBehaviorSanitizer (host code only: hipcc -Xarch_host -fsanitize=...) and driven through
    every conv / stem / pool / BatchNorm entry point of include/vc3d.h for the R3D-18/34/50 layer geometries at the
    BASELINE clip sizes, a ragged clip, the T = 1 reference geometry and the refused shapes (tests/abi_host_check.cpp).
    There is no GPU here: every launch fails with a message after its host-side set-up has run; the check is that no
    call crashes and neither sanitizer reports (UBSan is built to abort).  The same program replays a command list
    (vc_replay) whose entry point takes floats between its pointers and more integer arguments than registers, and
    requires the replayed call to fail exactly as the direct call does (same code, same message: every argument arrived in
    its place), plus the refusals of malformed lists.  First run builds the binary (~20 s)."""
    import shutil
    import subprocess
    if shutil.which("make") is None or not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("needs make and hipcc")
    csrc = os.path.join(ROOT, "video-classification_amd", "csrc")
    r = subprocess.run(["make", "-C", csrc, "-j8", "hostcheck"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    # (verify_asan_link_order=0: the environment may preload a library of its own ahead of the sanitizer runtime)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:verify_asan_link_order=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([os.path.join(csrc, "build_asan", "abi_host_check")], capture_output=True, text=True, timeout=300,
                       env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
    assert r.stdout.startswith("calls ") and int(r.stdout.split()[1].rstrip(",")) > 5000


# ---- one command, N GPUs (reference train.py:42-43,65; valid.py:42-43,55): the launcher, without a GPU

def _dry_children(stderr):
    import json as _json
    return sorted((_json.loads(ln.split(" ", 1)[1]) for ln in stderr.splitlines() if ln.startswith("DRY_LAUNCH_CHILD ")),
                  key=lambda r: int(r["RANK"]))


@pytest.mark.parametrize("n", [2, 8])
def test_bench_gpus_n_starts_n_rank_processes(n):
    """`python bench.py --gpus N` by itself (the driver's command form) starts N fresh rank processes with the
    torch.distributed environment and prints ONE JSON line whose n_gpus is the world size; --dry-launch stops each
    child before it touches the GPU."""
    import json as _json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--dry-launch"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    kids = _dry_children(r.stderr)
    assert [int(k["RANK"]) for k in kids] == list(range(n)) and [k["device"] for k in kids] == list(range(n))
    assert all(k["WORLD_SIZE"] == str(n) and k["MASTER_ADDR"] == "127.0.0.1" for k in kids)
    assert len({k["MASTER_PORT"] for k in kids}) == 1 and len({k["pid"] for k in kids}) == n
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and _json.loads(lines[0]) == {"dry_launch": True, "n_gpus": n, "gpus_requested": n}


def test_bench_refuses_a_world_size_that_is_not_gpus():
    import subprocess
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], env=env, capture_output=True,
                       text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE 2" in r.stderr and not r.stdout.strip()


@pytest.mark.parametrize("script", ["train.py", "valid.py"])
def test_config_gpus_fans_out_one_rank_per_listed_device(script):
    """config.GPUS = '1,3,6' (reference config.py:16, train.py:42): three ranks, rank r on the r-th listed device."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["VC_DRY_LAUNCH_PARENT"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "video-classification_amd", script), "GPUS=1,3,6",
                        "TRAIN.END_EPOCH=1"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    kids = _dry_children(r.stderr)
    assert [k["device"] for k in kids] == [1, 3, 6] and all(k["WORLD_SIZE"] == "3" for k in kids)
    assert all(k["VC_DEVICE_IDS"] == "1,3,6" for k in kids)


def test_launcher_propagates_a_failing_rank_and_stops_the_others(tmp_path):
    import launch
    import subprocess
    script = tmp_path / "rank.py"
    script.write_text("import os, sys, time\n"
                      "if os.environ['RANK'] == '1':\n    sys.exit(7)\n"
                      "print('rank', os.environ['RANK'], flush=True)\ntime.sleep(60)\n")
    drv = tmp_path / "drv.py"
    drv.write_text("import sys\nsys.path.insert(0, %r)\nimport launch\nsys.exit(launch.fan_out(%r, [], 3))\n"
                   % (os.path.dirname(launch.__file__), str(script)))
    import time as _time
    t0 = _time.time()
    r = subprocess.run([sys.executable, str(drv)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 7 and _time.time() - t0 < 30
    assert r.stdout.strip() == "rank 0"  # only rank 0's stdout reaches ours
    with pytest.raises(ValueError):
        launch.parse_gpus("0,0")
    assert launch.parse_gpus("0, 2,5") == [0, 2, 5]
