"""RCCL rehearsal on one GPU: the bucketed all-reduce path of DataParallel3D (side stream, events, AVG)
under backend "nccl" with a one-rank group, and bench.py launched the way the driver launches N > 1."""
import json
import os
import re
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _torchrun(script_args, timeout=300):
    # VC_HALO_NO_PAD=1: with the unpadded halo form every kernel of this small configuration accumulates its
    # BatchNorm sums in a fixed order, so two plain runs agree to ~1e-5 and the comparison below is sharp (the
    # tiny 8-clip network amplifies last-bit differences of fp32 atomics to percents otherwise)
    env = dict(os.environ, VC_DDP_FORCE_COMM="1", HSA_ENABLE_IPC_MODE_LEGACY="0", VC_HALO_NO_PAD="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port())] + script_args
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


def test_bucketed_allreduce_single_rank_rccl_is_identity():
    r = _torchrun([os.path.join(ROOT, "tests", "ddp_worker.py")])
    assert r.returncode == 0, r.stderr[-2000:]
    m = re.search(r"DDP_RESULT grad_rel=(\S+) param_rel=(\S+) rerun_grad_rel=(\S+) rerun_param_rel=(\S+) loss=\S+ "
                  r"buckets=(\d+) finite=(\d)", r.stdout)
    assert m, r.stdout[-2000:]
    print(m.group(0))
    grad_rel, param_rel, rerun_grad, rerun_param = (float(m.group(i)) for i in range(1, 5))
    # an AVG all-reduce over one rank is the identity: the wrapped run may differ from the plain run only by
    # what two plain runs differ by (fp32 atomics in the BatchNorm sums make the steps not bit-reproducible)
    assert rerun_grad < 1e-3, "the plain step is expected to be repeatable in this configuration"
    # floors = the noise of fp32 atomics in the BatchNorm sums: 1e-7 .. 3e-5 between two plain runs, up to 2e-4 for
    # the wrapped run, whose RCCL kernels run beside the backward and change the arrival order of the atomics (seen
    # once in eight runs); a bucket reduced before its gradients were final shows up at 1e-2 and more
    assert grad_rel <= max(4 * rerun_grad, 2e-3) and param_rel <= max(4 * rerun_param, 2e-5)
    assert int(m.group(5)) > 1 and m.group(6) == "1"


def test_bench_under_torchrun_with_rccl_buckets():
    r = _torchrun([os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "2", "--batch", "8",
                   "--no-cpu-baseline"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 1 and out["value"] > 0 and out["roofline"]["frac"] > 0
