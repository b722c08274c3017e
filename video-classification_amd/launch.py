"""One command, N GPUs: the fan-out the reference does inside one process with torch.nn.DataParallel
(/root/reference/train.py:42-43 `gpus = [int(i) for i in config.GPUS.split(',')]; DataParallel(model, device_ids=gpus)`,
global batch BATCH_SIZE * len(gpus) at :65; valid.py:42-43,55), done here as one process per GPU.

`fan_out` starts N fresh children of the calling script with the torch.distributed environment
(RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT), relays rank 0's stdout (the bench's one JSON line, the
logger's console lines) and returns the first non-zero exit code.  It must run BEFORE the parent touches the GPU
(nothing here imports torch or vc3d): children are new processes (subprocess), never a re-exec of a process that has
initialised HIP.

This module is stdlib only on purpose.
"""
import json
import os
import socket
import subprocess
import sys
import threading
import time

DRY_ENV = "VC_DRY_LAUNCH"
DEVICES_ENV = "VC_DEVICE_IDS"


def parse_gpus(spec):
    """config.GPUS ('0,1,2,3', reference config.py:16) -> [0, 1, 2, 3]; duplicates and junk are refused."""
    ids = [int(s) for s in str(spec).split(",") if s.strip() != ""]
    if not ids or len(set(ids)) != len(ids) or min(ids) < 0:
        raise ValueError("config.GPUS must be a comma list of distinct device ids, got %r" % (spec,))
    return ids


def launched():
    """True inside a rank process (ours or torch.distributed.run's)."""
    return "WORLD_SIZE" in os.environ and "RANK" in os.environ


def local_device_index():
    """The GPU this rank drives: the LOCAL_RANK-th entry of the launcher's device list, else LOCAL_RANK itself."""
    lr = int(os.environ.get("LOCAL_RANK", "0"))
    ids = os.environ.get(DEVICES_ENV)
    if ids:
        ids = parse_gpus(ids)
        if lr >= len(ids):
            raise RuntimeError("LOCAL_RANK %d but %s=%s" % (lr, DEVICES_ENV, os.environ[DEVICES_ENV]))
        return ids[lr]
    return lr


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def child_env(rank, world, port, device_ids=None, base=None):
    env = dict(os.environ if base is None else base)
    env.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_WORLD_SIZE": str(world),
                "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this driver: RCCL needs it
    if device_ids is not None:
        env[DEVICES_ENV] = ",".join(str(i) for i in device_ids)
    return env


def dry_child_report():
    """A child under --dry-launch: say who it is on stderr and stop before anything touches the GPU."""
    rec = {k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", DEVICES_ENV)}
    rec["device"] = local_device_index()
    rec["pid"] = os.getpid()
    os.write(2, ("DRY_LAUNCH_CHILD " + json.dumps(rec) + "\n").encode())  # one write: lines of N children do not interleave
    return rec


def fan_out(script, argv, n, device_ids=None, dry=False, timeout=None):
    """Run `python script argv...` as n rank processes on this node; returns the exit code for the parent.
    Rank 0's stdout is relayed line by line to ours; the other ranks' stdout goes to our stderr (so a consumer of
    stdout sees ONE JSON line); every child's stderr is ours.  If a rank fails, the others are terminated (by pid)."""
    if n < 1:
        raise ValueError("need at least one rank")
    if device_ids is not None and len(device_ids) != n:
        raise ValueError("%d ranks but device list %r" % (n, device_ids))
    port = _free_port()
    procs = []
    for r in range(n):
        env = child_env(r, n, port, device_ids)
        if dry:
            env[DRY_ENV] = "1"
        procs.append(subprocess.Popen([sys.executable, "-u", script] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=None, text=(r == 0)))
    t0 = time.time()
    code = 0

    def relay():  # rank 0's stdout (a JSON line, log lines) as it comes, so a long run shows progress
        for line in procs[0].stdout:
            sys.stdout.write(line)
            sys.stdout.flush()

    pump = threading.Thread(target=relay, daemon=True)
    pump.start()
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                rc = p.poll()
                if rc is None:
                    continue
                pending.remove(p)
                if rc != 0 and code == 0:
                    code = rc if rc > 0 else 128 - rc
                    print("launch: rank %d exited with %d; stopping the others" % (procs.index(p), rc), file=sys.stderr)
                    for q in pending:
                        q.terminate()
            if timeout is not None and time.time() - t0 > timeout and pending:
                code = code or 124
                for q in pending:
                    q.kill()
            time.sleep(0.02)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
                p.wait()
        pump.join(timeout=5)
    return code
