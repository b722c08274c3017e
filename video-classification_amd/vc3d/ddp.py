"""Data-parallel training: one process per GPU, gradients averaged with bucketed all-reduce
(RCCL over xGMI; torch.distributed backend "nccl") overlapped with the rest of backward.

Replaces torch.nn.DataParallel at reference train.py:42-43 / valid.py:42-43: no per-step
scatter / replicate / gather; each rank keeps a persistent replica and its own clips
(global batch = BATCH_SIZE x world size, reference train.py:65); BatchNorm statistics stay
per replica, as under DataParallel.

The engine writes gradients into one flat fp32 buffer in the order they become final, so a
bucket is a contiguous slice: no copy in, no copy out.  When the last gradient of a bucket
has been issued on the compute stream, an event is recorded and the bucket's all-reduce is
enqueued on a side stream behind that event; the compute stream joins the side stream only
at the end of backward.

Streams: backward uses the caller's stream plus the engine's weight-gradient stream; the reducer adds its own
and RCCL one more.  HIP multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (4 by default): with four
streams live, two of them share a queue and the overlap turns into head-of-line blocking (measured on one
MI355X with the buckets forced on: 6.70 ms/step at 4 queues, 6.02 at 8).  The entry points (bench.py, train.py,
valid.py) therefore default GPU_MAX_HW_QUEUES to 8 before the HIP runtime starts.
"""
import os

import torch
import torch.distributed as dist

# VC_DDP_FORCE_COMM=1: issue the bucket all-reduces even in a one-rank group, so that the RCCL path (side
# stream, events, AVG reduction, join) can be exercised and timed on a single GPU (tests/test_ddp_gpu.py)
FORCE_COMM = os.environ.get("VC_DDP_FORCE_COMM", "0") == "1"


class GradBucketReducer:
    def __init__(self, flat, ranges, group=None, bucket_bytes=32 << 20):
        """flat: the flat gradient tensor; ranges: [(key, offset, numel)] in readiness order."""
        self.flat = flat
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.backend = dist.get_backend(group) if dist.is_initialized() else None
        self.buckets = []  # [start, end, nparams]
        self.key2bucket = {}
        cur_start, cur_n = 0, 0
        esz = flat.element_size()
        for key, off, n in ranges:
            if cur_n == 0:
                cur_start = off
            self.key2bucket[key] = len(self.buckets)
            cur_n += 1
            if (off + n - cur_start) * esz >= bucket_bytes:
                self.buckets.append([cur_start, off + n, cur_n])
                cur_n = 0
        if cur_n:
            last_key, off, n = ranges[-1]
            self.buckets.append([cur_start, off + n, cur_n])
        self.comm_stream = torch.cuda.Stream() if flat.is_cuda else None
        self.producer_streams = lambda: ()  # extra streams that write gradients (the engine's weight-gradient stream)
        self.reset()

    def reset(self):
        self.pending = [b[2] for b in self.buckets]
        self.works = []
        self.launched = [False] * len(self.buckets)

    def on_ready(self, key):
        b = self.key2bucket.get(key)
        if b is None:
            return
        self.pending[b] -= 1
        if self.pending[b] == 0:
            self._launch(b)

    def _launch(self, b):
        if self.launched[b]:
            return
        self.launched[b] = True
        if self.world == 1 and not (FORCE_COMM and dist.is_initialized()):
            return
        start, end, _ = self.buckets[b]
        seg = self.flat[start:end]
        if self.flat.is_cuda:
            # the bucket's gradients were issued on the current stream and, for conv weights, on the engine's side
            # stream: the all-reduce waits for everything issued so far on each of them
            evs = []
            cur = torch.cuda.current_stream()
            for st in {cur, *[x for x in self.producer_streams() if x is not None]}:
                ev = torch.cuda.Event()
                ev.record(st)
                evs.append(ev)
            with torch.cuda.stream(self.comm_stream):
                for ev in evs:
                    self.comm_stream.wait_event(ev)
                if self.backend == "nccl":
                    w = dist.all_reduce(seg, op=dist.ReduceOp.AVG, group=self.group, async_op=True)
                else:
                    w = dist.all_reduce(seg, group=self.group, async_op=True)
            self.works.append((w, seg))
        else:
            dist.all_reduce(seg, group=self.group)
            seg.div_(self.world)

    def finish(self):
        """flush buckets that never filled (unused parameters), then join the side stream."""
        for b in range(len(self.buckets)):
            if not self.launched[b]:
                self._launch(b)
        if self.flat.is_cuda and self.works:
            for w, seg in self.works:
                with torch.cuda.stream(self.comm_stream):
                    w.wait()
                    if self.backend != "nccl":
                        seg.div_(self.world)
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        self.reset()


def broadcast_module(module, src=0, group=None):
    """one-time replica initialisation: parameters and buffers from rank `src`."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src, group=group)


class DataParallel3D(torch.nn.Module):
    """wraps a VedioClfNet for data-parallel training.  Usage mirrors
    ``torch.nn.DataParallel(model, device_ids=gpus).cuda()`` of the reference (train.py:43), except
    that every rank constructs it on its own GPU."""

    def __init__(self, model, group=None, bucket_mb=32):
        super().__init__()
        self.module = model
        self.group = group
        self.bucket_bytes = int(bucket_mb * (1 << 20))
        self.reducer = None
        broadcast_module(model, 0, group)
        eng = model.backbone.engine
        eng.grad_ready_cb = self._on_ready
        eng.backward_done_cb = self._on_done

    def _ensure(self):
        eng = self.module.backbone.engine
        if self.reducer is None or self.reducer.flat is not eng.flat_grad:
            ranges = [(id(p), off, n) for p, off, n in eng.grad_ranges]
            self.reducer = GradBucketReducer(eng.flat_grad, ranges, self.group, self.bucket_bytes)
            # gradients are written on the stream backward runs on (BatchNorm, head) and on the engine's side
            # stream (conv weights); a ready callback may come from inside either stream's context
            self.reducer.producer_streams = lambda: (eng._side, eng._main)
        return self.reducer

    def _on_ready(self, p):
        self._ensure().on_ready(id(p))

    def _on_done(self):
        self._ensure().finish()

    def forward(self, *a, **k):
        return self.module(*a, **k)
