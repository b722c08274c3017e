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
import contextlib
import os

import torch
import torch.distributed as dist

# VC_DDP_FORCE_COMM=1: issue the bucket all-reduces even in a one-rank group, so that the RCCL path (side
# stream, events, AVG reduction, join) can be exercised and timed on a single GPU (tests/test_ddp_gpu.py)
FORCE_COMM = os.environ.get("VC_DDP_FORCE_COMM", "0") == "1"


class GradBucketReducer:
    """Buckets = contiguous slices of the flat fp32 gradient buffer, in the order the engine finishes them.

    Sizes (xGMI arithmetic, SURVEY section 5: 7 links x ~153 GB/s per GPU, point to point).  R3D-50 has 46 M parameters:
    185 MB of fp32 gradients, 92 MB as bf16.  An 8-GPU all-reduce moves 2 x 7/8 of the message per GPU; RCCL spreads it
    over rings on different links, so a LARGE message sees a few hundred GB/s of bus bandwidth (92 MB in about
    0.5 ms), a small one only its latency (tens of microseconds per ring step): buckets of 32 MB keep the bandwidth
    term dominant, and 3-6 of them per step keep every all-reduce but the last behind the rest of backward (12 ms
    today, 3 ms at the MFMA target).  What cannot be hidden is the LAST bucket -- it is launched when backward ends --
    so the buckets are cut from the end: the final one holds at most `tail_bytes` (the stem and layer1, whose
    gradients are small and finish last), about 20 us of exposed all-reduce instead of a full bucket's.
    `grad_dtype` = "bf16" halves the bytes on the wire: the slice is rounded into a bf16 staging buffer on the
    communication stream, averaged there, and written back to the fp32 buffer (the optimizer's master gradients and
    parameters stay fp32; 2^-9 relative rounding per element, below the bf16 noise already in the gradients)."""

    def __init__(self, flat, ranges, group=None, bucket_bytes=32 << 20, tail_bytes=4 << 20, grad_dtype="fp32"):
        """flat: the flat gradient tensor; ranges: [(key, offset, numel)] in readiness order."""
        self.flat = flat
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.backend = dist.get_backend(group) if dist.is_initialized() else None
        if grad_dtype not in ("fp32", "bf16"):
            raise ValueError("grad_dtype must be 'fp32' or 'bf16'")
        self.grad_dtype = grad_dtype
        self.collective = None  # test hook: callable(slice) run on the communication stream in place of the all-reduce
        # callable(start, end) run on the communication stream right behind a bucket's all-reduce: a fused optimizer that
        # steps the bucket's parameters while the rest of the backward runs (optim.arm_backward)
        self.after_reduce = None
        self.enabled = True     # False inside DataParallel3D.no_sync()
        esz = flat.element_size()
        # cut from the end so that the bucket that finishes last (exposed at the end of backward) is the small one
        cuts, limit, hi = [], tail_bytes, None
        for i in range(len(ranges) - 1, -1, -1):
            _, off, n = ranges[i]
            if hi is None:
                hi, hi_i = off + n, i
            if (hi - off) * esz >= limit or i == 0:
                cuts.append((i, hi_i, off, hi))
                hi, limit = None, bucket_bytes
        self.buckets = []  # [start, end, nparams]
        self.key2bucket = {}
        for lo_i, hi_i, start, end in reversed(cuts):
            for i in range(lo_i, hi_i + 1):
                self.key2bucket[ranges[i][0]] = len(self.buckets)
            self.buckets.append([start, end, hi_i - lo_i + 1])
        self.comm_stream = torch.cuda.Stream() if flat.is_cuda else None
        self.comm_buf = torch.empty(flat.numel(), dtype=torch.bfloat16, device=flat.device) if grad_dtype == "bf16" else None
        self.producer_streams = lambda: ()  # extra streams that write gradients (the engine's weight-gradient stream)
        self.reset()

    def reset(self):
        self.pending = [b[2] for b in self.buckets]
        self.works = []
        self.launched = [False] * len(self.buckets)

    def on_ready(self, key):
        """-> True when this gradient completed a bucket and its all-reduce was launched"""
        if not self.enabled:
            return False
        b = self.key2bucket.get(key)
        if b is None:
            return False
        self.pending[b] -= 1
        if self.pending[b] == 0:
            self._launch(b)
            return True
        return False

    def _reduce(self, seg, start, end):
        """average `seg` over the group, on the current (communication) stream / thread."""
        if self.collective is not None:
            self.collective(seg)
            return
        buf = seg if self.comm_buf is None else self.comm_buf[start:end]
        if buf is not seg:
            buf.copy_(seg)  # fp32 -> bf16, one rounding
        if self.backend == "nccl":
            dist.all_reduce(buf, op=dist.ReduceOp.AVG, group=self.group)  # enqueued behind this stream, no host wait
        elif buf.is_cuda:  # gloo with device memory (tests: ranks sharing one GPU): staged through the host
            host = buf.cpu()  # waits for this stream, i.e. for the gradients of the bucket
            dist.all_reduce(host, group=self.group)
            buf.copy_(host.div_(self.world))
        else:  # gloo on host memory: no AVG
            dist.all_reduce(buf, group=self.group)
            buf.div_(self.world)
        if buf is not seg:
            seg.copy_(buf)

    def _launch(self, b):
        if self.launched[b]:
            return
        self.launched[b] = True
        if self.collective is None and self.world == 1 and not (FORCE_COMM and dist.is_initialized()):
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
                self._reduce(seg, start, end)
                if self.after_reduce is not None:
                    self.after_reduce(start, end)
            self.works.append(b)
        else:
            self._reduce(seg, start, end)
            if self.after_reduce is not None:
                self.after_reduce(start, end)

    def finish(self):
        """flush buckets that never filled (unused parameters), then join the side stream."""
        if not self.enabled:
            return
        for b in range(len(self.buckets)):
            if not self.launched[b]:
                self._launch(b)
        if self.flat.is_cuda and self.works:
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        self.reset()


def broadcast_module(module, src=0, group=None):
    """one-time replica initialisation: parameters and buffers from rank `src`."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    staged = dist.get_backend(group) != "nccl"  # gloo's device-memory collectives hang between ranks that share a GPU
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            if staged and t.is_cuda:
                host = t.data.cpu()
                dist.broadcast(host, src, group=group)
                t.data.copy_(host)
            else:
                dist.broadcast(t.data, src, group=group)


class DataParallel3D(torch.nn.Module):
    """wraps a VedioClfNet for data-parallel training.  Usage mirrors
    ``torch.nn.DataParallel(model, device_ids=gpus).cuda()`` of the reference (train.py:43), except
    that every rank constructs it on its own GPU."""

    def __init__(self, model, group=None, bucket_mb=32, tail_mb=4, grad_dtype=None):
        super().__init__()
        self.module = model
        self.group = group
        self.bucket_bytes = int(bucket_mb * (1 << 20))
        self.tail_bytes = int(tail_mb * (1 << 20))
        # Gradients travel as fp32 by default: the reference's DataParallel sums fp32 gradients (reduce_add_coalesced),
        # and a bf16 wire format rounds each rank's gradient AND every ring step of RCCL's bf16 reduction.  bf16 on the
        # wire (half the xGMI bytes; master gradients stay fp32) is opt-in: grad_dtype="bf16", config TRAIN.GRAD_DTYPE
        # through train.py, or VC_DDP_GRAD_DTYPE=bf16.
        self.grad_dtype = grad_dtype or os.environ.get("VC_DDP_GRAD_DTYPE", "fp32")
        self.reducer = None
        self.after_reduce = None  # see GradBucketReducer.after_reduce (set per step by optim.arm_backward)
        self._extras_queued = False
        self._sync = True
        broadcast_module(model, 0, group)
        eng = model.backbone.engine
        eng.grad_ready_cb = self._on_ready
        eng.backward_done_cb = self._on_done

    def _ensure(self):
        eng = self.module.backbone.engine
        if self.reducer is None or self.reducer.flat is not eng.flat_grad:
            ranges = [(id(p), off, n) for p, off, n in eng.grad_ranges]
            self.reducer = GradBucketReducer(eng.flat_grad, ranges, self.group, self.bucket_bytes, self.tail_bytes,
                                             self.grad_dtype)
            # gradients are written on the stream backward runs on (BatchNorm, head) and on the engine's side
            # stream (conv weights); a ready callback may come from inside either stream's context
            self.reducer.producer_streams = lambda: (eng._side, eng._main)
        return self.reducer

    def _on_ready(self, p):
        r = self._ensure()
        r.enabled, r.after_reduce = self._sync, self.after_reduce
        return r.on_ready(id(p))

    def _on_done(self):
        r = self._ensure()
        r.enabled, r.after_reduce = self._sync, self.after_reduce
        r.finish()
        # Parameters that live outside the engine's flat buffer -- the recurrent head of MODEL.USE_LSTM (LSTMCell,
        # act_head, clf_haed, crit_head: reference model.py:18-25, wrapped by DataParallel as part of the whole model at
        # train.py:43) -- are averaged once the WHOLE backward pass has finished (autograd's end-of-backward callback,
        # as torch DDP does): replicas would otherwise drift apart silently.
        if self._sync and not self._extras_queued and self.extra_params():
            self._extras_queued = True
            try:
                torch.autograd.Variable._execution_engine.queue_callback(self._reduce_extras)
            except RuntimeError:  # engine.backward called outside autograd: every gradient is final already
                self._reduce_extras()

    def extra_params(self):
        eng = self.module.backbone.engine
        cache = getattr(self, "_extras_cache", None)  # (per flat-buffer layout: this runs at the end of every backward)
        if cache is None or cache[0] is not eng.grad_ranges:
            flat = {id(p) for p, _, _ in eng.grad_ranges}
            cache = self._extras_cache = (eng.grad_ranges, [p for p in self.module.parameters() if id(p) not in flat])
        return [p for p in cache[1] if p.requires_grad]

    def _reduce_extras(self):
        self._extras_queued = False
        if not dist.is_initialized():
            return
        world = dist.get_world_size(self.group)
        if world == 1 and not FORCE_COMM:
            return
        grads = [p.grad for p in self.extra_params() if p.grad is not None]
        if not grads:
            return
        flat = torch._utils._flatten_dense_tensors(grads)
        if dist.get_backend(self.group) == "nccl":
            dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=self.group)
        else:  # gloo (tests; ranks may share one GPU): staged through the host
            host = flat.cpu()
            dist.all_reduce(host, group=self.group)
            flat.copy_(host.div_(world))
        for g, v in zip(grads, torch._utils._unflatten_dense_tensors(flat, grads)):
            g.copy_(v)

    @contextlib.contextmanager
    def no_sync(self):
        """gradient accumulation (torch DDP's no_sync): backward passes inside the context only accumulate into the
        flat gradient buffer; the first backward outside it averages the accumulated sum.  Without it every backward
        all-reduces whatever the buffer holds, which would average an already averaged gradient a second time."""
        keep, self._sync = self._sync, False
        try:
            yield
        finally:
            self._sync = keep

    def forward(self, *a, **k):
        return self.module(*a, **k)
