"""Command lists: a step's launch sequence recorded once and replayed by one C call (include/vc3d.h, vc_replay).

Recorder    : while `capi.RECORDER` is set, every entry point called through `capi.lib` is appended (function
              address, integer / pointer arguments, float arguments; descriptor structs are copied and kept), and the
              engine adds the cross-stream dependencies it creates with torch events (event_record / stream_wait).
CommandList : the frozen result -- a ctypes array of vc_cmd, the HIP events its dependencies use, the descriptor
              copies, and PATCH points: (command, argument) pairs that held the data pointer of a per-step input
              (the clip batch, the gradient of the features), rewritten before each replay.

The engine (vc3d/engine.py) decides when a recording is still valid; nothing here touches the GPU except the replay
call itself.
"""
import ctypes as C

from . import capi
from .capi import CMD_CALL, CMD_EVENT_RECORD, CMD_STREAM_WAIT, Cmd, Vc3dError

_INT_TYPES = (C.c_int32, C.c_int64, C.c_size_t, C.c_uint32, C.c_uint64)


class Recorder:
    def __init__(self):
        self.cmds = []      # (kind, fn_addr, ints, floats, name, tag)
        self.keep = []      # descriptor copies (their addresses are in the commands)
        self.n_events = 0
        self.tag = None     # set by ops around a conv launch: the KernelTimer key of the next command
        self.failed = None
        self.marks = []     # (number of commands recorded before it, payload): points where Python runs between commands
        self.opt_cmds = []  # indices of the commands a fused optimizer issued inside the backward (engine._ready)

    # ---- calls (capi._Lib)
    def add(self, name, fn, argtypes, args, rc):
        if rc != 0:
            self.failed = "%s returned %d while recording" % (name, rc)
            return
        ints, flts = [], []
        for t, a in zip(argtypes, args):
            if t is C.c_float:
                flts.append(float(a))
            elif t is C.c_void_p:
                ints.append(0 if a is None else int(a))
            elif t in _INT_TYPES:
                ints.append(int(a))
            elif isinstance(t, type) and issubclass(t, C._Pointer):
                if a is None:
                    ints.append(0)
                elif isinstance(a, C.Structure):
                    cp = type(a).from_buffer_copy(a)  # the caller may reuse or drop its descriptor
                    self.keep.append(cp)
                    ints.append(C.addressof(cp))
                elif isinstance(a, C.Array):
                    cp = type(a).from_buffer_copy(a)
                    self.keep.append(cp)
                    ints.append(C.addressof(cp))
                else:
                    self.failed = "%s: argument of type %r cannot be recorded" % (name, type(a))
                    return
            else:
                self.failed = "%s: argument type %r cannot be recorded" % (name, t)
                return
        if len(ints) > 16 or len(flts) > 8:
            self.failed = "%s: too many arguments for a command" % name
            return
        self.cmds.append((CMD_CALL, C.cast(fn, C.c_void_p).value, ints, flts, name, self.tag))
        self.tag = None

    # ---- cross-stream dependencies (the engine calls these beside its torch events)
    def event_record(self, stream_handle):
        ev = self.n_events
        self.n_events += 1
        self.cmds.append((CMD_EVENT_RECORD, 0, [ev, int(stream_handle)], [], "event_record", None))
        return ev

    def stream_wait(self, stream_handle, ev):
        self.cmds.append((CMD_STREAM_WAIT, 0, [ev, int(stream_handle)], [], "stream_wait", None))

    def mark(self, payload, cut=True):
        """a point between two commands where the replaying side has work of its own (the data-parallel wrapper's
        bucket launches): CommandList.replay_segments hands `payload` to its callback there.  cut=False: the callback
        enqueued nothing when the step was recorded (a gradient became final without completing a bucket) -- the
        replay need not stop for it and makes the call at the next mark that does cut, in order."""
        self.marks.append((len(self.cmds), payload, bool(cut)))

    def freeze(self, patch_values=()):
        """-> CommandList.  patch_values: {name: pointer value}; every integer argument equal to one of them becomes a
        patch point of that name."""
        if self.failed:
            raise Vc3dError("command list: " + self.failed)
        return CommandList(self, dict(patch_values))


class CommandList:
    def __init__(self, rec, patch_values):
        n = len(rec.cmds)
        self.n = n
        self.array = (Cmd * max(n, 1))()
        self.names = []
        self.tags = {}
        self.patches = {k: [] for k in patch_values}
        for i, (kind, fn, ints, flts, name, tag) in enumerate(rec.cmds):
            c = self.array[i]
            c.fn, c.kind, c.nint, c.nflt = fn, kind, len(ints), len(flts)
            for j, v in enumerate(ints):
                c.iarg[j] = v
                if kind == CMD_CALL:
                    for pname, pval in patch_values.items():
                        if pval and v == pval:
                            self.patches[pname].append((i, j))
            for j, v in enumerate(flts):
                c.farg[j] = v
            self.names.append(name)
            if tag is not None:
                self.tags.setdefault(tag, i)
        self.keep = rec.keep
        self.marks = list(rec.marks)
        self.n_events = rec.n_events
        self.events = (C.c_void_p * max(self.n_events, 1))()
        if self.n_events:
            capi.check(capi.lib.vc_replay_events_create(self.n_events, 0, self.events), "vc_replay_events_create")
        self._failed_at = C.c_int32(-1)

    def __del__(self):
        try:
            if self.n_events:
                capi.lib.vc_replay_events_destroy(self.n_events, self.events)
        except Exception:
            pass

    def patch(self, name, value):
        for i, j in self.patches[name]:
            self.array[i].iarg[j] = value

    def replay(self, time_cmd=-1, time_start=None, time_stop=None, first=0, end=None):
        """commands [first, end) (default: all of them)"""
        end = self.n if end is None else end
        if end <= first:
            return
        cmds = self.array if first == 0 else C.byref(self.array[first])
        rc = capi.lib.vc_replay(cmds, end - first, self.events, self.n_events, time_cmd - first if time_cmd >= first else -1,
                                time_start, time_stop, C.byref(self._failed_at))
        if rc != 0:
            i = self._failed_at.value + first
            raise Vc3dError("replay failed at command %d (%s): %s" % (i, self.names[i] if 0 <= i < self.n else "?",
                                                                      capi.lib.vc_last_error().decode()))

    def replay_segments(self, on_mark):
        """the whole list, stopping at every recorded mark to call on_mark(payload) -- everything recorded before the
        mark has been enqueued by then, nothing after it, exactly as when the step ran from Python."""
        pos, waiting = 0, []
        for at, payload, cut in self.marks:
            waiting.append(payload)
            if not cut:
                continue
            self.replay(first=pos, end=at)
            pos = at
            for w in waiting:
                on_mark(w)
            waiting = []
        self.replay(first=pos, end=self.n)
        for w in waiting:
            on_mark(w)
