// Replay of a recorded launch sequence (include/vc3d.h, "Command lists").
//
// The reference enqueues a training step from Python, one torch call per layer (lib/core/function.py:71-82).  The
// host side of this build's step is ~200 calls through this C ABI (R3D-50: ~410): 17 us each from Python, 3.4 ms per
// R3D-18 step.  The sequence of a step is the same every step -- same entry points, same descriptors, same device
// pointers (the engine keeps every intermediate tensor of a recorded step alive) -- so it is recorded once as an
// array of vc_cmd and replayed by ONE call: ~4 us per launch, all of it inside hipLaunchKernel.
// HIP graphs were measured first: the captured two-stream backward replays 5 % slower on the GPU
// than the eager queues; a command list issues exactly the eager sequence.
#include <stdio.h>

#include "../../include/vc3d.h"
#include "common.h"

namespace {

__global__ void incr_i64_kernel(long long* const* ptrs, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) *ptrs[i] += 1;
}

// Every entry point of this library takes integer / pointer arguments and at most 8 floats and returns int.  Under the
// x86-64 System V calling convention the two register classes are assigned independently (integers: rdi, rsi, rdx,
// rcx, r8, r9, then the stack in order; floats: xmm0-7), so ONE prototype with all integer slots first and all float
// slots after them calls any of them: integer argument i of the callee is integer slot i here whatever floats sit
// between them in its own signature, and surplus slots are ignored by the callee.
typedef int (*vc_generic_fn)(int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, float, float, float, float, float, float,
                             float, float, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t,
                             int64_t);
#if !defined(__x86_64__)
#error "vc_replay's generic call relies on the x86-64 System V calling convention"
#endif

}  // namespace

extern "C" {

int vc_memset_async(void* ptr, int32_t value, size_t bytes, void* stream) {
    if (bytes == 0) return 0;
    if (ptr == nullptr) {
        vc_set_error("vc_memset_async: null pointer");
        return -2;
    }
    const hipError_t e = hipMemsetAsync(ptr, value, bytes, (hipStream_t)stream);
    if (e != hipSuccess) {
        vc_set_error("vc_memset_async: %s", hipGetErrorString(e));
        return -1;
    }
    return 0;
}

int vc_incr_i64(const void* ptr_table, int32_t n, void* stream) {
    if (n <= 0) return 0;
    if (ptr_table == nullptr) {
        vc_set_error("vc_incr_i64: null table");
        return -2;
    }
    (void)hipGetLastError();
    hipLaunchKernelGGL(incr_i64_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, (long long* const*)ptr_table, n);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        vc_set_error("vc_incr_i64: %s", hipGetErrorString(e));
        return -1;
    }
    return 0;
}

int vc_replay_events_create(int32_t n, int32_t timing, void** out) {
    if (n < 0 || (n > 0 && out == nullptr)) {
        vc_set_error("vc_replay_events_create: bad arguments");
        return -2;
    }
    for (int i = 0; i < n; ++i) {
        hipEvent_t ev;
        const hipError_t e = hipEventCreateWithFlags(&ev, timing ? hipEventDefault : hipEventDisableTiming);
        if (e != hipSuccess) {
            for (int j = 0; j < i; ++j) (void)hipEventDestroy((hipEvent_t)out[j]);
            vc_set_error("vc_replay_events_create: %s", hipGetErrorString(e));
            return -1;
        }
        out[i] = (void*)ev;
    }
    return 0;
}

int vc_replay_events_destroy(int32_t n, void** events) {
    for (int i = 0; i < n; ++i)
        if (events[i] != nullptr) (void)hipEventDestroy((hipEvent_t)events[i]), events[i] = nullptr;
    return 0;
}

int vc_replay_event_elapsed_ms(void* start, void* stop, float* ms) {
    const hipError_t e = hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop);
    if (e != hipSuccess) {
        vc_set_error("vc_replay_event_elapsed_ms: %s", hipGetErrorString(e));
        return -1;
    }
    return 0;
}

// (no_sanitize("function"): calling through vc_generic_fn is outside the C++ standard by design -- it is the documented
// behaviour of the platform ABI named above that this relies on; UBSan's function-type check would flag every command)
__attribute__((no_sanitize("function")))
int vc_replay(const vc_cmd* cmds, int32_t n, void* const* events, int32_t n_events, int32_t time_cmd, void* time_start,
              void* time_stop, int32_t* failed_at) {
    if (failed_at != nullptr) *failed_at = -1;
    if (n < 0 || (n > 0 && cmds == nullptr)) {
        vc_set_error("vc_replay: bad arguments");
        return -2;
    }
    for (int i = 0; i < n; ++i) {
        const vc_cmd& c = cmds[i];
        int rc = 0;
        if (c.kind == VC_CMD_CALL) {
            if (c.fn == nullptr || c.nint < 0 || c.nint > VC_CMD_MAX_INT || c.nflt < 0 || c.nflt > VC_CMD_MAX_FLT) {
                vc_set_error("vc_replay: malformed command %d", i);
                rc = -2;
            } else {
                // the stream of an entry point is its last integer argument (include/vc3d.h: "launches on `stream`")
                const hipStream_t st = c.nint > 0 ? (hipStream_t)c.iarg[c.nint - 1] : nullptr;
                if (i == time_cmd && time_start != nullptr) (void)hipEventRecord((hipEvent_t)time_start, st);
                const int64_t* a = c.iarg;
                const float* f = c.farg;
                rc = ((vc_generic_fn)c.fn)(a[0], a[1], a[2], a[3], a[4], a[5], f[0], f[1], f[2], f[3], f[4], f[5], f[6], f[7],
                                           a[6], a[7], a[8], a[9], a[10], a[11], a[12], a[13], a[14], a[15]);
                if (i == time_cmd && time_stop != nullptr) (void)hipEventRecord((hipEvent_t)time_stop, st);
            }
        } else if (c.kind == VC_CMD_EVENT_RECORD || c.kind == VC_CMD_STREAM_WAIT) {
            const int64_t ev = c.iarg[0];
            if (ev < 0 || ev >= n_events || events == nullptr) {
                vc_set_error("vc_replay: command %d names event %lld of %d", i, (long long)ev, n_events);
                rc = -2;
            } else {
                const hipError_t e = c.kind == VC_CMD_EVENT_RECORD
                                         ? hipEventRecord((hipEvent_t)events[ev], (hipStream_t)c.iarg[1])
                                         : hipStreamWaitEvent((hipStream_t)c.iarg[1], (hipEvent_t)events[ev], 0);
                if (e != hipSuccess) {
                    vc_set_error("vc_replay: event command %d: %s", i, hipGetErrorString(e));
                    rc = -1;
                }
            }
        } else {
            vc_set_error("vc_replay: unknown command kind %d at %d", c.kind, i);
            rc = -2;
        }
        if (rc != 0) {
            if (failed_at != nullptr) *failed_at = i;
            return rc;  // the failing entry point's message stays in vc_last_error
        }
    }
    return 0;
}

}  // extern "C"
