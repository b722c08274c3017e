// Host-side check of the C ABI (include/vc3d.h) under AddressSanitizer + UndefinedBehaviorSanitizer, no GPU needed.
// Every conv / stem / pool / BatchNorm entry point is called for the layer geometries of R3D-18/34/50 at the BASELINE
// clip sizes (and a few ragged ones) with dummy device pointers: the planners, tap-table builders, split planners,
// sub-batch loops and argument checks all run on the host before the launch, which then fails (no device here) or is
// refused -- either way with a return code and a message, never a crash or a sanitizer report.
// Built and run by tests/test_host_cpu.py::test_c_abi_host_code_under_sanitizers (make -C csrc hostcheck).
#include <cstdio>
#include <cstring>
#include <set>
#include <string>
#include <vector>

#include "vc3d.h"

static std::set<std::string> g_msgs;
static long g_calls = 0, g_ok = 0;

static void note(int rc) {
    ++g_calls;
    if (rc == 0) {
        ++g_ok;
        return;
    }
    const char* m = vc_last_error();
    std::string s = m ? m : "(null)";
    if (s.size() > 48) s.resize(48);
    g_msgs.insert(s);
}

static void conv_all(int N, int Ci, int T, int H, int W, int Co, int k, int s, int p) {
    vc_conv3d_desc d = {N, Ci, T, H, W, Co, k, k, k, s, s, s, p, p, p};
    void* P = (void*)(uintptr_t)0x10000;  // never dereferenced on the host
    float* F = (float*)P;
    int32_t To, Ho, Wo;
    if (vc_conv3d_out_dims(&d, &To, &Ho, &Wo) != 0) {
        note(-1);
        return;
    }
    static uint32_t ticket_dummy;
    vc_bn_fin fin;
    memset(&fin, 0, sizeof(fin));
    fin.ticket = (uint32_t*)P, fin.gamma = F, fin.beta = F, fin.running_mean = F, fin.running_var = F, fin.params = F;
    fin.count = (float)N * To * Ho * Wo, fin.momentum = 0.1f, fin.eps = 1e-5f, fin.shift = F, fin.shift_next = F;
    (void)ticket_dummy;
    for (int with_ws = 0; with_ws < 2; ++with_ws) {
        const size_t w0 = with_ws ? vc_conv3d_workspace_bytes(&d, 0) : 0, w1 = with_ws ? vc_conv3d_workspace_bytes(&d, 1) : 0,
                     w2 = with_ws ? vc_conv3d_workspace_bytes(&d, 2) : 0;
        note(vc_conv3d_fwd(&d, P, P, P, F, w0 ? P : nullptr, w0, nullptr));
        note(vc_conv3d_fwd_bn(&d, P, P, P, F, &fin, w0 ? P : nullptr, w0, nullptr));
        note(vc_conv3d_fwd_act(&d, P, P, P, F, (void*)(uintptr_t)0x20000, 1, w0 ? P : nullptr, w0, nullptr));
        note(vc_conv3d_fwd_act(&d, P, P, P, F, nullptr, 0, w0 ? P : nullptr, w0, nullptr));
        note(vc_conv3d_dgrad(&d, P, P, P, 0, w1 ? P : nullptr, w1, nullptr));
        note(vc_conv3d_dgrad(&d, P, P, P, 1, w1 ? P : nullptr, w1, nullptr));
        {   // the fused form: reduction of the next BatchNorm backward and / or the masked accumulate
            vc_bn_bwd_red red = {P, F, (const uint8_t*)P, F, (uint32_t*)P, F, F, F};
            note(vc_conv3d_dgrad_fused(&d, P, P, P, 0, nullptr, &red, w1 ? P : nullptr, w1, nullptr));
            red.relu_bits = nullptr;  // mask mode 2
            note(vc_conv3d_dgrad_fused(&d, P, P, P, 1, (const uint8_t*)P, &red, w1 ? P : nullptr, w1, nullptr));
            note(vc_conv3d_dgrad_fused(&d, P, P, P, 1, (const uint8_t*)P, nullptr, w1 ? P : nullptr, w1, nullptr));
            note(vc_conv3d_dgrad_fused(&d, P, P, P, 0, (const uint8_t*)P, nullptr, w1 ? P : nullptr, w1, nullptr));  // refused
        }
        note(vc_conv3d_wgrad(&d, P, P, F, w2 ? P : nullptr, w2, nullptr));
    }
    note(vc_conv3d_fwd_noalloc(&d, P, P, P, F, nullptr));
    note(vc_pack_weight(F, P, P, Co, k * k * k, Ci, nullptr));
}

static void net(int depth, int N, int T, int HW) {
    void* P = (void*)(uintptr_t)0x10000;
    float* F = (float*)P;
    // stem + pool
    note(vc_stem_pack_input(P, 1, P, N, T, HW, HW, nullptr));
    note(vc_stem_fwd(P, P, P, F, N, T, HW, HW, 64, nullptr));
    const size_t sw = vc_stem_wgrad_workspace_bytes(N, T, HW, HW, 64);
    note(vc_stem_wgrad(P, P, F, N, T, HW, HW, 64, sw ? P : nullptr, sw, nullptr));
    note(vc_stem_wgrad(P, P, F, N, T, HW, HW, 64, nullptr, 0, nullptr));
    vc_pool3d_desc pd = {N, 64, T, HW / 2, HW / 2, 3, 3, 3, 2, 2, 2, 1, 1, 1};
    int32_t pt, ph, pw;
    note(vc_pool3d_out_dims(&pd, &pt, &ph, &pw));
    note(vc_maxpool3d_fwd(&pd, P, F, P, (uint8_t*)P, nullptr));
    note(vc_maxpool3d_bwd(&pd, P, (const uint8_t*)P, P, nullptr));
    note(vc_bn_bwd_reduce_pooled(&pd, P, P, (const uint8_t*)P, P, F, F, nullptr));
    note(vc_bn_bwd_apply_pooled(&pd, P, P, (const uint8_t*)P, F, F, P, nullptr));
    const bool bottleneck = depth >= 50;
    const int layers18[4] = {2, 2, 2, 2}, layers34[4] = {3, 4, 6, 3};
    const int* L = depth == 18 ? layers18 : layers34;
    int t = (T + 1) / 2, hw = (HW / 2 + 1) / 2, cin = 64;
    for (int st = 0; st < 4; ++st) {
        const int planes = 64 << st, stride = st == 0 ? 1 : 2, cout = planes * (bottleneck ? 4 : 1);
        for (int b = 0; b < L[st]; ++b) {
            const int s = b == 0 ? stride : 1;
            const int to = (t - 1) / s + 1, ho = (hw - 1) / s + 1;
            if (bottleneck) {
                conv_all(N, cin, t, hw, hw, planes, 1, 1, 0);
                conv_all(N, planes, t, hw, hw, planes, 3, s, 1);
                conv_all(N, planes, to, ho, ho, cout, 1, 1, 0);
            } else {
                conv_all(N, cin, t, hw, hw, planes, 3, s, 1);
                conv_all(N, planes, to, ho, ho, planes, 3, 1, 1);
            }
            if (b == 0 && (s != 1 || cin != cout)) conv_all(N, cin, t, hw, hw, cout, 1, s, 0);
            const long rows = (long)N * to * ho * ho;
            note(vc_bn_stats(P, F, F, rows, cout, nullptr));
            note(vc_bn_apply(P, F, P, P, rows * cout, cout, 1, nullptr));
            note(vc_bn_apply_mask(P, F, P, P, (uint8_t*)P, rows * cout, cout, nullptr));
            for (int mask = 0; mask < 4; ++mask) {
                note(vc_bn_bwd_reduce(P, P, P, F, F, rows, cout, mask, nullptr));
                note(vc_bn_bwd_apply(P, P, P, F, F, P, P, rows * cout, cout, mask, nullptr));
            }
            cin = cout, t = to, hw = ho;
        }
    }
    note(vc_sgd_step(F, F, F, P, 1000003, 0.1f, 0.9f, 1e-4f, 1, nullptr));
    note(vc_adam_step(F, F, F, F, P, 1000003, 1e-3f, 0.9f, 0.999f, 1e-8f, 0.f, 7, nullptr));
    note(vc_adam_step(F, F, F, F, P, 16, 1e-3f, 1.0f, 0.999f, 1e-8f, 0.f, 1, nullptr));  // refused: beta1 = 1
    note(vc_gavgpool_fwd(P, F, N, t * hw * hw, cin, nullptr));
    note(vc_linear_fwd(F, F, F, F, N, cin, 400, nullptr));
}

// Command lists (vc_replay): the generic call must hand integer AND float arguments to the entry point exactly as a
// direct call does.  Without a device both forms fail at the launch, after the argument checks: the same return code
// and the same message show that every argument arrived in its place (a misplaced float would trip "bad arguments"
// or change the refusal).  Malformed lists are refused with the index of the offending command.
static int replay_checks() {
    void* P = (void*)(uintptr_t)0x10000;
    float* F = (float*)P;
    auto msg = []() { const char* m = vc_last_error(); return std::string(m ? m : ""); };
    // direct: floats interleaved with pointers (count, momentum, eps), 10 integer-class arguments (4 of them on the stack)
    const int rc_direct = vc_bn_finalize(F, 64, 128.f, F, F, nullptr, nullptr, 0.1f, 1e-5f, F, nullptr, nullptr, nullptr);
    const std::string m_direct = msg();
    vc_cmd c[4];
    memset(c, 0, sizeof(c));
    c[0].fn = (void*)&vc_set_deterministic, c[0].kind = VC_CMD_CALL, c[0].nint = 1, c[0].iarg[0] = 1;
    c[1].fn = (void*)&vc_set_deterministic, c[1].kind = VC_CMD_CALL, c[1].nint = 1, c[1].iarg[0] = 0;
    c[2].fn = (void*)&vc_bn_finalize, c[2].kind = VC_CMD_CALL, c[2].nint = 10, c[2].nflt = 3;
    const int64_t ia[10] = {(int64_t)(uintptr_t)F, 64, (int64_t)(uintptr_t)F, (int64_t)(uintptr_t)F, 0, 0, (int64_t)(uintptr_t)F, 0, 0, 0};
    for (int i = 0; i < 10; ++i) c[2].iarg[i] = ia[i];
    c[2].farg[0] = 128.f, c[2].farg[1] = 0.1f, c[2].farg[2] = 1e-5f;
    c[3] = c[0];
    int32_t at = -7;
    const int rc = vc_replay(c, 4, nullptr, 0, -1, nullptr, nullptr, &at);
    if (rc_direct == 0 || rc != rc_direct || at != 2 || msg() != m_direct) {
        printf("replay: direct rc %d '%s' vs replayed rc %d at %d '%s'\n", rc_direct, m_direct.c_str(), rc, at, msg().c_str());
        return 1;
    }
    if (vc_get_deterministic() != 0) return 1;  // commands 0 and 1 ran, command 3 (after the failure) did not
    // a wrong float would be seen: count <= 0 is refused before the launch with another message
    c[2].farg[0] = 0.f;
    if (vc_replay(c, 3, nullptr, 0, -1, nullptr, nullptr, &at) != -2 || at != 2 || msg() == m_direct) return 1;
    // malformed lists
    vc_cmd bad[1];
    memset(bad, 0, sizeof(bad));
    bad[0].kind = VC_CMD_CALL;  // null function
    if (vc_replay(bad, 1, nullptr, 0, -1, nullptr, nullptr, &at) != -2 || at != 0) return 1;
    bad[0].fn = (void*)&vc_set_deterministic, bad[0].nint = VC_CMD_MAX_INT + 1;
    if (vc_replay(bad, 1, nullptr, 0, -1, nullptr, nullptr, &at) != -2 || at != 0) return 1;
    bad[0].nint = 0, bad[0].kind = 77;  // unknown kind
    if (vc_replay(bad, 1, nullptr, 0, -1, nullptr, nullptr, &at) != -2 || at != 0) return 1;
    bad[0].kind = VC_CMD_EVENT_RECORD, bad[0].iarg[0] = 3;  // event 3 of 0
    if (vc_replay(bad, 1, nullptr, 0, -1, nullptr, nullptr, &at) != -2 || at != 0) return 1;
    if (vc_replay(nullptr, 0, nullptr, 0, -1, nullptr, nullptr, &at) != 0 || at != -1) return 1;  // the empty list
    if (vc_replay(nullptr, 2, nullptr, 0, -1, nullptr, nullptr, &at) != -2) return 1;
    if (vc_memset_async(nullptr, 0, 16, nullptr) != -2 || vc_memset_async(nullptr, 0, 0, nullptr) != 0) return 1;
    if (vc_incr_i64(nullptr, 4, nullptr) != -2 || vc_incr_i64(nullptr, 0, nullptr) != 0) return 1;
    return 0;
}

int main() {
    if (vc_abi_version() != VC_ABI_VERSION) return 2;
    if (replay_checks() != 0) return 3;
    const int depths[3] = {18, 34, 50};
    for (int depth : depths) {
        net(depth, 64, 16, 112);  // BASELINE configs 1-3
        net(depth, 2, 8, 64);     // the small test clip
        net(depth, 3, 5, 30);     // ragged: odd T, sizes that do not divide
    }
    net(50, 64, 64, 224);  // config 4 at batch 64: tensors beyond 2^31 elements, sub-batch loops
    net(50, 1, 1, 224);    // the 2D-reference geometry (T = 1): dead temporal taps
    // refusals
    conv_all(1, 48, 2, 4, 4, 64, 3, 1, 1);     // Ci not a multiple of 64
    conv_all(1, 64, 2, 4, 4, 64, 5, 1, 2);     // 125 taps
    conv_all(4096, 2048, 64, 224, 224, 2048, 3, 1, 1);  // one sample beyond 2^31
    conv_all(0, 64, 2, 4, 4, 64, 3, 1, 1);
    conv_all(1, 64, 1, 1, 1, 64, 3, 2, 0);     // kernel larger than the padded input
    printf("calls %ld, returned 0: %ld, distinct messages %zu\n", g_calls, g_ok, g_msgs.size());
    for (const auto& m : g_msgs) printf("  %s\n", m.c_str());
    return 0;
}
