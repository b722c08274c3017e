// LSTM cell, the element-wise half (gfx950).  Reference: lib/models/lstm.py:16,25-27 (nn.LSTMCell with the state
// kept in the module).  The two GEMMs of a cell, gates = [x h] [W_ih W_hh]^T, run on the bf16 MFMA implicit-GEMM
// kernels as a 1x1x1 conv over N "positions" (capi.hip: vc_conv3d_fwd / dgrad / wgrad); what remains is bandwidth
// work on N x 4H values, fused into one pass each way:
//   forward : (i f g o) = gates + b_ih + b_hh;  c' = sig(f) c + sig(i) tanh(g);  h' = sig(o) tanh(c')
//             writes h', c' (fp32, the module's state), h' once more as bf16 (the next GEMM operand) and the four
//             activations + tanh(c') for the backward pass
//   backward: from dh', dc' -> dgates (bf16 for the two GEMMs of the backward pass), dc, and the bias gradient
//             (sum over the batch, fp32 atomics per workgroup: N is a batch size, 4H = 8192 columns)
// Gate order is torch's: i, f, g, o blocks of H columns each.
#include "common.h"
#include "elementwise.h"

namespace {

__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + __expf(-x)); }

__global__ __launch_bounds__(256) void lstm_pointwise_fwd_kernel(const bf16* __restrict__ gates, const float* __restrict__ b_ih,
                                                                 const float* __restrict__ b_hh, const float* __restrict__ c_prev,
                                                                 float* __restrict__ h, float* __restrict__ c,
                                                                 bf16* __restrict__ h_bf16, float* __restrict__ saved, int N,
                                                                 int H) {
    const long total = (long)N * H;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int n = (int)(e / H), j = (int)(e - (long)n * H);
        const bf16* g = gates + (size_t)n * 4 * H;
        float a[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) a[q] = bf2f(g[q * H + j]) + b_ih[q * H + j] + b_hh[q * H + j];
        const float ig = sigm(a[0]), fg = sigm(a[1]), gg = tanhf(a[2]), og = sigm(a[3]);
        const float cn = fg * c_prev[e] + ig * gg;
        const float tc = tanhf(cn);
        const float hn = og * tc;
        c[e] = cn;
        h[e] = hn;
        if (h_bf16 != nullptr) h_bf16[e] = f2bf(hn);
        if (saved != nullptr) {
            float* s = saved + (size_t)n * 5 * H;
            s[j] = ig, s[H + j] = fg, s[2 * H + j] = gg, s[3 * H + j] = og, s[4 * H + j] = tc;
        }
    }
}

// one block handles column j-range x ALL rows, so the bias gradient needs no atomics: thread = one column of one gate
// block; the loop over n is short (batch size)
__global__ __launch_bounds__(256) void lstm_pointwise_bwd_kernel(const float* __restrict__ dh, const float* __restrict__ dc_next,
                                                                 const float* __restrict__ saved, const float* __restrict__ c_prev,
                                                                 bf16* __restrict__ dgates, float* __restrict__ dc_prev,
                                                                 float* __restrict__ db_ih, float* __restrict__ db_hh, int N,
                                                                 int H) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= H) return;
    float sb[4] = {0.f, 0.f, 0.f, 0.f};
    for (int n = 0; n < N; ++n) {
        const float* s = saved + (size_t)n * 5 * H;
        const float ig = s[j], fg = s[H + j], gg = s[2 * H + j], og = s[3 * H + j], tc = s[4 * H + j];
        const size_t e = (size_t)n * H + j;
        const float dhn = dh != nullptr ? dh[e] : 0.f;
        const float dcn = (dc_next != nullptr ? dc_next[e] : 0.f) + dhn * og * (1.f - tc * tc);
        float d[4];
        d[0] = dcn * gg * ig * (1.f - ig);          // i
        d[1] = dcn * c_prev[e] * fg * (1.f - fg);   // f
        d[2] = dcn * ig * (1.f - gg * gg);          // g
        d[3] = dhn * tc * og * (1.f - og);          // o
        bf16* dg = dgates + (size_t)n * 4 * H;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            dg[q * H + j] = f2bf(d[q]);
            sb[q] += d[q];
        }
        if (dc_prev != nullptr) dc_prev[e] = dcn * fg;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (db_ih != nullptr) db_ih[q * H + j] += sb[q];
        if (db_hh != nullptr) db_hh[q * H + j] += sb[q];
    }
}

}  // namespace

int vc_ew_lstm_pointwise_fwd(const void* gates, const float* b_ih, const float* b_hh, const float* c_prev, float* h, float* c,
                             void* h_bf16, float* saved, int N, int H, hipStream_t s) {
    const long total = (long)N * H;
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    (void)hipGetLastError();
    hipLaunchKernelGGL(lstm_pointwise_fwd_kernel, dim3(grid), dim3(256), 0, s, (const bf16*)gates, b_ih, b_hh, c_prev, h, c,
                       (bf16*)h_bf16, saved, N, H);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        vc_set_error("lstm_pointwise_fwd launch failed: %s", hipGetErrorString(e));
        return -1;
    }
    return 0;
}

int vc_ew_lstm_pointwise_bwd(const float* dh, const float* dc_next, const float* saved, const float* c_prev, void* dgates,
                             float* dc_prev, float* db_ih, float* db_hh, int N, int H, hipStream_t s) {
    (void)hipGetLastError();
    hipLaunchKernelGGL(lstm_pointwise_bwd_kernel, dim3((H + 255) / 256), dim3(256), 0, s, dh, dc_next, saved, c_prev,
                       (bf16*)dgates, dc_prev, db_ih, db_hh, N, H);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        vc_set_error("lstm_pointwise_bwd launch failed: %s", hipGetErrorString(e));
        return -1;
    }
    return 0;
}
