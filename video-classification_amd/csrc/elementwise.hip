// Bandwidth-bound kernels of the 3D-ResNet path (gfx950): BatchNorm3d statistics /
// apply / backward, ReLU and residual add, 3D max / average pooling, the classifier
// head, layout packing and weight preparation.  Activations are channels-last (NTHWC)
// bf16, so every kernel moves 16 bytes (8 channels) per lane and indexes its per-channel
// parameters by (element index mod C).
#include "common.h"
#include "bn_fin.h"
#include "elementwise.h"

namespace {

__device__ __forceinline__ void unpack8(const bf16x8& v, float (&f)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = bf2f(v[i]);
}
__device__ __forceinline__ bf16x8 pack8(const float (&f)[8]) {
    bf16x8 v;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = f2bf(f[i]);
    return v;
}

// ------------------------------------------------------------------ BN finalize
// stats [R][2][C] (sum, sumsq) -> params [4][C] = mean, invstd, scale, shift; running stats.
__global__ void bn_finalize_kernel(const float* __restrict__ stats, int C, float count, const float* gamma,
                                   const float* beta, float* running_mean, float* running_var, float momentum,
                                   float eps, float* params, const float* shift, float* shift_out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    vc_bn_finalize_channel<false>(stats, C, c, count, gamma, beta, running_mean, running_var, momentum, eps, params, shift,
                                  shift_out);
}

// eval mode: params from running statistics
__global__ void bn_eval_params_kernel(int C, const float* gamma, const float* beta, const float* running_mean,
                                      const float* running_var, float eps, float* params) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float invstd = rsqrtf(running_var[c] + eps);
    const float scale = gamma[c] * invstd;
    params[c] = running_mean[c];
    params[C + c] = invstd;
    params[2 * C + c] = scale;
    params[3 * C + c] = beta[c] - running_mean[c] * scale;
}

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

// ------------------------------------------------------------------ BN apply (+res)(+relu)
// BITS: also write one byte per 8 channels, bit k = (out[k] > 0) -- the ReLU mask of a residual block's output for
// the backward passes (MASK 3 below), which then read 1 bit per element instead of the 16-bit activation.
template <bool RES, bool RELU, bool BITS = false>
__global__ __launch_bounds__(256) void bn_apply_kernel(const bf16x8* __restrict__ y, const float* __restrict__ params,
                                                       const bf16x8* __restrict__ res, bf16x8* __restrict__ out,
                                                       long n8, int C, uint8_t* __restrict__ bits = nullptr) {
    const float* scale = params + 2 * C;
    const float* shift = params + 3 * C;
    const long stride = (long)gridDim.x * blockDim.x;
    // the stride is a multiple of the channel groups per row whenever C/8 divides 256 (every ResNet
    // width): then a thread keeps its 8 channels for the whole loop and loads their parameters once
    const bool fixed = ((stride * 8) % C) == 0;
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    int c = (int)((i * 8) % C);
    f32x4 s0 = *(const f32x4*)(scale + c), s1 = *(const f32x4*)(scale + c + 4);
    f32x4 b0 = *(const f32x4*)(shift + c), b1 = *(const f32x4*)(shift + c + 4);
    for (; i < n8; i += stride) {
        if (!fixed) {
            c = (int)((i * 8) % C);
            s0 = *(const f32x4*)(scale + c), s1 = *(const f32x4*)(scale + c + 4);
            b0 = *(const f32x4*)(shift + c), b1 = *(const f32x4*)(shift + c + 4);
        }
        float f[8], r[8];
        unpack8(y[i], f);
        if (RES) unpack8(res[i], r);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float v = f[k] * (k < 4 ? s0[k] : s1[k - 4]) + (k < 4 ? b0[k] : b1[k - 4]);
            if (RES) v += r[k];
            if (RELU) v = v > 0.f ? v : 0.f;
            f[k] = v;
        }
        const bf16x8 o = pack8(f);
        out[i] = o;
        if (BITS) {  // from the rounded values: the mask is exactly (out > 0)
            const u32x4 w = __builtin_bit_cast(u32x4, o);
            unsigned b = 0;
#pragma unroll
            for (int m = 0; m < 4; ++m)
                b |= ((w[m] & 0xffffu) != 0u ? 1u << (2 * m) : 0u) | ((w[m] >> 16) != 0u ? 2u << (2 * m) : 0u);
            bits[i] = (uint8_t)b;
        }
    }
}

// ------------------------------------------------------------------ per-channel reductions
// MODE 0: sum y, sum y^2 (standalone BN statistics)
// MODE 1: BN backward sums: sum g, sum g*xhat with g = dz * mask
//   MASK 0 none, 1 mask = (aux > 0) with aux the saved post-activation output,
//        2 mask = (y*scale+shift > 0) recomputed,
//        3 mask = bit k of byte aux[i] (written by bn_apply_kernel<.., BITS>: the same mask as 1 at 1/16 of the bytes)
// A block owns a 64-channel slice (8 lanes x 16 bytes = one 128-byte piece of every row it visits) and every RS-th
// group of 32 rows; it reduces its partial sums through LDS and adds them to one of VC_STAT_REPLICAS copies of
// sums[2][C].  The number of float atomics of a launch is RS x 2C whatever the channel split, so wide layers get
// their parallelism from channel slices (C / 64 of them) and a small RS: with whole rows per block the 2-3 million
// device-scope atomics of a C >= 512 layer took 90-120 us for a few MB of input (R3D-50 layer3/4).
// y = (z - shift) / scale carries z's bf16 rounding, |z / gamma| * 2^-9 in xhat: fine unless |beta| > 16 |gamma| (or gamma == 0)
__device__ __forceinline__ bool vc_pooled_z_ok(float mean, float invstd, float scale, float shift) {
    const float beta = shift + mean * scale;  // scale = gamma * invstd
    return scale != 0.f && fabsf(beta) * invstd <= 16.f * fabsf(scale);
}

template <int MODE, int MASK>
__global__ __launch_bounds__(256) void chan_reduce_kernel(const bf16x8* __restrict__ y, const bf16x8* __restrict__ dz,
                                                          const bf16x8* __restrict__ aux,
                                                          const float* __restrict__ params, float* sums, long rows,
                                                          int C, int CS, VcBnFin ffin, VcBnBwdFin bfin,
                                                          float* shift_out) {
    __shared__ float sm[256 * 17];
    const int groups = C >> 3;             // 8-channel groups per row
    const int cs = blockIdx.x % CS;        // channel slice: groups [8cs, 8cs + 8)
    const int rs = blockIdx.x / CS;        // row phase
    const int RS = gridDim.x / CS;
    const int g = cs * 8 + (threadIdx.x & 7);
    const int r = threadIdx.x >> 3;        // row lane 0..31
    float a0[8], a1[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) a0[k] = a1[k] = 0.f;
    float mean[8], invstd[8], scale[8], shift[8];
    float sh0[8];  // MODE 0 with shift_out: the sums are taken relative to row 0 of y (within a few sigma of the mean
                   // of any channel, and known to every block without a pass of its own); written out for the finalize
#pragma unroll
    for (int k = 0; k < 8; ++k) sh0[k] = 0.f;
    if (MODE == 0 && shift_out != nullptr) {
        unpack8(y[g], sh0);
        if (rs == 0 && r == 0) {
#pragma unroll
            for (int k = 0; k < 8; ++k) shift_out[g * 8 + k] = sh0[k];
        }
    }
    if (MODE == 1) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            mean[k] = params[g * 8 + k];
            invstd[k] = params[C + g * 8 + k];
            scale[k] = params[2 * C + g * 8 + k];
            shift[k] = params[3 * C + g * 8 + k];
        }
    }
    // MASK 4: `y` holds z = relu(scale * y + shift) rounded to bf16 (the max pool's output at its arg-max) and y is
    // solved from it, z > 0 being the ReLU mask; channels where that is ill-conditioned (vc_pooled_z_ok) add nothing
    // here and are taken by bn_bwd_reduce_pooled_kernel's gather in a second launch.
    float rscale[8];
    unsigned zokm = 0u;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        rscale[k] = 0.f;
        if (MASK == 4 && vc_pooled_z_ok(mean[k], invstd[k], scale[k], shift[k])) rscale[k] = 1.f / scale[k], zokm |= 1u << k;
    }
    for (long row = (long)rs * 32 + r; row < rows; row += (long)RS * 32) {
        const long i = row * groups + g;
        float f[8];
        unpack8(y[i], f);
        if (MODE == 0) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float u = f[k] - sh0[k];
                a0[k] += u;
                a1[k] += u * u;
            }
        } else {
            float d[8], m[8];
            unpack8(dz[i], d);
            if (MASK == 1) unpack8(aux[i], m);
            const unsigned mb = MASK == 3 ? ((const uint8_t*)aux)[i] : 0u;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                float gk = d[k];
                if (MASK == 1) gk = m[k] > 0.f ? gk : 0.f;
                if (MASK == 2) gk = (f[k] * scale[k] + shift[k]) > 0.f ? gk : 0.f;
                if (MASK == 3) gk = ((mb >> k) & 1u) ? gk : 0.f;
                if (MASK == 4) {
                    gk = (f[k] > 0.f && ((zokm >> k) & 1u)) ? gk : 0.f;
                    f[k] = (f[k] - shift[k]) * rscale[k];
                }
                a0[k] += gk;
                a1[k] += gk * (f[k] - mean[k]) * invstd[k];
            }
        }
    }
    // block reduce over the 32 row lanes: thread (k, gg) sums value k of channel group gg
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        sm[threadIdx.x * 17 + k] = a0[k];
        sm[threadIdx.x * 17 + 8 + k] = a1[k];
    }
    __syncthreads();
    if (threadIdx.x < 128) {
        const int gg = threadIdx.x & 7, k = threadIdx.x >> 3;
        float s = 0.f;
#pragma unroll 8
        for (int rr = 0; rr < 32; ++rr) s += sm[(rr * 8 + gg) * 17 + k];
        vc_stat_add(sums, (unsigned)rs, C, (k < 8 ? 0 : C) + (cs * 8 + gg) * 8 + (k & 7), s);
    }
    if (MODE == 0) vc_bn_fin_last(ffin, sums, C);
    else vc_bn_bwd_fin_last(bfin, sums, C);
}

// ------------------------------------------------------------------ split-K finalize
// sum the fp32 slabs of a split-K conv -> bf16 activation (optionally += the existing value)
// and, optionally, the BatchNorm sum / sum-of-squares of the unrounded result.
// STATS 0: nothing more; 1: BatchNorm sums of the result (forward); 2 / 3: the BatchNorm BACKWARD sums (sum g, sum g*xhat,
// chan_reduce_kernel MODE 1) of the BatchNorm that consumes the finished gradient next, with the ReLU mask recomputed
// from its y (2) or read from relu bits (3) -- the small-plane data gradients (layers 3-4) are chains of short
// launches, and this one replaces finalize + reduce (6 + 12 us at 1 k rows) by one pass.
struct VcFinRed {
    const bf16x8* y;
    const float* params;
    const uint8_t* bits;
    VcBnBwdFin fin;
};
template <int STATS>
__global__ __launch_bounds__(256) void splitk_finalize_kernel(const float* __restrict__ slabs, int splits,
                                                              long long stride, bf16x8* __restrict__ out,
                                                              int accumulate, float* stats, long rows, int C,
                                                              int CS, VcBnFin fin, const float* stat_shift, VcFinRed red) {
    // same decomposition as chan_reduce_kernel: 64-channel slice x row phase (float atomics per launch = RS x 2C)
    __shared__ float sm[STATS ? 256 * 17 : 1];
    const int groups = C >> 3;
    const int cs = blockIdx.x % CS, rs = blockIdx.x / CS, RS = gridDim.x / CS;
    const int g = cs * 8 + (threadIdx.x & 7);
    const int r = threadIdx.x >> 3;
    float a0[8], a1[8], sh0[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        a0[k] = a1[k] = 0.f;
        sh0[k] = (STATS == 1 && stat_shift != nullptr) ? stat_shift[g * 8 + k] : 0.f;  // shifted sums (bn_fin.h)
    }
    float mean[8], invstd[8], scale[8], shift[8];
    if (STATS >= 2) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            mean[k] = red.params[g * 8 + k];
            invstd[k] = red.params[C + g * 8 + k];
            scale[k] = red.params[2 * C + g * 8 + k];
            shift[k] = red.params[3 * C + g * 8 + k];
        }
    }
    for (long row = (long)rs * 32 + r; row < rows; row += (long)RS * 32) {
        const long i = row * groups + g;
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = 0.f;
        // four slabs' loads in flight at a time, summed in slab order (one at a time is a chain of `splits` memory
        // latencies per row: 15 us for an 8-slab layer whose slabs total 16 MB)
        int z = 0;
        for (; z + 4 <= splits; z += 4) {
            f32x4 lo[4], hi[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const f32x4* src = (const f32x4*)(slabs + (size_t)(z + u) * stride + i * 8);
                lo[u] = src[0], hi[u] = src[1];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    v[k] += lo[u][k];
                    v[4 + k] += hi[u][k];
                }
        }
        for (; z < splits; ++z) {
            const f32x4* src = (const f32x4*)(slabs + (size_t)z * stride + i * 8);
            const f32x4 lo = src[0], hi = src[1];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                v[k] += lo[k];
                v[4 + k] += hi[k];
            }
        }
        if (accumulate) {
            float o[8];
            unpack8(out[i], o);
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] += o[k];
        }
        const bf16x8 ov = pack8(v);
        out[i] = ov;
        if (STATS == 1) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float u = v[k] - sh0[k];
                a0[k] += u;
                a1[k] += u * u;
            }
        }
        if (STATS >= 2) {  // over the ROUNDED gradient: what the separate reduction reads back
            float d[8], f[8];
            unpack8(ov, d);
            unpack8(red.y[i], f);
            const unsigned mb = STATS == 3 ? red.bits[i] : 0u;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                float gk = d[k];
                if (STATS == 2) gk = (f[k] * scale[k] + shift[k]) > 0.f ? gk : 0.f;
                if (STATS == 3) gk = ((mb >> k) & 1u) ? gk : 0.f;
                a0[k] += gk;
                a1[k] += gk * (f[k] - mean[k]) * invstd[k];
            }
        }
    }
    if (STATS) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            sm[threadIdx.x * 17 + k] = a0[k];
            sm[threadIdx.x * 17 + 8 + k] = a1[k];
        }
        __syncthreads();
        if (threadIdx.x < 128) {
            const int gg = threadIdx.x & 7, k = threadIdx.x >> 3;
            float s = 0.f;
#pragma unroll 8
            for (int rr = 0; rr < 32; ++rr) s += sm[(rr * 8 + gg) * 17 + k];
            vc_stat_add(stats, (unsigned)rs, C, (k < 8 ? 0 : C) + (cs * 8 + gg) * 8 + (k & 7), s);
        }
        if (STATS == 1) vc_bn_fin_last(fin, stats, C);
        else vc_bn_bwd_fin_last(red.fin, stats, C);
    }
}

// dw += sum over slabs (deterministic split-K reduction of the weight gradient)
// (256 / PH) float4 positions x PH slab phases per block: phase q sums slabs q, q + PH, ... (independent loads in
// flight), the PH partial sums are combined through LDS in a fixed order (deterministic).  PH = 16 for the many-slab
// case (a 1x1x1 conv's 256 x 64 gradient comes as 512 slabs: with 4 phases a thread walked 128 of them in a
// dependent chain, 19 us for 64 KB of output), PH = 4 otherwise.
template <int PH>
__global__ __launch_bounds__(256) void splitk_reduce_wgrad_kernel(const float* __restrict__ slabs, int splits,
                                                                  long long stride, float* __restrict__ dw, long n4,
                                                                  int ci4, int taps, unsigned live) {
    constexpr int NP = 256 / PH;
    __shared__ f32x4 part[PH - 1][NP];
    const int q = threadIdx.x / NP, l = threadIdx.x % NP;
    const long i = (long)blockIdx.x * NP + l;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    // `live` (bit t: tap t of [Co][taps][Ci] was computed): the weight-gradient kernels never write the slab entries
    // of a dropped tap (one that only ever reads zero padding), so those are neither read nor added: dw stays as it is
    bool on = i < n4;
    if (on && live != 0xffffffffu) on = ((live >> (unsigned)((i / ci4) % taps)) & 1u) != 0u;
    if (on) {
        // four independent chains per thread: the walk is a chain of memory latencies, not of bytes
        f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = s1, s3 = s1;
        int z = q;
        for (; z + 3 * PH < splits; z += 4 * PH) {
            const f32x4 v0 = ((const f32x4*)(slabs + (size_t)z * stride))[i];
            const f32x4 v1 = ((const f32x4*)(slabs + (size_t)(z + PH) * stride))[i];
            const f32x4 v2 = ((const f32x4*)(slabs + (size_t)(z + 2 * PH) * stride))[i];
            const f32x4 v3 = ((const f32x4*)(slabs + (size_t)(z + 3 * PH) * stride))[i];
            s += v0, s1 += v1, s2 += v2, s3 += v3;
        }
        for (; z < splits; z += PH) s += ((const f32x4*)(slabs + (size_t)z * stride))[i];
        s += s1;
        s2 += s3;
        s += s2;
    }
    if (q > 0) part[q - 1][l] = s;
    __syncthreads();
    if (q == 0 && on) {
#pragma unroll
        for (int k = 0; k < PH - 1; ++k) s += part[k][l];
        f32x4 o = ((const f32x4*)dw)[i];
        o += s;
        ((f32x4*)dw)[i] = o;
    }
}

// BN backward finalize: sums [R][2][C] -> coef [2][C] = (sum g)/n, (sum g xhat)/n ; dgamma += sum g xhat ; dbeta += sum g
__global__ void bn_bwd_finalize_kernel(const float* __restrict__ sums, int C, float count, float* coef, float* dgamma,
                                       float* dbeta) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    vc_bn_bwd_finalize_channel<false>(sums, C, c, count, coef, dgamma, dbeta);
}

// BN backward apply: dy = scale * (g - c1 - xhat*c2); optionally writes g (the masked
// incoming gradient, which is also the gradient of the residual branch).
struct BnBwdCoef {  // per-lane constants of 8 consecutive channels
    float mean[8], invstd[8], scale[8], shift[8], c1[8], c2[8];
};
__device__ __forceinline__ void load_bwd_coef(BnBwdCoef& k, const float* params, const float* coef, int C, int c) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const f32x4 m = *(const f32x4*)(params + c + 4 * h), iv = *(const f32x4*)(params + C + c + 4 * h),
                    sc = *(const f32x4*)(params + 2 * C + c + 4 * h), sh = *(const f32x4*)(params + 3 * C + c + 4 * h),
                    a = *(const f32x4*)(coef + c + 4 * h), b = *(const f32x4*)(coef + C + c + 4 * h);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            k.mean[4 * h + j] = m[j], k.invstd[4 * h + j] = iv[j], k.scale[4 * h + j] = sc[j];
            k.shift[4 * h + j] = sh[j], k.c1[4 * h + j] = a[j], k.c2[4 * h + j] = b[j];
        }
    }
}

template <int MASK, bool WRITE_G>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const bf16x8* __restrict__ y, const bf16x8* __restrict__ dz,
                                                           const bf16x8* __restrict__ aux,
                                                           const float* __restrict__ params,
                                                           const float* __restrict__ coef, bf16x8* __restrict__ dy,
                                                           bf16x8* __restrict__ gout, long n8, int C) {
    const long stride = (long)gridDim.x * blockDim.x;
    const bool fixed = ((stride * 8) % C) == 0;  // see bn_apply_kernel
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    BnBwdCoef kc;
    load_bwd_coef(kc, params, coef, C, (int)((i * 8) % C));
    for (; i < n8; i += stride) {
        if (!fixed) load_bwd_coef(kc, params, coef, C, (int)((i * 8) % C));
        float f[8], d[8], m[8], o[8];
        unpack8(y[i], f);
        unpack8(dz[i], d);
        if (MASK == 1) unpack8(aux[i], m);
        const unsigned mb = MASK == 3 ? ((const uint8_t*)aux)[i] : 0u;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float gk = d[k];
            if (MASK == 1) gk = m[k] > 0.f ? gk : 0.f;
            if (MASK == 2) gk = (f[k] * kc.scale[k] + kc.shift[k]) > 0.f ? gk : 0.f;
            if (MASK == 3) gk = ((mb >> k) & 1u) ? gk : 0.f;
            const float xhat = (f[k] - kc.mean[k]) * kc.invstd[k];
            o[k] = kc.scale[k] * (gk - kc.c1[k] - xhat * kc.c2[k]);
            d[k] = gk;
        }
        dy[i] = pack8(o);
        if (WRITE_G) gout[i] = pack8(d);
    }
}

// ------------------------------------------------------------------ add (bf16): out = a + b
// x = x where its ReLU bit is set, else 0 (one byte of bits per 16-byte piece): the general form of the masked
// accumulate of vc_conv3d_dgrad_fused, for geometries whose conv kernel does not apply the mask itself
__global__ __launch_bounds__(256) void mask_bits_kernel(bf16x8* __restrict__ x, const uint8_t* __restrict__ bits, long n8) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
        bf16x8 v = x[i];
        const unsigned mb = bits[i];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = ((mb >> k) & 1u) ? v[k] : (bf16)0;
        x[i] = v;
    }
}

__global__ __launch_bounds__(256) void add_kernel(const bf16x8* __restrict__ a, const bf16x8* __restrict__ b,
                                                  bf16x8* __restrict__ out, long n8) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
        float x[8], z[8];
        unpack8(a[i], x);
        unpack8(b[i], z);
#pragma unroll
        for (int k = 0; k < 8; ++k) x[k] += z[k];
        out[i] = pack8(x);
    }
}

// ------------------------------------------------------------------ max pool (fused BN apply + ReLU in front)
// y raw [N][T][H][W][C] -> out [N][To][Ho][Wo][C] (bf16) + argmax tap index (uint8)
template <bool BN>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const bf16x8* __restrict__ y, const float* __restrict__ params,
                                                          bf16x8* __restrict__ out, uint8_t* __restrict__ idx,
                                                          VcPoolGeom g) {
    const int groups = g.C >> 3;
    const long total = (long)g.N * g.To * g.Ho * g.Wo * groups;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int cg = (int)(i % groups);
        long pos = i / groups;
        const int wo = (int)(pos % g.Wo);
        pos /= g.Wo;
        const int ho = (int)(pos % g.Ho);
        pos /= g.Ho;
        const int to = (int)(pos % g.To);
        const int n = (int)(pos / g.To);
        float sc[8], sh[8];
        if (BN) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                sc[k] = params[2 * g.C + cg * 8 + k];
                sh[k] = params[3 * g.C + cg * 8 + k];
            }
        }
        float best[8];
        int bi[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            best[k] = -INFINITY;
            bi[k] = 0;
        }
        for (int a = 0; a < g.kt; ++a) {
            const int t = to * g.st - g.pt + a;
            if ((unsigned)t >= (unsigned)g.T) continue;
            if (g.kh == 3 && g.kw == 3) {
                // the nine loads of a plane go out together (clamped addresses, validity applied at the compare):
                // one load per tap behind its own bounds branch left a thread waiting out 27 memory latencies in a row
                bf16x8 v9[9];
                const int h0 = ho * g.sh - g.ph, w0 = wo * g.sw - g.pw;
                const long plane = ((long)n * g.T + t) * g.H;
#pragma unroll
                for (int j = 0; j < 9; ++j) {
                    const int h = min(max(h0 + j / 3, 0), g.H - 1), w = min(max(w0 + j % 3, 0), g.W - 1);
                    v9[j] = y[((plane + h) * g.W + w) * groups + cg];
                }
#pragma unroll
                for (int j = 0; j < 9; ++j) {
                    const bool ok = (unsigned)(h0 + j / 3) < (unsigned)g.H && (unsigned)(w0 + j % 3) < (unsigned)g.W;
                    float f[8];
                    unpack8(v9[j], f);
                    const int tap = a * 9 + j;
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        float v = f[k];
                        if (BN) {
                            v = v * sc[k] + sh[k];
                            v = v > 0.f ? v : 0.f;
                            v = bf2f(f2bf(v));
                        }
                        if (ok && v > best[k]) {
                            best[k] = v;
                            bi[k] = tap;
                        }
                    }
                }
                continue;
            }
            for (int b = 0; b < g.kh; ++b) {
                const int h = ho * g.sh - g.ph + b;
                if ((unsigned)h >= (unsigned)g.H) continue;
                for (int c = 0; c < g.kw; ++c) {
                    const int w = wo * g.sw - g.pw + c;
                    if ((unsigned)w >= (unsigned)g.W) continue;
                    const long src = ((((long)n * g.T + t) * g.H + h) * g.W + w) * groups + cg;
                    float f[8];
                    unpack8(y[src], f);
                    const int tap = (a * g.kh + b) * g.kw + c;
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        float v = f[k];
                        if (BN) {
                            v = v * sc[k] + sh[k];
                            v = v > 0.f ? v : 0.f;
                            v = bf2f(f2bf(v));  // the value a materialised activation would hold
                        }
                        if (v > best[k]) {
                            best[k] = v;
                            bi[k] = tap;
                        }
                    }
                }
            }
        }
        out[i] = pack8(best);
        uint64_t packed = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) packed |= (uint64_t)(bi[k] & 0xff) << (8 * k);
        *(uint64_t*)(idx + i * 8) = packed;
    }
}

// The reference's pool (kernel 3, stride 2, padding 1 on every axis) with BatchNorm + ReLU in front, tiled.
// The thread-per-output kernel above reads every input element through ~3.4 overlapping windows (measured: 862 MB of
// HBM traffic for a 411 MB tensor) and transforms it as often.  Here a workgroup owns one pooled row (n, ho) and walks
// the T input planes once: the three input rows of a plane are loaded with coalesced 16-byte loads, transformed ONCE
// (scale, shift, ReLU, bf16 rounding) and staged in LDS (double buffered; the next plane's loads are in flight while
// this one is reduced).  Activations are >= 0, so their bf16 bits order like unsigned integers and value and tap go
// into one key, (bits << 16) | (255 - tap): the maximum with the first tap winning a tie is an integer max.  The
// in-plane 9-tap maximum of an odd plane 2c+1 serves window c (as kt = 2) and window c+1 (kt = 0): the tap byte is
// 255 - 9 kt - j, so the second use is the same key plus 18 and each plane is reduced once.
template <int MAXV, int NO>  // staged vectors / outputs per thread: 3 * rowv <= 256 * MAXV, Wo * groups <= 256 * NO
__global__ __launch_bounds__(256) void maxpool_bn_tiled_kernel(const bf16x8* __restrict__ y, const float* __restrict__ params,
                                                               bf16x8* __restrict__ out, uint8_t* __restrict__ idx,
                                                               VcPoolGeom g) {
    extern __shared__ __attribute__((aligned(16))) char pool_sm[];
    const int groups = g.C >> 3;
    const int rowv = g.W * groups;  // 16-byte vectors per input row
    bf16x8* sZ = (bf16x8*)pool_sm;  // [2][3][rowv]
    // consecutive pooled rows (which share an input row) on the same XCD: workgroups are dealt round-robin to the 8 XCDs
    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    if ((nwg & 7) == 0) bid = (bid & 7) * (nwg >> 3) + (bid >> 3);
    const int ho = bid % g.Ho, n = bid / g.Ho;
    const int h0 = 2 * ho - 1;
    const int cg = threadIdx.x % groups;  // 256 % groups == 0 (host check): fixed per thread in every loop below
    float sc[8], sh[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        sc[k] = params[2 * g.C + cg * 8 + k];
        sh[k] = params[3 * g.C + cg * 8 + k];
    }
    // per-thread constants of the plane loop (no divisions or 64-bit multiplies inside it: the loop is VALU-bound)
    // two planes of loads in flight (one left HBM at 2.5 TB/s: too few bytes in flight per CU).  Held as plain dwords: a
    // __bf16 vector that lives across the predicated loads below is "repacked" half by half at every merge point
    // (v_lshrrev + v_perm_b32 per dword, 900 of this kernel's 2700 instructions, and the kernel is VALU-bound)
    u32x4 ldA[MAXV], ldB[MAXV];
    int src_off[MAXV];  // vector offset inside an input plane; -1: nothing to stage
#pragma unroll
    for (int u = 0; u < MAXV; ++u) {
        const int i = threadIdx.x + u * 256;
        const int b = i / rowv, r = i - b * rowv;
        const int h = min(max(h0 + b, 0), g.H - 1);  // rows outside the image are never read back
        src_off[u] = i < 3 * rowv ? h * rowv + r : -1;
    }
    const bf16x8* yplane = y + (long)n * g.T * g.H * rowv;
    const long plane_stride = (long)g.H * rowv;
    auto fetch = [&](u32x4 (&ld)[MAXV]) {
#pragma unroll
        for (int u = 0; u < MAXV; ++u)
            if (src_off[u] >= 0) ld[u] = *(const u32x4*)&yplane[src_off[u]];
        yplane += plane_stride;
    };
    auto stage = [&](int buf, const u32x4 (&ld)[MAXV]) {
        bf16x8* dst = sZ + buf * 3 * rowv + threadIdx.x;
#pragma unroll
        for (int u = 0; u < MAXV; ++u) {
            if (src_off[u] >= 0) {
                float f[8];
                unpack8(__builtin_bit_cast(bf16x8, ld[u]), f);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float v = f[k] * sc[k] + sh[k];
                    f[k] = v > 0.f ? v : 0.f;
                }
                dst[u * 256] = pack8(f);
            }
        }
    };
    int wo_[NO], tap_off[NO][3];  // pooled column; LDS vector offset of the window's three columns in a staged row
    bool okw0[NO], okw2[NO];
#pragma unroll
    for (int o = 0; o < NO; ++o) {
        wo_[o] = (threadIdx.x + o * 256) / groups;
        const int w0 = 2 * wo_[o] - 1;
        okw0[o] = w0 >= 0, okw2[o] = w0 + 2 < g.W;
        tap_off[o][0] = max(w0, 0) * groups + cg;
        tap_off[o][1] = min(w0 + 1, g.W - 1) * groups + cg;
        tap_off[o][2] = min(w0 + 2, g.W - 1) * groups + cg;
    }
    const bool okh[3] = {h0 >= 0, true, h0 + 2 < g.H};
    unsigned cur[NO][8], nxt[NO][8];  // the window being filled; the next one's kt = 0 plane
#pragma unroll
    for (int o = 0; o < NO; ++o)
#pragma unroll
        for (int k = 0; k < 8; ++k) cur[o][k] = nxt[o][k] = 0u;  // below every valid key (tap byte >= 1)
    auto emit = [&](int to) {
        if (to >= g.To) return;
#pragma unroll
        for (int o = 0; o < NO; ++o) {
            const int wo = wo_[o];
            if (wo >= g.Wo) continue;
            float best[8];
            uint64_t packed = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                best[k] = bf2f(__builtin_bit_cast(bf16, (unsigned short)(cur[o][k] >> 16)));
                packed |= (uint64_t)(255u - (cur[o][k] & 0xffu)) << (8 * k);
            }
            const long e = ((((long)n * g.To + to) * g.Ho + ho) * g.Wo + wo) * groups + cg;
            out[e] = pack8(best);
            *(uint64_t*)(idx + e * 8) = packed;
        }
    };
    auto plane = [&](int t, u32x4 (&ld)[MAXV]) {
        const int buf = t & 1;
        stage(buf, ld);
        __syncthreads();
        if (t + 2 < g.T) fetch(ld);
        const bf16x8* zb = sZ + buf * 3 * rowv;
#pragma unroll
        for (int o = 0; o < NO; ++o) {
            if (wo_[o] >= g.Wo) continue;  // threads beyond the row only stage
            // one v_perm_b32 per key: bytes (bits.hi, bits.lo, 0, tap byte) picked from {v, lb}; a column outside the
            // image selects four zero bytes instead (key 0, below every valid key)
            const unsigned selw[3][2] = {{okw0[o] ? 0x05040c00u : 0x0c0c0c0cu, okw0[o] ? 0x07060c00u : 0x0c0c0c0cu},
                                         {0x05040c00u, 0x07060c00u},
                                         {okw2[o] ? 0x05040c00u : 0x0c0c0c0cu, okw2[o] ? 0x07060c00u : 0x0c0c0c0cu}};
            unsigned kp[8];  // in-plane maximum with tap byte 255 - j (kt = 0)
#pragma unroll
            for (int k = 0; k < 8; ++k) kp[k] = 0u;
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                if (!okh[b]) continue;  // uniform
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const u32x4 v = *(const u32x4*)&zb[b * rowv + tap_off[o][c]];
                    const unsigned lb = 255u - (unsigned)(b * 3 + c);
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        kp[2 * m] = max(kp[2 * m], __builtin_amdgcn_perm(v[m], lb, selw[c][0]));
                        kp[2 * m + 1] = max(kp[2 * m + 1], __builtin_amdgcn_perm(v[m], lb, selw[c][1]));
                    }
                }
            }
            if ((t & 1) == 0) {  // plane 2c: kt = 1 of window c
#pragma unroll
                for (int k = 0; k < 8; ++k) cur[o][k] = max(cur[o][k], kp[k] - 9u);
            } else {  // plane 2c+1: kt = 2 of window c; its kt = 0 use (window c+1) follows the emit below
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    cur[o][k] = max(cur[o][k], kp[k] - 18u);
                    nxt[o][k] = kp[k];
                }
            }
        }
        if ((t & 1) == 1) {
            emit(t >> 1);
#pragma unroll
            for (int o = 0; o < NO; ++o)
#pragma unroll
                for (int k = 0; k < 8; ++k) cur[o][k] = nxt[o][k];
        } else if (t == g.T - 1) {
            emit(t >> 1);
        }
    };
    fetch(ldA);
    if (g.T > 1) fetch(ldB);
    for (int t = 0; t < g.T; t += 2) {
        plane(t, ldA);
        if (t + 1 < g.T) plane(t + 1, ldB);
    }
}

// gradient of max pool w.r.t. its input, gather form: every input position looks at the
// windows that contain it and takes the pooled gradient where it was the arg-max.
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const bf16x8* __restrict__ dout, const uint8_t* __restrict__ idx,
                                                          bf16x8* __restrict__ din, VcPoolGeom g) {
    const int groups = g.C >> 3;
    const long total = (long)g.N * g.T * g.H * g.W * groups;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int cg = (int)(i % groups);
        long pos = i / groups;
        const int w = (int)(pos % g.W);
        pos /= g.W;
        const int h = (int)(pos % g.H);
        pos /= g.H;
        const int t = (int)(pos % g.T);
        const int n = (int)(pos / g.T);
        float acc[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = 0.f;
        // windows o with o*s - p <= i <= o*s - p + k - 1
        const int to_lo = max(0, (t + g.pt - g.kt + g.st) / g.st), to_hi = min(g.To - 1, (t + g.pt) / g.st);
        const int ho_lo = max(0, (h + g.ph - g.kh + g.sh) / g.sh), ho_hi = min(g.Ho - 1, (h + g.ph) / g.sh);
        const int wo_lo = max(0, (w + g.pw - g.kw + g.sw) / g.sw), wo_hi = min(g.Wo - 1, (w + g.pw) / g.sw);
        for (int to = to_lo; to <= to_hi; ++to)
            for (int ho = ho_lo; ho <= ho_hi; ++ho)
                for (int wo = wo_lo; wo <= wo_hi; ++wo) {
                    const int tap = ((t + g.pt - to * g.st) * g.kh + (h + g.ph - ho * g.sh)) * g.kw +
                                    (w + g.pw - wo * g.sw);
                    const long o = ((((long)n * g.To + to) * g.Ho + ho) * g.Wo + wo) * groups + cg;
                    const uint64_t packed = *(const uint64_t*)(idx + o * 8);
                    float d[8];
                    unpack8(dout[o], d);
#pragma unroll
                    for (int k = 0; k < 8; ++k)
                        if ((int)((packed >> (8 * k)) & 0xff) == tap) acc[k] += d[k];
                }
        din[i] = pack8(acc);
    }
}


// ------------------------------------------------------------------ BN backward through a max pool (the stem)
// The gradient entering BN+ReLU is the max-pool input gradient.  Materialising it costs a full-size
// write and two full-size reads; instead both BN-backward passes gather it from the pooled gradient
// (1/8 of the size, cache resident): position (t,h,w) takes dout[o] from every window o that contains
// it and whose stored arg-max tap is this position.  MASK-2 semantics (ReLU mask recomputed from y).
// Both kernels walk (n,t,h) rows: the row's coordinates and its window ranges along t and h are
// wave-uniform (scalar arithmetic); only w and the channel group vary per lane.
struct PoolRow {
    int n, t, h, to_lo, to_hi, ho_lo, ho_hi;
};
__device__ __forceinline__ PoolRow pool_row(const VcPoolGeom& g, int row) {
    PoolRow r;
    r.h = row % g.H;
    const int q = row / g.H;
    r.t = q % g.T;
    r.n = q / g.T;
    r.to_lo = max(0, (r.t + g.pt - g.kt + g.st) / g.st), r.to_hi = min(g.To - 1, (r.t + g.pt) / g.st);
    r.ho_lo = max(0, (r.h + g.ph - g.kh + g.sh) / g.sh), r.ho_hi = min(g.Ho - 1, (r.h + g.ph) / g.sh);
    return r;
}
// Gradient of the pool input at (row r, w), 8 channels: at most 2 windows per axis contain a position
// (the host checks kernel <= 2*stride), so the 8 candidates are laid out straight-line: all their loads
// are issued before the first compare instead of one dependent load pair per loop trip.
__device__ __forceinline__ void pool_gather_row8(float (&acc)[8], const bf16x8* __restrict__ dout,
                                                 const uint8_t* __restrict__ idx, const VcPoolGeom& g,
                                                 const PoolRow& r, int w, int cg, int groups) {
    const int wo_lo = max(0, (w + g.pw - g.kw + g.sw) / g.sw), wo_hi = min(g.Wo - 1, (w + g.pw) / g.sw);
    uint64_t pk[8];
    bf16x8 dv[8];
    int tp[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int to = r.to_lo + (j >> 2), ho = r.ho_lo + ((j >> 1) & 1), wo = wo_lo + (j & 1);
        const bool ok = to <= r.to_hi && ho <= r.ho_hi && wo <= wo_hi;
        tp[j] = ok ? ((r.t + g.pt - to * g.st) * g.kh + (r.h + g.ph - ho * g.sh)) * g.kw + (w + g.pw - wo * g.sw) : 256;
        pk[j] = 0;
        dv[j] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
        if (ok) {
            const long o = ((((long)r.n * g.To + to) * g.Ho + ho) * g.Wo + wo) * groups + cg;
            pk[j] = *(const uint64_t*)(idx + o * 8);
            dv[j] = dout[o];
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if ((int)((pk[j] >> (8 * k)) & 0xff) == tp[j]) acc[k] += bf2f(dv[j][k]);
    }
}

// sums of g and g*xhat.  The pool-input gradient is non-zero only at arg-max positions, so the sums are
// taken from the POOLED side (1/8 of the elements): every pooled element adds its gradient at its own
// arg-max input position (found from the stored tap), masked by the recomputed ReLU.
// With `pooled` (the pool's bf16 output z = relu(scale*y + shift) at the arg-max) the value of y there is
// solved from z instead of being gathered with one 2-byte load per element (the gather touches a new
// sector per element: 0.8 GB of HBM reads for the stem, against 0.1 GB for the two pooled tensors).
// z > 0 <=> the ReLU passed; y = (z - shift) / scale carries z's bf16 rounding, |z / gamma| * 2^-9 in
// xhat, so a channel with |beta| > 16 |gamma| (or gamma == 0) keeps the gather.
__global__ __launch_bounds__(256) void bn_bwd_reduce_pooled_kernel(const bf16* __restrict__ y,
                                                                   const bf16x8* __restrict__ dout,
                                                                   const uint8_t* __restrict__ idx,
                                                                   const bf16x8* __restrict__ pooled,
                                                                   const float* __restrict__ params, float* sums,
                                                                   VcPoolGeom g, VcBnBwdFin fin, int fixup) {
    // fixup: the channels solved from z have been summed by chan_reduce_kernel<1, 4>; this launch adds the others
    // (normally none: every thread skips its loop and its atomics) and runs the finalize
    __shared__ float sm[256 * 17];
    __shared__ int lut[256];  // tap -> input row offset from the window origin
    const int C = g.C, groups = C >> 3;
    const int cg = threadIdx.x % groups;  // fixed per lane: the grid stride is a multiple of `groups`
    {
        const int tap = threadIdx.x;
        const int c = tap % g.kw, b = (tap / g.kw) % g.kh, a = tap / (g.kw * g.kh);
        lut[tap] = (a * g.H + b) * g.W + c;
    }
    float a0[8], a1[8], mean[8], invstd[8], scale[8], shift[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        a0[k] = a1[k] = 0.f;
        mean[k] = params[cg * 8 + k], invstd[k] = params[C + cg * 8 + k];
        scale[k] = params[2 * C + cg * 8 + k], shift[k] = params[3 * C + cg * 8 + k];
    }
    // channels whose y is solved from the pooled value
    const bool from_z = pooled != nullptr;
    bool zok[8], all_z = from_z;
    float rscale[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        zok[k] = vc_pooled_z_ok(mean[k], invstd[k], scale[k], shift[k]);
        all_z = all_z && zok[k];
        rscale[k] = zok[k] ? 1.f / scale[k] : 0.f;
    }
    const bool idle = __syncthreads_and(fixup && all_z) != 0;  // (also: lut is written)
    const long total = idle ? 0 : (long)g.N * g.To * g.Ho * g.Wo * groups;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        if (fixup && all_z) continue;
        float d[8];
        unpack8(dout[i], d);
        float yv[8];
        bool on[8];
        if (from_z) {
            float z[8];
            unpack8(pooled[i], z);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                on[k] = z[k] > 0.f;
                yv[k] = (z[k] - shift[k]) * rscale[k];
            }
        }
        if (!all_z) {
            // (the position is only needed for the gather: its 64-bit divisions cost the pure streaming case, which every
            // sane BatchNorm is, 117 us for 0.1 GB)
            long pos = i / groups;
            const int wo = (int)(pos % g.Wo);
            pos /= g.Wo;
            const int ho = (int)(pos % g.Ho);
            pos /= g.Ho;
            const int to = (int)(pos % g.To);
            const int n = (int)(pos / g.To);
            const long origin = (((long)n * g.T + (to * g.st - g.pt)) * g.H + (ho * g.sh - g.ph)) * g.W + (wo * g.sw - g.pw);
            const uint64_t packed = *(const uint64_t*)(idx + i * 8);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (!from_z || !zok[k]) {
                    const int tap = (int)((packed >> (8 * k)) & 0xff);
                    yv[k] = bf2f(y[(origin + lut[tap]) * C + cg * 8 + k]);
                    on[k] = (yv[k] * scale[k] + shift[k]) > 0.f;
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float gk = (on[k] && !(fixup && zok[k])) ? d[k] : 0.f;
            a0[k] += gk;
            a1[k] += gk * (yv[k] - mean[k]) * invstd[k];
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        sm[threadIdx.x * 17 + k] = a0[k];
        sm[threadIdx.x * 17 + 8 + k] = a1[k];
    }
    __syncthreads();
    if (threadIdx.x < groups && !idle) {
        const int rl = 256 / groups;
        for (int k = 0; k < 16; ++k) {
            float s = 0.f;
            for (int rr = 0; rr < rl; ++rr) s += sm[(rr * groups + threadIdx.x) * 17 + k];
            vc_stat_add(sums, blockIdx.x, C, (k < 8 ? 0 : C) + threadIdx.x * 8 + (k & 7), s);
        }
    }
    vc_bn_bwd_fin_last(fin, sums, C);
}

__global__ __launch_bounds__(256) void bn_bwd_apply_pooled_kernel(const bf16x8* __restrict__ y,
                                                                  const bf16x8* __restrict__ dout,
                                                                  const uint8_t* __restrict__ idx,
                                                                  const float* __restrict__ params,
                                                                  const float* __restrict__ coef,
                                                                  bf16x8* __restrict__ dy, VcPoolGeom g) {
    const int C = g.C, groups = C >> 3;
    const int cg = threadIdx.x % groups;
    const int per_row = g.W * groups;
    const int nrows = g.N * g.T * g.H;
    BnBwdCoef kc;
    load_bwd_coef(kc, params, coef, C, cg * 8);
    for (int row = blockIdx.x; row < nrows; row += gridDim.x) {
        const PoolRow r = pool_row(g, row);
        const bf16x8* yrow = y + (long)row * per_row;
        bf16x8* drow = dy + (long)row * per_row;
        for (int i = threadIdx.x; i < per_row; i += 256) {
            const int w = i / groups;
            float f[8], d[8], o[8];
            unpack8(yrow[i], f);
            pool_gather_row8(d, dout, idx, g, r, w, cg, groups);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float gk = (f[k] * kc.scale[k] + kc.shift[k]) > 0.f ? d[k] : 0.f;
                const float xhat = (f[k] - kc.mean[k]) * kc.invstd[k];
                o[k] = kc.scale[k] * (gk - kc.c1[k] - xhat * kc.c2[k]);
            }
            drow[i] = pack8(o);
        }
    }
}

// The same pass for the reference's pool (kernel 3, stride 2, padding 1 on every axis), tiled: a workgroup owns the
// input rows t in {2a+1, 2a+2} x h in {2b+1, 2b+2} of one clip and stages the four pooled rows (to in {a, a+1},
// ho in {b, b+1}: gradient and arg-max bytes) that their windows live in into LDS with coalesced loads.  An odd
// coordinate 2c+1 lies in windows c (as tap 2) and c+1 (tap 0), an even one 2c+2 in window c+1 only (tap 1), so the
// candidate windows of a position are known from the parities: t and h are uniform over the workgroup and the w
// loop runs over even then odd columns, which makes the candidate list wave-uniform -- 3.4 candidates per position
// on average instead of eight predicated ones (the row-at-a-time kernel above spends its time in the 8 x 8 byte
// compares: VALU-bound at 37 % of the HBM rate).  Sums are taken in (to, ho, wo) order like maxpool_bwd_kernel.
__device__ __forceinline__ void pool_match8(float (&acc)[8], uint64_t pk, bf16x8 dv, int tap) {
    const unsigned lo = (unsigned)pk, hi = (unsigned)(pk >> 32);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if ((int)((lo >> (8 * k)) & 0xffu) == tap) acc[k] += bf2f(dv[k]);
        if ((int)((hi >> (8 * k)) & 0xffu) == tap) acc[4 + k] += bf2f(dv[4 + k]);
    }
}
__global__ __launch_bounds__(256) void bn_bwd_apply_pooled_tiled_kernel(const bf16x8* __restrict__ y,
                                                                        const bf16x8* __restrict__ dout,
                                                                        const uint8_t* __restrict__ idx,
                                                                        const float* __restrict__ params,
                                                                        const float* __restrict__ coef,
                                                                        bf16x8* __restrict__ dy, VcPoolGeom g, int nbt,
                                                                        int nbh) {
    extern __shared__ __attribute__((aligned(16))) char pool_sm[];
    const int C = g.C, groups = C >> 3;
    const int prow = g.Wo * groups;  // 16-byte vectors per pooled row
    bf16x8* sD = (bf16x8*)pool_sm;                             // [2][2][prow]
    uint64_t* sI = (uint64_t*)(pool_sm + (size_t)4 * prow * 16);  // [2][2][prow]
    // neighbouring workgroups stage the same pooled rows (each row serves 2 x 2 of them): XCD k takes the k-th contiguous
    // eighth of the workgroups, so that the re-reads are L2 hits (they were 0.3 GB of this launch's HBM reads for 76 MB
    // of pooled tensors; workgroups are dealt round-robin to the 8 XCDs)
    unsigned bid = blockIdx.x;
    if ((gridDim.x & 7u) == 0u) bid = (bid & 7u) * (gridDim.x >> 3) + (bid >> 3);
    const int bh = (int)bid % nbh;
    const int q = (int)bid / nbh;
    const int bt = q % nbt, n = q / nbt;
    const int a = bt - 1, b = bh - 1;
    for (int s = 0; s < 4; ++s) {
        const int to = a + (s >> 1), ho = b + (s & 1);
        if (to < 0 || to >= g.To || ho < 0 || ho >= g.Ho) continue;  // uniform
        const long base = (((long)n * g.To + to) * g.Ho + ho) * prow;
        for (int i = threadIdx.x; i < prow; i += 256) {
            sD[s * prow + i] = dout[base + i];
            sI[s * prow + i] = *(const uint64_t*)(idx + (base + i) * 8);
        }
    }
    const int cg = threadIdx.x % groups;  // 256 is a multiple of groups (host check)
    BnBwdCoef kc;
    load_bwd_coef(kc, params, coef, C, cg * 8);
    __syncthreads();
    const int per_row = g.W * groups;
    const int half = 256 / groups;  // columns of one parity covered per trip
    for (int r = 0; r < 4; ++r) {
        const int t = 2 * a + 1 + (r >> 1), h = 2 * b + 1 + (r & 1);
        if (t < 0 || t >= g.T || h < 0 || h >= g.H) continue;  // uniform
        // candidate (slot, tap) pairs along t and h: odd -> windows {c (tap 2), c+1 (tap 0)}, even -> {c+1 (tap 1)}
        const bool todd = (r >> 1) == 0, hodd = (r & 1) == 0;
        const bool tv[2] = {todd && a >= 0 && a < g.To, a + 1 < g.To}, hv[2] = {hodd && b >= 0 && b < g.Ho, b + 1 < g.Ho};
        const int tt[2] = {2, todd ? 0 : 1}, ht[2] = {2, hodd ? 0 : 1};  // slot i = window c + i
        const long row = ((long)n * g.T + t) * g.H + h;
        const bf16x8* yrow = y + row * per_row;
        bf16x8* drow = dy + row * per_row;
        for (int par = 0; par < 2; ++par) {
            for (int j0 = 0; 2 * j0 + par < g.W; j0 += half) {
                const int jw = j0 + threadIdx.x / groups;
                const int w = 2 * jw + par;
                if (w >= g.W) continue;
                // y is read once and dy written once (0.8 GB): keep them from evicting the pooled rows, which the
                // neighbouring workgroups re-read from L2
                const bf16x8 yv = __builtin_nontemporal_load(&yrow[w * groups + cg]);
                float d[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) d[k] = 0.f;
#pragma unroll
                for (int it = 0; it < 2; ++it)
#pragma unroll
                    for (int ih = 0; ih < 2; ++ih) {
                        if (!tv[it] || !hv[ih]) continue;  // uniform
                        const int slot = (it * 2 + ih) * prow + cg;
                        const int tap = (tt[it] * 3 + ht[ih]) * 3;
                        if (par == 0) {  // w = 2 jw: window jw, tap 1
                            if (jw < g.Wo) pool_match8(d, sI[slot + jw * groups], sD[slot + jw * groups], tap + 1);
                        } else {  // w = 2 jw + 1: windows jw (tap 2) and jw + 1 (tap 0)
                            if (jw < g.Wo) pool_match8(d, sI[slot + jw * groups], sD[slot + jw * groups], tap + 2);
                            if (jw + 1 < g.Wo)
                                pool_match8(d, sI[slot + (jw + 1) * groups], sD[slot + (jw + 1) * groups], tap);
                        }
                    }
                float f[8], o[8];
                unpack8(yv, f);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float gk = (f[k] * kc.scale[k] + kc.shift[k]) > 0.f ? d[k] : 0.f;
                    const float xhat = (f[k] - kc.mean[k]) * kc.invstd[k];
                    o[k] = kc.scale[k] * (gk - kc.c1[k] - xhat * kc.c2[k]);
                }
                __builtin_nontemporal_store(pack8(o), &drow[w * groups + cg]);
            }
        }
    }
}

// The tiled pass with every y load of the workgroup issued up front (NT trips of 256 / groups columns per parity and
// row: 8 NT 16-byte loads per thread in flight before the pooled rows are staged).  The loop form above has one load
// per thread in flight (7 waves per SIMD x 16 B: ~7 MB over the chip, under what 2 us of HBM latency wants at 5 TB/s).
template <int NT>
__global__ __launch_bounds__(256) void bn_bwd_apply_pooled_pf_kernel(const bf16x8* __restrict__ y,
                                                                     const bf16x8* __restrict__ dout,
                                                                     const uint8_t* __restrict__ idx,
                                                                     const float* __restrict__ params,
                                                                     const float* __restrict__ coef,
                                                                     bf16x8* __restrict__ dy, VcPoolGeom g, int nbt, int nbh) {
    extern __shared__ __attribute__((aligned(16))) char pool_sm[];
    const int C = g.C, groups = C >> 3;
    const int prow = g.Wo * groups;
    bf16x8* sD = (bf16x8*)pool_sm;                                // [2][2][prow]
    uint64_t* sI = (uint64_t*)(pool_sm + (size_t)4 * prow * 16);  // [2][2][prow]
    unsigned bid = blockIdx.x;
    if ((gridDim.x & 7u) == 0u) bid = (bid & 7u) * (gridDim.x >> 3) + (bid >> 3);
    const int bh = (int)bid % nbh;
    const int q = (int)bid / nbh;
    const int bt = q % nbt, n = q / nbt;
    const int a = bt - 1, b = bh - 1;
    const int cg = threadIdx.x % groups;
    const int per_row = g.W * groups;
    const int half = 256 / groups;
    const int jw0 = threadIdx.x / groups;
    bf16x8 yv[4][2][NT];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int t = 2 * a + 1 + (r >> 1), h = 2 * b + 1 + (r & 1);
        if (t < 0 || t >= g.T || h < 0 || h >= g.H) continue;  // uniform
        const bf16x8* yrow = y + (((long)n * g.T + t) * g.H + h) * per_row;
#pragma unroll
        for (int par = 0; par < 2; ++par)
#pragma unroll
            for (int u = 0; u < NT; ++u) {
                const int w = 2 * (u * half + jw0) + par;
                if (w < g.W) yv[r][par][u] = __builtin_nontemporal_load(&yrow[w * groups + cg]);
            }
    }
    for (int s = 0; s < 4; ++s) {
        const int to = a + (s >> 1), ho = b + (s & 1);
        if (to < 0 || to >= g.To || ho < 0 || ho >= g.Ho) continue;  // uniform
        const long base = (((long)n * g.To + to) * g.Ho + ho) * prow;
        for (int i = threadIdx.x; i < prow; i += 256) {
            sD[s * prow + i] = dout[base + i];
            sI[s * prow + i] = *(const uint64_t*)(idx + (base + i) * 8);
        }
    }
    BnBwdCoef kc;
    load_bwd_coef(kc, params, coef, C, cg * 8);
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int t = 2 * a + 1 + (r >> 1), h = 2 * b + 1 + (r & 1);
        if (t < 0 || t >= g.T || h < 0 || h >= g.H) continue;  // uniform
        const bool todd = (r >> 1) == 0, hodd = (r & 1) == 0;
        const bool tv[2] = {todd && a >= 0 && a < g.To, a + 1 < g.To}, hv[2] = {hodd && b >= 0 && b < g.Ho, b + 1 < g.Ho};
        const int tt[2] = {2, todd ? 0 : 1}, ht[2] = {2, hodd ? 0 : 1};
        bf16x8* drow = dy + (((long)n * g.T + t) * g.H + h) * per_row;
#pragma unroll
        for (int par = 0; par < 2; ++par)
#pragma unroll
            for (int u = 0; u < NT; ++u) {
                const int jw = u * half + jw0;
                const int w = 2 * jw + par;
                if (w >= g.W) continue;
                float d[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) d[k] = 0.f;
#pragma unroll
                for (int it = 0; it < 2; ++it)
#pragma unroll
                    for (int ih = 0; ih < 2; ++ih) {
                        if (!tv[it] || !hv[ih]) continue;  // uniform
                        const int slot = (it * 2 + ih) * prow + cg;
                        const int tap = (tt[it] * 3 + ht[ih]) * 3;
                        if (par == 0) {
                            if (jw < g.Wo) pool_match8(d, sI[slot + jw * groups], sD[slot + jw * groups], tap + 1);
                        } else {
                            if (jw < g.Wo) pool_match8(d, sI[slot + jw * groups], sD[slot + jw * groups], tap + 2);
                            if (jw + 1 < g.Wo)
                                pool_match8(d, sI[slot + (jw + 1) * groups], sD[slot + (jw + 1) * groups], tap);
                        }
                    }
                float f[8], o[8];
                unpack8(yv[r][par][u], f);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float gk = (f[k] * kc.scale[k] + kc.shift[k]) > 0.f ? d[k] : 0.f;
                    const float xhat = (f[k] - kc.mean[k]) * kc.invstd[k];
                    o[k] = kc.scale[k] * (gk - kc.c1[k] - xhat * kc.c2[k]);
                }
                __builtin_nontemporal_store(pack8(o), &drow[w * groups + cg]);
            }
    }
}

// ------------------------------------------------------------------ global average pool
// in [N][S][C] bf16 -> out [N][C] fp32
__global__ void avgpool_fwd_kernel(const bf16* __restrict__ in, float* __restrict__ out, int N, int S, int C) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * C) return;
    const int n = i / C, c = i - n * C;
    float s = 0.f;
    for (int k = 0; k < S; ++k) s += bf2f(in[((size_t)n * S + k) * C + c]);
    out[i] = s / (float)S;
}
// dfeat [N][C] fp32 -> din [N][S][C] bf16
__global__ void avgpool_bwd_kernel(const float* __restrict__ dfeat, bf16* __restrict__ din, int N, int S, int C) {
    const long total = (long)N * S * C;
    const long stride = (long)gridDim.x * blockDim.x;
    const float inv = 1.f / (float)S;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int c = (int)(i % C);
        const int n = (int)(i / ((long)S * C));
        din[i] = f2bf(dfeat[(size_t)n * C + c] * inv);
    }
}

// ------------------------------------------------------------------ classifier head (fp32)
// one wave per output: logits[n][k] = sum_f feat[n][f] W[k][f] + b[k]
__global__ __launch_bounds__(256) void linear_fwd_kernel(const float* __restrict__ feat, const float* __restrict__ W,
                                                         const float* __restrict__ bias, float* __restrict__ out, int N,
                                                         int F, int K) {
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (o >= N * K) return;
    const int n = o / K, k = o - n * K;
    float s = 0.f;
    for (int f = lane; f < F; f += 64) s += feat[(size_t)n * F + f] * W[(size_t)k * F + f];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) out[o] = s + (bias ? bias[k] : 0.f);
}
// The same sums (same per-lane order), a wave on 4 samples x 4 classes: a loaded value feeds four outputs, which
// quarters the 16 KB of L2 reads per output of the kernel above; with four trips' loads in flight the R3D-50 head
// (64 x 2048 -> 400) went from 43 to 15 us.  The same tiling of the two backward kernels halved their thread count and
// measured slower (data gradient 20 -> 32 us): they stay one thread per output.
__global__ __launch_bounds__(256) void linear_fwd_tiled_kernel(const float* __restrict__ feat, const float* __restrict__ W,
                                                               const float* __restrict__ bias, float* __restrict__ out,
                                                               int N, int F, int K) {
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int kt = (K + 3) >> 2;
    const int tn = wave / kt, tk = wave - tn * kt;
    const int n0 = tn * 4, k0 = tk * 4;
    if (n0 >= N) return;  // wave-uniform
    float s[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) s[i][j] = 0.f;
#pragma unroll 4
    for (int f = lane; f < F; f += 64) {  // (four trips' loads in flight: the trip is latency-bound otherwise)
        float a[4], w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = n0 + i < N ? feat[(size_t)(n0 + i) * F + f] : 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = k0 + j < K ? W[(size_t)(k0 + j) * F + f] : 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) s[i][j] += a[i] * w[j];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v = s[i][j];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
            if (lane == 0 && n0 + i < N && k0 + j < K) out[(size_t)(n0 + i) * K + k0 + j] = v + (bias ? bias[k0 + j] : 0.f);
        }
}
// dfeat[n][f] = sum_k dlog[n][k] W[k][f]
__global__ void linear_bwd_data_kernel(const float* __restrict__ dlog, const float* __restrict__ W,
                                       float* __restrict__ dfeat, int N, int F, int K) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * F) return;
    const int n = i / F, f = i - n * F;
    // eight loads in flight, summed in k order (the plain loop waited out one memory latency per k: 95 us for K = 400)
    float s = 0.f;
    int k = 0;
    for (; k + 8 <= K; k += 8) {
        float w[8], d[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) w[u] = W[(size_t)(k + u) * F + f], d[u] = dlog[(size_t)n * K + k + u];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += d[u] * w[u];
    }
    for (; k < K; ++k) s += dlog[(size_t)n * K + k] * W[(size_t)k * F + f];
    dfeat[i] = s;
}
// dW[k][f] += sum_n dlog[n][k] feat[n][f] ; db[k] += sum_n dlog[n][k]
__global__ void linear_bwd_weight_kernel(const float* __restrict__ dlog, const float* __restrict__ feat,
                                         float* __restrict__ dW, float* __restrict__ db, int N, int F, int K) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= K * F) return;
    const int k = i / F, f = i - k * F;
    float s = 0.f, sb = 0.f;
    int n = 0;
    for (; n + 8 <= N; n += 8) {
        float x[8], d[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = feat[(size_t)(n + u) * F + f], d[u] = dlog[(size_t)(n + u) * K + k];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += d[u] * x[u], sb += d[u];
    }
    for (; n < N; ++n) {
        const float d = dlog[(size_t)n * K + k];
        s += d * feat[(size_t)n * F + f];
        sb += d;
    }
    dW[i] += s;
    if (f == 0 && db != nullptr) db[k] += sb;
}

// ------------------------------------------------------------------ layout
// clip batch x [N][3][T][H][W] (fp32 or bf16) -> zero-padded space-to-depth stem source
// xs [N][T+6][H/2+3][Wp][12], Wp = W/2+3 rounded up to even (rows are then whole 16-byte units: LDS-DMA),
// channel = (bh*2 + bw)*3 + c, pads: T 3/3, H' 2/1, W' 2/1(+1)
template <typename TIN>
__global__ __launch_bounds__(256) void stem_pack_input_kernel(const TIN* __restrict__ x, bf16* __restrict__ xs, int N, int T,
                                                              int H, int W) {
    const int Tp = T + 6, Hp = H / 2 + 3, Wp = (W / 2 + 4) & ~1;
    const long total = (long)N * Tp * Hp * Wp;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        long pos = i;
        const int wp = (int)(pos % Wp);
        pos /= Wp;
        const int hp = (int)(pos % Hp);
        pos /= Hp;
        const int tp = (int)(pos % Tp);
        const int n = (int)(pos / Tp);
        const int t = tp - 3, h2 = hp - 2, w2 = wp - 2;
        bf16 v[12];
        const bool in = (unsigned)t < (unsigned)T && (unsigned)h2 < (unsigned)(H / 2) && (unsigned)w2 < (unsigned)(W / 2);
        // the two pixels (bw = 0, 1) of a source row are adjacent and pair-aligned (W is even): one load for both
        struct alignas(2 * sizeof(TIN)) Pair {
            TIN a, b;
        };
#pragma unroll
        for (int bh = 0; bh < 2; ++bh)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float f0 = 0.f, f1 = 0.f;
                if (in) {
                    const Pair pr = *(const Pair*)(x + ((((long)n * 3 + c) * T + t) * H + (2 * h2 + bh)) * W + 2 * w2);
                    f0 = (float)pr.a, f1 = (float)pr.b;
                }
                v[(bh * 2 + 0) * 3 + c] = f2bf(f0);
                v[(bh * 2 + 1) * 3 + c] = f2bf(f1);
            }
        bf16x4* dst = (bf16x4*)(xs + i * 12);
        dst[0] = (bf16x4){v[0], v[1], v[2], v[3]};
        dst[1] = (bf16x4){v[4], v[5], v[6], v[7]};
        dst[2] = (bf16x4){v[8], v[9], v[10], v[11]};
    }
}

// Clip input pipeline: decoded uint8 frames [N][Ts][Hs][Ws][3] (RGB interleaved, what a video decoder hands over)
// -> temporal sampling + crop + horizontal flip + per-channel normalisation -> the SAME packed stem source as
// above, in one pass: v = u8 * a[c] + b[c] (one fp32 fma; a = 1 / (255 std), b = -mean / std), rounded to bf16.
// crops[n] = {t0, y0, x0, flip}: output (t, h, w) reads frame t0 + t*tstep, row y0 + h, column x0 + (flip ? W-1-w : w).
// A thread owns one packed position = 2 x 2 pixels: two runs of 6 contiguous source bytes; neighbouring threads
// read neighbouring runs.  HBM-bound: 3 B read + 6 B written per pixel (+ the zero padding of the packed form).
__global__ __launch_bounds__(256) void clip_pack_u8_kernel(const uint8_t* __restrict__ frames,
                                                           const int* __restrict__ crops, bf16* __restrict__ xs, int N,
                                                           int T, int H, int W, int Ts, int Hs, int Ws, int tstep,
                                                           float a0, float a1, float a2, float b0, float b1, float b2) {
    const int Tp = T + 6, Hp = H / 2 + 3, Wp = (W / 2 + 4) & ~1;
    const long total = (long)N * Tp * Hp * Wp;
    const long stride = (long)gridDim.x * blockDim.x;
    const float a[3] = {a0, a1, a2}, b[3] = {b0, b1, b2};
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        long pos = i;
        const int wp = (int)(pos % Wp);
        pos /= Wp;
        const int hp = (int)(pos % Hp);
        pos /= Hp;
        const int tp = (int)(pos % Tp);
        const int n = (int)(pos / Tp);
        const int t = tp - 3, h2 = hp - 2, w2 = wp - 2;
        bf16 v[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) v[k] = f2bf(0.f);
        if ((unsigned)t < (unsigned)T && (unsigned)h2 < (unsigned)(H / 2) && (unsigned)w2 < (unsigned)(W / 2)) {
            const int t0 = crops[n * 4], y0 = crops[n * 4 + 1], x0 = crops[n * 4 + 2], flip = crops[n * 4 + 3];
            const uint8_t* fr = frames + ((long)n * Ts + (t0 + t * tstep)) * Hs * Ws * 3;
            // the pixel pair (2*w2, 2*w2+1) of the crop is source columns xl, xl+1 (mirrored pair when flipped)
            const int xl = flip ? x0 + (W - 2 - 2 * w2) : x0 + 2 * w2;
#pragma unroll
            for (int bh = 0; bh < 2; ++bh) {
                const uint8_t* row = fr + ((long)(y0 + 2 * h2 + bh) * Ws + xl) * 3;
                uint8_t px[6];
#pragma unroll
                for (int k = 0; k < 6; ++k) px[k] = row[k];
#pragma unroll
                for (int bw = 0; bw < 2; ++bw)
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const int sb = flip ? 1 - bw : bw;
                        v[(bh * 2 + bw) * 3 + c] = f2bf(fmaf((float)px[sb * 3 + c], a[c], b[c]));
                    }
            }
        }
        bf16x4* dst = (bf16x4*)(xs + i * 12);
        dst[0] = (bf16x4){v[0], v[1], v[2], v[3]};
        dst[1] = (bf16x4){v[4], v[5], v[6], v[7]};
        dst[2] = (bf16x4){v[8], v[9], v[10], v[11]};
    }
}

// generic NCTHW (fp32|bf16) <-> NTHWC bf16, S = T*H*W
template <typename TIN>
__global__ void ncs_to_nsc_kernel(const TIN* __restrict__ in, bf16* __restrict__ out, int N, int C, long S) {
    const long total = (long)N * C * S;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int c = (int)(i % C);
        const long s = (i / C) % S;
        const long n = i / ((long)C * S);
        out[i] = f2bf((float)in[(n * C + c) * S + s]);
    }
}
template <typename TOUT>
__global__ void nsc_to_ncs_kernel(const bf16* __restrict__ in, TOUT* __restrict__ out, int N, int C, long S) {
    const long total = (long)N * C * S;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const long s = i % S;
        const int c = (int)((i / S) % C);
        const long n = i / ((long)C * S);
        out[i] = (TOUT)bf2f(in[(n * S + s) * C + c]);
    }
}

// ------------------------------------------------------------------ weight preparation
// master fp32 [Co][taps][Ci] -> bf16 same layout (forward / wgrad) and bf16 [Ci][taps][Co] (dgrad)
__global__ void pack_weight_kernel(const float* __restrict__ w, bf16* __restrict__ wf, bf16* __restrict__ wd, int Co,
                                   int taps, int Ci) {
    const long total = (long)Co * taps * Ci;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int ci = (int)(i % Ci);
        const int tap = (int)((i / Ci) % taps);
        const int co = (int)(i / ((long)Ci * taps));
        const bf16 v = f2bf(w[i]);
        wf[i] = v;
        if (wd != nullptr) wd[((long)ci * taps + tap) * Co + co] = v;
    }
}

// stem: master fp32 [64][7][7][7][3] -> bf16 [64][1344], K = ((kt*4 + ah)*4 + aw)*12 + (bh*2+bw)*3 + c
// with kh = 2*ah + bh - 1, kw = 2*aw + bw - 1 (index -1 = zero column)
__device__ __forceinline__ int stem_k_to_src(int k, int& valid) {
    const int c = k % 3;
    const int bw = (k / 3) & 1;
    const int bh = (k / 6) & 1;
    const int aw = (k / 12) & 3;
    const int ah = (k / 48) & 3;
    const int kt = k / 192;
    const int kh = 2 * ah + bh - 1, kw = 2 * aw + bw - 1;
    valid = (kh >= 0 && kw >= 0);
    return ((kt * 7 + kh) * 7 + kw) * 3 + c;
}
__global__ void stem_pack_weight_kernel(const float* __restrict__ w, bf16* __restrict__ ws, int Co) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Co * 1344) return;
    const int co = i / 1344, k = i - co * 1344;
    int valid;
    const int src = stem_k_to_src(k, valid);
    ws[i] = f2bf(valid ? w[co * 1029 + src] : 0.f);
}
__global__ void stem_unpack_wgrad_kernel(const float* __restrict__ dws, float* __restrict__ dw, int Co) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Co * 1344) return;
    const int co = i / 1344, k = i - co * 1344;
    int valid;
    const int src = stem_k_to_src(k, valid);
    if (valid) dw[co * 1029 + src] += dws[i];
}


// ------------------------------------------------------------------ optimizer over the flat buffers
// torch.optim.SGD arithmetic (dampening 0): g = grad + wd*p; buf = momentum*buf + g;
// p -= lr * (nesterov ? g + momentum*buf : buf).  Also writes the bf16 shadow of the new parameters:
// a conv weight's slice of it IS its forward / wgrad operand ([Co][taps][Ci], K contiguous).
__global__ __launch_bounds__(256) void sgd_step_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                       float* __restrict__ mom, bf16* __restrict__ shadow, long n,
                                                       float lr, float momentum, float wd, int nesterov) {
    const long n4 = n >> 2;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        f32x4 pv = ((const f32x4*)p)[i];
        f32x4 gv = ((const f32x4*)g)[i];
        gv += wd * pv;
        if (mom != nullptr) {
            f32x4 mv = ((const f32x4*)mom)[i];
            mv = momentum * mv + gv;
            ((f32x4*)mom)[i] = mv;
            gv = nesterov ? gv + momentum * mv : mv;
        }
        pv -= lr * gv;
        ((f32x4*)p)[i] = pv;
        if (shadow != nullptr) ((bf16x4*)shadow)[i] = (bf16x4){f2bf(pv[0]), f2bf(pv[1]), f2bf(pv[2]), f2bf(pv[3])};
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {  // tail
        const long i = (n4 << 2) + threadIdx.x;
        float pv = p[i], gv = g[i] + wd * pv;
        if (mom != nullptr) {
            const float mv = momentum * mom[i] + gv;
            mom[i] = mv;
            gv = nesterov ? gv + momentum * mv : mv;
        }
        pv -= lr * gv;
        p[i] = pv;
        if (shadow != nullptr) shadow[i] = f2bf(pv);
    }
}

// torch.optim.Adam's step (amsgrad off, maximize off) over the flat buffers, one launch for the whole model, in the
// order torch applies it: g += wd p; m += (1 - b1)(g - m)  [lerp]; v = b2 v + (1 - b2) g g;
// p -= (lr / bias1) m / (sqrt(v) / sqrt(bias2) + eps); also writes the bf16 shadow.  bias1 = 1 - b1^t and
// sqrt(bias2) = sqrt(1 - b2^t) come from the host (step count t), as torch computes them.
__global__ __launch_bounds__(256) void adam_step_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                        float* __restrict__ m, float* __restrict__ v,
                                                        bf16* __restrict__ shadow, long n, float step_size, float b1,
                                                        float b2, float eps, float wd, float bias2_sqrt) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride * 4) {
        if (i + 4 <= n) {
            f32x4 pv = *(const f32x4*)(p + i), gv = *(const f32x4*)(g + i), mv = *(const f32x4*)(m + i),
                  vv = *(const f32x4*)(v + i);
            gv += wd * pv;
            mv += (1.f - b1) * (gv - mv);
            vv = b2 * vv + (1.f - b2) * gv * gv;
#pragma unroll
            for (int k = 0; k < 4; ++k) pv[k] -= step_size * (mv[k] / (sqrtf(vv[k]) / bias2_sqrt + eps));
            *(f32x4*)(p + i) = pv, *(f32x4*)(m + i) = mv, *(f32x4*)(v + i) = vv;
            if (shadow != nullptr) *(bf16x4*)(shadow + i) = (bf16x4){f2bf(pv[0]), f2bf(pv[1]), f2bf(pv[2]), f2bf(pv[3])};
        } else {
            for (long j = i; j < n; ++j) {
                float pv = p[j], gv = g[j] + wd * p[j], mv = m[j], vv = v[j];
                mv += (1.f - b1) * (gv - mv);
                vv = b2 * vv + (1.f - b2) * gv * gv;
                pv -= step_size * (mv / (sqrtf(vv) / bias2_sqrt + eps));
                p[j] = pv, m[j] = mv, v[j] = vv;
                if (shadow != nullptr) shadow[j] = f2bf(pv);
            }
        }
    }
}

// fp32 -> bf16 copy of the flat parameters (after a checkpoint load, or with a non-fused optimizer)
__global__ __launch_bounds__(256) void shadow_kernel(const float* __restrict__ p, bf16* __restrict__ shadow, long n) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) shadow[i] = f2bf(p[i]);
}

// All convs in one launch: bf16 [Co][taps][Ci] (slice of the shadow) -> bf16 [Ci][taps][Co] (the dgrad
// operand).  One 64(co) x 64(ci) tile of one tap per block, transposed through LDS so that both the
// reads and the writes are 128-byte rows.  table: int32 [nconv][6] = src_off, dst_off, Co, taps, Ci,
// first tile.
__global__ __launch_bounds__(256) void pack_dgrad_batched_kernel(const bf16* __restrict__ shadow, bf16* __restrict__ wd,
                                                                 const int* __restrict__ table, int nconv) {
    __shared__ bf16 tile[64][66];
    int e = 0;
    for (int i = 1; i < nconv; ++i)
        if (table[i * 6 + 5] <= (int)blockIdx.x) e = i;
    const int* tb = table + e * 6;
    const int Co = tb[2], taps = tb[3], Ci = tb[4];
    int t = (int)blockIdx.x - tb[5];
    const int nci = Ci >> 6, nco = Co >> 6;
    const int cib = t % nci;
    t /= nci;
    const int cob = t % nco;
    const int tap = t / nco;
    const bf16* src = shadow + tb[0];
    bf16* dst = wd + tb[1];
    const int r = threadIdx.x >> 2, sg = threadIdx.x & 3;
    {
        const bf16x8* q = (const bf16x8*)(src + ((size_t)(cob * 64 + r) * taps + tap) * Ci + cib * 64 + sg * 16);
        const bf16x8 a = q[0], b = q[1];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            tile[r][sg * 16 + k] = a[k];
            tile[r][sg * 16 + 8 + k] = b[k];
        }
    }
    __syncthreads();
    bf16x8 a, b;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        a[k] = tile[sg * 16 + k][r];
        b[k] = tile[sg * 16 + 8 + k][r];
    }
    bf16x8* o = (bf16x8*)(dst + ((size_t)(cib * 64 + r) * taps + tap) * Co + cob * 64 + sg * 16);
    o[0] = a;
    o[1] = b;
}

// ------------------------------------------------------------------ cross-entropy of the class scores
// loss = mean_n (logsumexp(logits[n]) - logits[n][y[n]]) and its gradient d loss / d logits = (softmax - onehot) / N in ONE
// launch of one workgroup (N x K is 64 x 101 .. 400: nothing to parallelise over, a chain of four to seven dependent tiny
// launches to save).  16 waves; a wave takes R rows at a time with all their loads issued before the first use (a row
// at a time is a chain of ~2 us round trips), a row lives in registers (K <= 64 KPL); fixed summation order; labels
// outside [0, K) leave NaN in the loss (torch raises for them, which needs the host to see the labels).
template <int KPL, int R>  // K <= 64 KPL; R rows per wave in flight
__global__ __launch_bounds__(1024) void softmax_ce_kernel(const float* __restrict__ logits, const long long* __restrict__ y,
                                                          float* __restrict__ loss, float* __restrict__ dlogits, int N, int K) {
    __shared__ float sm[16];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float part = 0.f;
    for (int n0 = wave * R; n0 < N; n0 += 16 * R) {
        float v[R][KPL];
        long long t[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int n = n0 + r;
            t[r] = n < N ? y[n] : 0;
#pragma unroll
            for (int i = 0; i < KPL; ++i) {
                const int k = lane + i * 64;
                v[r][i] = (n < N && k < K) ? logits[(size_t)n * K + k] : -INFINITY;
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int n = n0 + r;
            if (n >= N) break;  // (wave-uniform)
            float m = -INFINITY;
#pragma unroll
            for (int i = 0; i < KPL; ++i) m = fmaxf(m, v[r][i]);
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
            float e = 0.f, zt = 0.f;
            const bool ok = t[r] >= 0 && t[r] < K;
#pragma unroll
            for (int i = 0; i < KPL; ++i) {
                if (lane + i * 64 == t[r]) zt = v[r][i];
                v[r][i] = expf(v[r][i] - m);  // (exp(-inf) = 0 for the padding)
                e += v[r][i];
            }
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) {
                e += __shfl_xor(e, o, 64);
                zt += __shfl_xor(zt, o, 64);
            }
            part += ok ? (m + logf(e)) - zt : NAN;
            if (dlogits != nullptr) {
                const float inv = 1.f / e, invn = 1.f / (float)N;
#pragma unroll
                for (int i = 0; i < KPL; ++i) {
                    const int k = lane + i * 64;
                    if (k < K) dlogits[(size_t)n * K + k] = (v[r][i] * inv - (k == t[r] ? 1.f : 0.f)) * invn;
                }
            }
        }
    }
    if (lane == 0) sm[wave] = part;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) s += sm[w];
        loss[0] = s / (float)N;
    }
}

inline int grid_for(long n, int block = 256, int cap = 256 * 16) {
    long g = (n + block - 1) / block;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace

// drop any stale error another user of the runtime (torch) left behind, then launch
#define VC_LAUNCH(...)                  \
    do {                                \
        (void)hipGetLastError();        \
        hipLaunchKernelGGL(__VA_ARGS__); \
    } while (0)
static inline int vc_check_launch() {
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess) return 0;
    vc_set_error("kernel launch failed: %s", hipGetErrorString(e));
    return -1;
}
#define VC_CHECK_LAUNCH() vc_check_launch()

int vc_ew_bn_finalize(const float* stats, int C, float count, const float* gamma, const float* beta, float* rmean,
                      float* rvar, float momentum, float eps, float* params, const float* shift, float* shift_out,
                      hipStream_t s) {
    stats = vc_det_tag(stats);
    VC_LAUNCH(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, s, stats, C, count, gamma, beta, rmean,
                       rvar, momentum, eps, params, shift, shift_out);
    return VC_CHECK_LAUNCH();
}
int vc_ew_bn_eval_params(int C, const float* gamma, const float* beta, const float* rmean, const float* rvar, float eps,
                         float* params, hipStream_t s) {
    VC_LAUNCH(bn_eval_params_kernel, dim3((C + 255) / 256), dim3(256), 0, s, C, gamma, beta, rmean, rvar, eps,
                       params);
    return VC_CHECK_LAUNCH();
}
int vc_ew_bn_apply(const void* y, const float* params, const void* res, void* out, long n, int C, int relu,
                   uint8_t* bits, hipStream_t s) {
    const long n8 = n / 8;
    const int g = grid_for(n8);
    const bf16x8* yy = (const bf16x8*)y;
    const bf16x8* rr = (const bf16x8*)res;
    bf16x8* oo = (bf16x8*)out;
    if (bits && res)
        VC_LAUNCH((bn_apply_kernel<true, true, true>), dim3(g), dim3(256), 0, s, yy, params, rr, oo, n8, C, bits);
    else if (bits)
        VC_LAUNCH((bn_apply_kernel<false, true, true>), dim3(g), dim3(256), 0, s, yy, params, rr, oo, n8, C, bits);
    else if (res && relu)
        VC_LAUNCH((bn_apply_kernel<true, true>), dim3(g), dim3(256), 0, s, yy, params, rr, oo, n8, C, (uint8_t*)nullptr);
    else if (res)
        VC_LAUNCH((bn_apply_kernel<true, false>), dim3(g), dim3(256), 0, s, yy, params, rr, oo, n8, C, (uint8_t*)nullptr);
    else if (relu)
        VC_LAUNCH((bn_apply_kernel<false, true>), dim3(g), dim3(256), 0, s, yy, params, rr, oo, n8, C, (uint8_t*)nullptr);
    else
        VC_LAUNCH((bn_apply_kernel<false, false>), dim3(g), dim3(256), 0, s, yy, params, rr, oo, n8, C, (uint8_t*)nullptr);
    return VC_CHECK_LAUNCH();
}

// chan_reduce_kernel: CS = C / 64 channel slices x RS row phases; RS keeps the launch near 2048 blocks, its float
// atomics (RS x 2C) at 256k or fewer, and at least two 32-row groups per block
static void chan_reduce_geometry(long rows, int C, int& grid, int& CS) {
    CS = C / 64;
    long RS = 2048 / CS;
    if (RS > 131072 / C) RS = 131072 / C;
    const long by_rows = (rows + 63) / 64;
    if (RS > by_rows) RS = by_rows;
    if (RS < 1) RS = 1;
    grid = (int)(RS * CS);
}
int vc_ew_bn_stats(const void* y, float* sums, long rows, int C, float* shift_out, hipStream_t s) {
    int grid, slices;
    chan_reduce_geometry(rows, C, grid, slices);
    sums = vc_det_tag(sums);
    VC_LAUNCH((chan_reduce_kernel<0, 0>), dim3(grid), dim3(256), 0, s, (const bf16x8*)y, (const bf16x8*)nullptr,
                       (const bf16x8*)nullptr, (const float*)nullptr, sums, rows, C, slices, VcBnFin{}, VcBnBwdFin{},
                       shift_out);
    return VC_CHECK_LAUNCH();
}
int vc_ew_bn_bwd_reduce(const void* y, const void* dz, const void* aux, const float* params, float* sums, long rows,
                        int C, int mask, const VcBnBwdFin* fin, hipStream_t s) {
    int grid, slices;
    chan_reduce_geometry(rows, C, grid, slices);
    const bf16x8 *yy = (const bf16x8*)y, *dd = (const bf16x8*)dz, *aa = (const bf16x8*)aux;
    const VcBnBwdFin bf = fin ? *fin : VcBnBwdFin{};
    sums = vc_det_tag(sums);
    if (mask == 0)
        VC_LAUNCH((chan_reduce_kernel<1, 0>), dim3(grid), dim3(256), 0, s, yy, dd, aa, params, sums, rows, C, slices,
                  VcBnFin{}, bf, (float*)nullptr);
    else if (mask == 1)
        VC_LAUNCH((chan_reduce_kernel<1, 1>), dim3(grid), dim3(256), 0, s, yy, dd, aa, params, sums, rows, C, slices,
                  VcBnFin{}, bf, (float*)nullptr);
    else if (mask == 3)
        VC_LAUNCH((chan_reduce_kernel<1, 3>), dim3(grid), dim3(256), 0, s, yy, dd, aa, params, sums, rows, C, slices,
                  VcBnFin{}, bf, (float*)nullptr);
    else if (mask == 4)  // internal: vc_ew_bn_bwd_reduce_pooled
        VC_LAUNCH((chan_reduce_kernel<1, 4>), dim3(grid), dim3(256), 0, s, yy, dd, aa, params, sums, rows, C, slices,
                  VcBnFin{}, bf, (float*)nullptr);
    else
        VC_LAUNCH((chan_reduce_kernel<1, 2>), dim3(grid), dim3(256), 0, s, yy, dd, aa, params, sums, rows, C, slices,
                  VcBnFin{}, bf, (float*)nullptr);
    return VC_CHECK_LAUNCH();
}
int vc_ew_splitk_finalize(const float* slabs, int splits, long long stride, void* out, int accumulate, float* stats,
                          long rows, int C, const VcBnFin* fin, hipStream_t s) {
    int grid, slices;
    chan_reduce_geometry(rows, C, grid, slices);
    stats = vc_det_tag(stats);
    if (stats)
        VC_LAUNCH((splitk_finalize_kernel<1>), dim3(grid), dim3(256), 0, s, slabs, splits, stride, (bf16x8*)out,
                  accumulate, stats, rows, C, slices, fin ? *fin : VcBnFin{}, fin ? fin->shift : (const float*)nullptr,
                  VcFinRed{});
    else
        VC_LAUNCH((splitk_finalize_kernel<0>), dim3(grid), dim3(256), 0, s, slabs, splits, stride, (bf16x8*)out,
                  accumulate, stats, rows, C, slices, VcBnFin{}, (const float*)nullptr, VcFinRed{});
    return VC_CHECK_LAUNCH();
}
// the finalize of a split-K DATA gradient with the BatchNorm backward reduction over the finished gradient folded in
// (= vc_ew_splitk_finalize without stats, then vc_ew_bn_bwd_reduce(y, out, bits, params, sums, rows, C, bits ? 3 : 2, fin))
int vc_ew_splitk_finalize_red(const float* slabs, int splits, long long stride, void* out, int accumulate, long rows, int C,
                              const void* y, const float* params, const uint8_t* bits, float* sums,
                              const VcBnBwdFin& fin, hipStream_t s) {
    int grid, slices;
    chan_reduce_geometry(rows, C, grid, slices);
    sums = vc_det_tag(sums);
    const VcFinRed red{(const bf16x8*)y, params, bits, fin};
    if (bits != nullptr)
        VC_LAUNCH((splitk_finalize_kernel<3>), dim3(grid), dim3(256), 0, s, slabs, splits, stride, (bf16x8*)out,
                  accumulate, sums, rows, C, slices, VcBnFin{}, (const float*)nullptr, red);
    else
        VC_LAUNCH((splitk_finalize_kernel<2>), dim3(grid), dim3(256), 0, s, slabs, splits, stride, (bf16x8*)out,
                  accumulate, sums, rows, C, slices, VcBnFin{}, (const float*)nullptr, red);
    return VC_CHECK_LAUNCH();
}
int vc_ew_splitk_reduce_wgrad(const float* slabs, int splits, long long stride, float* dw, long n, int Ci, int taps,
                              unsigned live, hipStream_t s) {
    if (taps <= 0 || taps > 32 || Ci % 4 != 0) live = 0xffffffffu, taps = 1, Ci = 4;
    else if (taps < 32 && live == (1u << taps) - 1u) live = 0xffffffffu;  // every tap computed: no test in the kernel
    if (splits >= 32)
        VC_LAUNCH((splitk_reduce_wgrad_kernel<16>), dim3((unsigned)((n / 4 + 15) / 16)), dim3(256), 0, s, slabs, splits,
                  stride, dw, n / 4, Ci / 4, taps, live);
    else
        VC_LAUNCH((splitk_reduce_wgrad_kernel<4>), dim3((unsigned)((n / 4 + 63) / 64)), dim3(256), 0, s, slabs, splits,
                  stride, dw, n / 4, Ci / 4, taps, live);
    return VC_CHECK_LAUNCH();
}
int vc_ew_bn_bwd_finalize(const float* sums, int C, float count, float* coef, float* dgamma, float* dbeta,
                          hipStream_t s) {
    sums = vc_det_tag(sums);
    VC_LAUNCH(bn_bwd_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, s, sums, C, count, coef, dgamma,
                       dbeta);
    return VC_CHECK_LAUNCH();
}
int vc_ew_bn_bwd_apply(const void* y, const void* dz, const void* aux, const float* params, const float* coef, void* dy,
                       void* gout, long n, int C, int mask, hipStream_t s) {
    const long n8 = n / 8;
    const int g = grid_for(n8);
    const bf16x8 *yy = (const bf16x8*)y, *dd = (const bf16x8*)dz, *aa = (const bf16x8*)aux;
    bf16x8 *oo = (bf16x8*)dy, *gg = (bf16x8*)gout;
#define VC_BWD_APPLY(M, G) \
    VC_LAUNCH((bn_bwd_apply_kernel<M, G>), dim3(g), dim3(256), 0, s, yy, dd, aa, params, coef, oo, gg, n8, C)
    if (gout) {
        if (mask == 0) VC_BWD_APPLY(0, true);
        else if (mask == 1) VC_BWD_APPLY(1, true);
        else if (mask == 3) VC_BWD_APPLY(3, true);
        else VC_BWD_APPLY(2, true);
    } else {
        if (mask == 0) VC_BWD_APPLY(0, false);
        else if (mask == 1) VC_BWD_APPLY(1, false);
        else if (mask == 3) VC_BWD_APPLY(3, false);
        else VC_BWD_APPLY(2, false);
    }
#undef VC_BWD_APPLY
    return VC_CHECK_LAUNCH();
}
int vc_ew_mask_bits(void* x, const uint8_t* bits, long n, hipStream_t s) {
    const long n8 = n / 8;
    VC_LAUNCH(mask_bits_kernel, dim3(grid_for(n8)), dim3(256), 0, s, (bf16x8*)x, bits, n8);
    return VC_CHECK_LAUNCH();
}
int vc_ew_add(const void* a, const void* b, void* out, long n, hipStream_t s) {
    const long n8 = n / 8;
    VC_LAUNCH(add_kernel, dim3(grid_for(n8)), dim3(256), 0, s, (const bf16x8*)a, (const bf16x8*)b, (bf16x8*)out,
                       n8);
    return VC_CHECK_LAUNCH();
}
int vc_ew_maxpool_fwd(const void* y, const float* params, void* out, uint8_t* idx, const VcPoolGeom& g, hipStream_t s) {
    const long total = (long)g.N * g.To * g.Ho * g.Wo * (g.C / 8);
    const int groups = g.C / 8;
    const bool ref_pool = g.kt == 3 && g.kh == 3 && g.kw == 3 && g.st == 2 && g.sh == 2 && g.sw == 2 && g.pt == 1 &&
                          g.ph == 1 && g.pw == 1;
    if (params && ref_pool && getenv("VC_POOL_FWD_GATHER") == nullptr && groups <= 256 && 256 % groups == 0 &&
        g.W * groups <= 1024 && g.Wo * groups <= 512) {
        const size_t lds = (size_t)2 * 3 * g.W * groups * 16;
        const bool small = 3 * g.W * groups <= 256 * 6 && g.Wo * groups <= 256;
        if (lds > 64 * 1024) {
            static VcPerDeviceOnce once;
            once.run([&] {
                (void)hipFuncSetAttribute((const void*)maxpool_bn_tiled_kernel<12, 2>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 3 * 1024 * 16);
            });
        }
        if (small)
            VC_LAUNCH((maxpool_bn_tiled_kernel<6, 1>), dim3((unsigned)(g.N * g.Ho)), dim3(256), lds, s, (const bf16x8*)y,
                      params, (bf16x8*)out, idx, g);
        else
            VC_LAUNCH((maxpool_bn_tiled_kernel<12, 2>), dim3((unsigned)(g.N * g.Ho)), dim3(256), lds, s, (const bf16x8*)y,
                      params, (bf16x8*)out, idx, g);
        return VC_CHECK_LAUNCH();
    }
    if (params)
        VC_LAUNCH((maxpool_fwd_kernel<true>), dim3(grid_for(total)), dim3(256), 0, s, (const bf16x8*)y, params,
                           (bf16x8*)out, idx, g);
    else
        VC_LAUNCH((maxpool_fwd_kernel<false>), dim3(grid_for(total)), dim3(256), 0, s, (const bf16x8*)y, params,
                           (bf16x8*)out, idx, g);
    return VC_CHECK_LAUNCH();
}
int vc_ew_maxpool_bwd(const void* dout, const uint8_t* idx, void* din, const VcPoolGeom& g, hipStream_t s) {
    const long total = (long)g.N * g.T * g.H * g.W * (g.C / 8);
    VC_LAUNCH(maxpool_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, s, (const bf16x8*)dout, idx, (bf16x8*)din,
                       g);
    return VC_CHECK_LAUNCH();
}
int vc_ew_avgpool_fwd(const void* in, float* out, int N, int S, int C, hipStream_t s) {
    VC_LAUNCH(avgpool_fwd_kernel, dim3((N * C + 255) / 256), dim3(256), 0, s, (const bf16*)in, out, N, S, C);
    return VC_CHECK_LAUNCH();
}
int vc_ew_avgpool_bwd(const float* dfeat, void* din, int N, int S, int C, hipStream_t s) {
    VC_LAUNCH(avgpool_bwd_kernel, dim3(grid_for((long)N * S * C)), dim3(256), 0, s, dfeat, (bf16*)din, N, S, C);
    return VC_CHECK_LAUNCH();
}
int vc_ew_linear_fwd(const float* feat, const float* W, const float* bias, float* out, int N, int F, int K,
                     hipStream_t s) {
    static const bool plain = getenv("VC_LINEAR_PLAIN") != nullptr;  // one wave per output (A/B runs)
    if (plain) {
        VC_LAUNCH(linear_fwd_kernel, dim3((N * K + 3) / 4), dim3(256), 0, s, feat, W, bias, out, N, F, K);
    } else {
        const int waves = ((N + 3) / 4) * ((K + 3) / 4);
        VC_LAUNCH(linear_fwd_tiled_kernel, dim3((waves + 3) / 4), dim3(256), 0, s, feat, W, bias, out, N, F, K);
    }
    return VC_CHECK_LAUNCH();
}
int vc_ew_softmax_ce(const float* logits, const long long* y, float* loss, float* dlogits, int N, int K, hipStream_t s) {
    if (K <= 128) VC_LAUNCH((softmax_ce_kernel<2, 4>), dim3(1), dim3(1024), 0, s, logits, y, loss, dlogits, N, K);
    else if (K <= 512) VC_LAUNCH((softmax_ce_kernel<8, 4>), dim3(1), dim3(1024), 0, s, logits, y, loss, dlogits, N, K);
    else if (K <= 1024) VC_LAUNCH((softmax_ce_kernel<16, 2>), dim3(1), dim3(1024), 0, s, logits, y, loss, dlogits, N, K);
    else {
        vc_set_error("softmax_ce: K=%d classes (at most 1024: a row lives in registers)", K);
        return -2;
    }
    return VC_CHECK_LAUNCH();
}
int vc_ew_linear_bwd(const float* dlog, const float* feat, const float* W, float* dfeat, float* dW, float* db, int N,
                     int F, int K, hipStream_t s) {
    if (dfeat)
        VC_LAUNCH(linear_bwd_data_kernel, dim3((N * F + 255) / 256), dim3(256), 0, s, dlog, W, dfeat, N, F, K);
    if (dW)
        VC_LAUNCH(linear_bwd_weight_kernel, dim3((K * F + 255) / 256), dim3(256), 0, s, dlog, feat, dW, db, N, F,
                           K);
    return VC_CHECK_LAUNCH();
}
int vc_ew_stem_pack_input(const void* x, int x_is_bf16, void* xs, int N, int T, int H, int W, hipStream_t s) {
    const long total = (long)N * (T + 6) * (H / 2 + 3) * ((W / 2 + 4) & ~1);
    if (x_is_bf16)
        VC_LAUNCH((stem_pack_input_kernel<bf16>), dim3(grid_for(total)), dim3(256), 0, s, (const bf16*)x,
                           (bf16*)xs, N, T, H, W);
    else
        VC_LAUNCH((stem_pack_input_kernel<float>), dim3(grid_for(total)), dim3(256), 0, s, (const float*)x,
                           (bf16*)xs, N, T, H, W);
    return VC_CHECK_LAUNCH();
}
int vc_ew_clip_pack_u8(const uint8_t* frames, const int* crops, void* xs, int N, int T, int H, int W, int Ts, int Hs,
                       int Ws, int tstep, const float* a, const float* b, hipStream_t s) {
    const long total = (long)N * (T + 6) * (H / 2 + 3) * ((W / 2 + 4) & ~1);
    VC_LAUNCH(clip_pack_u8_kernel, dim3(grid_for(total)), dim3(256), 0, s, frames, crops, (bf16*)xs, N, T, H, W, Ts, Hs,
              Ws, tstep, a[0], a[1], a[2], b[0], b[1], b[2]);
    return VC_CHECK_LAUNCH();
}
int vc_ew_ncs_to_nsc(const void* in, int in_is_bf16, void* out, int N, int C, long S, hipStream_t s) {
    const long total = (long)N * C * S;
    if (in_is_bf16)
        VC_LAUNCH((ncs_to_nsc_kernel<bf16>), dim3(grid_for(total)), dim3(256), 0, s, (const bf16*)in, (bf16*)out,
                           N, C, S);
    else
        VC_LAUNCH((ncs_to_nsc_kernel<float>), dim3(grid_for(total)), dim3(256), 0, s, (const float*)in,
                           (bf16*)out, N, C, S);
    return VC_CHECK_LAUNCH();
}
int vc_ew_nsc_to_ncs(const void* in, void* out, int out_is_bf16, int N, int C, long S, hipStream_t s) {
    const long total = (long)N * C * S;
    if (out_is_bf16)
        VC_LAUNCH((nsc_to_ncs_kernel<bf16>), dim3(grid_for(total)), dim3(256), 0, s, (const bf16*)in, (bf16*)out,
                           N, C, S);
    else
        VC_LAUNCH((nsc_to_ncs_kernel<float>), dim3(grid_for(total)), dim3(256), 0, s, (const bf16*)in,
                           (float*)out, N, C, S);
    return VC_CHECK_LAUNCH();
}
int vc_ew_pack_weight(const float* w, void* wf, void* wd, int Co, int taps, int Ci, hipStream_t s) {
    VC_LAUNCH(pack_weight_kernel, dim3(grid_for((long)Co * taps * Ci)), dim3(256), 0, s, w, (bf16*)wf, (bf16*)wd,
                       Co, taps, Ci);
    return VC_CHECK_LAUNCH();
}
int vc_ew_stem_pack_weight(const float* w, void* ws, int Co, hipStream_t s) {
    VC_LAUNCH(stem_pack_weight_kernel, dim3((Co * 1344 + 255) / 256), dim3(256), 0, s, w, (bf16*)ws, Co);
    return VC_CHECK_LAUNCH();
}
int vc_ew_stem_unpack_wgrad(const float* dws, float* dw, int Co, hipStream_t s) {
    VC_LAUNCH(stem_unpack_wgrad_kernel, dim3((Co * 1344 + 255) / 256), dim3(256), 0, s, dws, dw, Co);
    return VC_CHECK_LAUNCH();
}

int vc_ew_sgd_step(float* p, const float* g, float* mom, void* shadow, long n, float lr, float momentum, float wd,
                   int nesterov, hipStream_t s) {
    VC_LAUNCH(sgd_step_kernel, dim3(grid_for(n / 4 + 1, 256, 2048)), dim3(256), 0, s, p, g, mom, (bf16*)shadow, n, lr,
              momentum, wd, nesterov);
    return VC_CHECK_LAUNCH();
}
int vc_ew_adam_step(float* p, const float* g, float* m, float* v, void* shadow, long n, float step_size, float b1, float b2,
                    float eps, float wd, float bias2_sqrt, hipStream_t s) {
    VC_LAUNCH(adam_step_kernel, dim3(grid_for(n / 4 + 1, 256, 2048)), dim3(256), 0, s, p, g, m, v, (bf16*)shadow, n,
              step_size, b1, b2, eps, wd, bias2_sqrt);
    return VC_CHECK_LAUNCH();
}
int vc_ew_shadow(const float* p, void* shadow, long n, hipStream_t s) {
    VC_LAUNCH(shadow_kernel, dim3(grid_for(n, 256, 2048)), dim3(256), 0, s, p, (bf16*)shadow, n);
    return VC_CHECK_LAUNCH();
}
int vc_ew_pack_dgrad_batched(const void* shadow, void* wd, const int* table, int nconv, int tiles, hipStream_t s) {
    VC_LAUNCH(pack_dgrad_batched_kernel, dim3(tiles), dim3(256), 0, s, (const bf16*)shadow, (bf16*)wd, table, nconv);
    return VC_CHECK_LAUNCH();
}

int vc_ew_bn_bwd_reduce_pooled(const void* y, const void* dout, const uint8_t* idx, const void* pooled, const float* params, float* sums,
                               const VcPoolGeom& g, const VcBnBwdFin* fin, hipStream_t s) {
    const long total = (long)g.N * g.To * g.Ho * g.Wo * (g.C / 8);
    const int groups = g.C / 8;
    sums = vc_det_tag(sums);
    if (pooled != nullptr && groups <= 256 && 256 % groups == 0 && g.C % 64 == 0 && getenv("VC_POOLRED_GATHER") == nullptr) {
        // the sums are a plain streaming reduction over the two pooled tensors (chan_reduce_kernel: 5-6 TB/s); the one-
        // kernel form below spent 117 us on 0.1 GB, most of it in its 262 k float atomics.  A small second launch takes
        // the channels that cannot be solved from z (normally none) and runs the finalize after both.
        const int rc = vc_ew_bn_bwd_reduce(pooled, dout, nullptr, params, sums, total / groups, g.C, 4, nullptr, s);
        if (rc) return rc;
        VC_LAUNCH(bn_bwd_reduce_pooled_kernel, dim3(grid_for(total, 256, 256)), dim3(256), 0, s, (const bf16*)y,
                  (const bf16x8*)dout, idx, (const bf16x8*)pooled, params, sums, g, fin ? *fin : VcBnBwdFin{}, 1);
        return VC_CHECK_LAUNCH();
    }
    VC_LAUNCH(bn_bwd_reduce_pooled_kernel, dim3(grid_for(total, 256, 2048)), dim3(256), 0, s, (const bf16*)y,
              (const bf16x8*)dout, idx, (const bf16x8*)pooled, params, sums, g, fin ? *fin : VcBnBwdFin{}, 0);
    return VC_CHECK_LAUNCH();
}
int vc_ew_bn_bwd_apply_pooled(const void* y, const void* dout, const uint8_t* idx, const float* params, const float* coef,
                              void* dy, const VcPoolGeom& g, hipStream_t s) {
    const int nrows = g.N * g.T * g.H;
    const int groups = g.C / 8;
    const size_t lds = (size_t)4 * g.Wo * groups * 24;
    const bool ref_pool = g.kt == 3 && g.kh == 3 && g.kw == 3 && g.st == 2 && g.sh == 2 && g.sw == 2 && g.pt == 1 &&
                          g.ph == 1 && g.pw == 1;
    const bool rowwise = getenv("VC_POOL_BWD_ROWWISE") != nullptr;  // the row-at-a-time kernel (A/B runs, tests)
    if (ref_pool && !rowwise && groups <= 256 && 256 % groups == 0 && lds <= 64 * 1024) {
        const int nbt = (g.T >> 1) + 1, nbh = (g.H >> 1) + 1;
        const int trips = ((g.W + 1) / 2 + 256 / groups - 1) / (256 / groups);  // per parity and row
        const dim3 grid((unsigned)(g.N * nbt * nbh));
        if (trips == 1 && getenv("VC_POOL_BWD_NO_PF") == nullptr)
            VC_LAUNCH(bn_bwd_apply_pooled_pf_kernel<1>, grid, dim3(256), lds, s, (const bf16x8*)y, (const bf16x8*)dout, idx,
                      params, coef, (bf16x8*)dy, g, nbt, nbh);
        else if (trips == 2 && getenv("VC_POOL_BWD_NO_PF") == nullptr)
            VC_LAUNCH(bn_bwd_apply_pooled_pf_kernel<2>, grid, dim3(256), lds, s, (const bf16x8*)y, (const bf16x8*)dout, idx,
                      params, coef, (bf16x8*)dy, g, nbt, nbh);
        else
            VC_LAUNCH(bn_bwd_apply_pooled_tiled_kernel, grid, dim3(256), lds, s, (const bf16x8*)y, (const bf16x8*)dout, idx,
                      params, coef, (bf16x8*)dy, g, nbt, nbh);
        return VC_CHECK_LAUNCH();
    }
    VC_LAUNCH(bn_bwd_apply_pooled_kernel, dim3(nrows < 4096 ? nrows : 4096), dim3(256), 0, s, (const bf16x8*)y,
              (const bf16x8*)dout, idx, params, coef, (bf16x8*)dy, g);
    return VC_CHECK_LAUNCH();
}
