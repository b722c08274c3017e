// Conv3d weight gradient for 3x3x3 / stride 1 / pad 1 with the nine (dh,dw) taps of one temporal
// offset sharing ONE activation image in LDS (gfx950).
//
// dW[co][tap][ci] = sum_m dY[m][co] * X[m + shift(tap)][ci],  shift = dt*H*W + dh*W + dw.
// The plain weight-gradient kernel (wgrad_conv.hip) stages a separate 64-row X tile per tap: 40 KB of
// global->LDS traffic per 64 rows and 4 taps, which is what bounds it (~22 % of the MFMA peak).  For a
// fixed dt the nine (dh,dw) taps of CH consecutive rows m read the CONTIGUOUS row range
// [m0 + dt*HW - W - 1, m0 + CH + dt*HW + W + 1): one image of CH + 2W + 2 rows serves all nine, so a chunk
// moves (2*CH + 2W + 2) * 128 bytes for 9 taps: about 2.5x less per MAC.
//
// Work split: blockIdx.x = (dt, 64-wide ci chunk), blockIdx.y = 64-wide co chunk, blockIdx.z = slice of m.
// 8 waves: wave w owns co half (w & 1) x ci quarter (w >> 1) x all 9 taps = 18 MFMA tiles of 16x16
// (72 accumulator registers); per 32-row k-step it reads 2 dY^T and 9 X^T fragments.  Both operand tiles
// stay row-major [m][channel] in LDS as they arrive and the fragments come from ds_read_b64_tr_b16 (the
// transposing LDS read), exactly as in wgrad_conv.hip; a tap is a constant row shift of the X fragment.
// Rows a tap must not see (outside the sample: zero padding) are redirected to a zero row, from a 9-bit
// mask per row m computed by the thread that stages that row.
// Staging is register double-buffered (global loads of chunk c+1 are issued before the MFMAs of chunk c
// and written to the other LDS buffer after them): one barrier per chunk.
#include <stdlib.h>

#include "common.h"
#include "kernels.h"

namespace {

constexpr int RS = 160;   // bytes per LDS row: 128 data + 32 pad (conflict-free transposed reads)
constexpr int CH = 128;   // rows of m per chunk
constexpr int NT = 512;   // threads
constexpr int MAXSEG = 6; // 16-byte staging loads per thread per chunk (CH + XR <= 384 rows)

typedef s16x4 __attribute__((address_space(3))) * lds_s16x4_p;

__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(2, 2))) void wgrad9_kernel(const VcWgrad9Params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nci = p.Cs >> 6;
    const int dtg = p.dt0 + blockIdx.x / nci;  // 0..2  -> dt = dtg - 1
    const int cic = blockIdx.x % nci;
    const int coc = blockIdx.y;
    const int HW = p.H * p.W;
    const int XR = CH + 2 * p.W + 2;            // image rows per chunk
    const int dt = dtg - 1;
    const long Mtot = (long)p.NP * HW;

    // LDS: [zero row][buf0: sY | sX | sM][buf1: ...]
    const int BUF = (CH + XR) * RS + CH * 48;  // rows + the per-(row, tap) address table [CH][12] dwords
    char* zrow = smem;
    auto sY = [&](int b) { return smem + RS + b * BUF; };
    auto sX = [&](int b) { return smem + RS + b * BUF + CH * RS; };
    auto sT = [&](int b) { return (unsigned*)(smem + RS + b * BUF + (CH + XR) * RS); };
    if (tid < RS / 4) ((unsigned*)zrow)[tid] = 0u;

    const long mb = (long)blockIdx.z * p.rows_per_split;
    const long mend = mb + p.rows_per_split < Mtot ? mb + p.rows_per_split : Mtot;
    const int nchunks = (int)((mend - mb + CH - 1) / CH);

    // ---- staging: 16-byte segment e = tid + NT*i of the (CH + XR) x 8 segments of a chunk
    const int nseg = (CH + XR) * 8;
    bf16x8 st[MAXSEG];
    unsigned sttab[3] = {0, 0, 0};  // address-table entries of (row tid % CH, taps 3*(tid / CH) ..), image-relative
    unsigned stvalid = 0; // bit i: st[i] holds real data (else the segment is stored as zeros)
    auto load_chunk = [&](int c) {
        const long m0 = mb + (long)c * CH;
        // every load is issued unconditionally from a clamped address and zeroed at store time: a
        // load under a per-element branch makes hipcc wait for it before the next one is issued
        stvalid = 0;
#pragma unroll
        for (int i = 0; i < MAXSEG; ++i) {
            const int e = tid + NT * i;
            const int row = e >> 3, part = e & 7;
            const long m = m0 + row;
            const long g = m0 + (long)dt * HW - (p.W + 1) + (row - CH);
            const bool isy = row < CH;
            const bool ok = e < nseg && (isy ? m < mend : (g >= 0 && g < Mtot));
            const bf16* src = isy ? p.dy + ((ok ? m : 0) * p.Co + coc * 64 + part * 8)
                                  : p.x + ((ok ? g : 0) * p.Cs + cic * 64 + part * 8);
            st[i] = *(const bf16x8*)src;
            stvalid |= ok ? (1u << i) : 0u;
        }
        if (tid < 3 * CH) {
            // Address table of the chunk: for row r = tid % CH and taps j = 3g .. 3g+2 (g = tid / CH, i.e. one
            // dh), the LDS byte offset the X^T fragment of (r, j) is read from: its image row, or the zero
            // row when (t+dt, h+dh, w+dw) lies outside the sample.  Computed once per chunk by the staging
            // threads, so the MFMA loop spends one add per read instead of a test and a select.
            const int r = tid & (CH - 1), g = tid >> 7;  // CH == 128
            const long m = m0 + r;
            unsigned vw = 0;
            bool rowok = false;
            if (m < mend) {
                const unsigned P = vc_div((unsigned)m, p.dHW);
                const int hw = (int)(m - (long)P * HW);
                const unsigned n = vc_div(P, p.dT);
                const int t = (int)(P - n * p.T);
                const int h = (int)vc_div((unsigned)hw, p.dW);
                const int w = hw - h * p.W;
                rowok = (unsigned)(t + dt) < (unsigned)p.T && (unsigned)(h + g - 1) < (unsigned)p.H;
#pragma unroll
                for (int k = 0; k < 3; ++k) vw |= ((unsigned)(w + k - 1) < (unsigned)p.W) ? (1u << k) : 0u;
            }
#pragma unroll
            for (int k = 0; k < 3; ++k)
                sttab[k] = (rowok && ((vw >> k) & 1u)) ? (unsigned)((r + p.W + 1 + (g - 1) * p.W + (k - 1)) * RS) : 0xffffffffu;
        }
    };
    auto store_chunk = [&](int b) {
        char* base = sY(b);  // sY and sX are contiguous: row index runs over CH + XR rows
#pragma unroll
        for (int i = 0; i < MAXSEG; ++i) {
            const int e = tid + NT * i;
            const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
            if (e < nseg) *(bf16x8*)(base + (e >> 3) * RS + (e & 7) * 16) = ((stvalid >> i) & 1u) ? st[i] : z;
        }
        if (tid < 3 * CH) {
            const unsigned xb = (unsigned)(sX(b) - smem);
            unsigned* trow = sT(b) + (tid & (CH - 1)) * 12 + (tid >> 7) * 3;
#pragma unroll
            for (int k = 0; k < 3; ++k) trow[k] = sttab[k] == 0xffffffffu ? 0u : xb + sttab[k];  // 0 = the zero row
        }
    };

    // ---- wave roles
    const int ch = wave & 1;    // co half: channels [32ch, 32ch+32)
    const int cib = wave >> 1;  // ci quarter: channels [16cib, 16cib+16)
    const int l15 = lane & 15;
    const int lg = lane >> 4;
    const int q = l15 >> 2, pp = l15 & 3;
    const int krow = 4 * lg + q;                       // this lane's row inside a 16-row block
    const unsigned yoff = (unsigned)krow * RS + (unsigned)(ch * 32 + 4 * pp) * 2;

    f32x4 acc[9][2];
#pragma unroll
    for (int j = 0; j < 9; ++j)
#pragma unroll
        for (int a = 0; a < 2; ++a) acc[j][a] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if (nchunks > 0) {
        load_chunk(0);
        store_chunk(0);
    }
    __syncthreads();

    long long tq[5] = {0, 0, 0, 0, 0};
    long long t_prev = (long long)__builtin_amdgcn_s_memtime();
    const long long t_begin = t_prev;
#define VC_W9_LAP(k)                                                  \
    do {                                                              \
        if (p.dbg != nullptr) {                                       \
            const long long now = (long long)__builtin_amdgcn_s_memtime(); \
            tq[k] += now - t_prev;                                    \
            t_prev = now;                                             \
        }                                                             \
    } while (0)
    for (int c = 0; c < nchunks; ++c) {
        const int b = c & 1;
        const bool more = c + 1 < nchunks;
        if (more) load_chunk(c + 1);
        VC_W9_LAP(0);
        const unsigned ybase = (unsigned)(sY(b) - smem);
        const unsigned* tab = sT(b);
        const unsigned xcol = (unsigned)(cib * 16 + 4 * pp) * 2;  // this lane's column inside an image (or zero) row
#pragma unroll
        for (int ks = 0; ks < CH / 32; ++ks) {
            // the 9 fragment addresses of this lane's two rows of the k-step: 2 x (two b128 + one b32)
            const unsigned* tlo = tab + (ks * 32 + krow) * 12;
            const unsigned* thi = tlo + 16 * 12;
            unsigned alo[9], ahi[9];
            {
                const uint4 a0 = *(const uint4*)tlo, a1 = *(const uint4*)(tlo + 4);
                const uint4 b0 = *(const uint4*)thi, b1 = *(const uint4*)(thi + 4);
                alo[0] = a0.x, alo[1] = a0.y, alo[2] = a0.z, alo[3] = a0.w, alo[4] = a1.x, alo[5] = a1.y, alo[6] = a1.z,
                alo[7] = a1.w, alo[8] = tlo[8];
                ahi[0] = b0.x, ahi[1] = b0.y, ahi[2] = b0.z, ahi[3] = b0.w, ahi[4] = b1.x, ahi[5] = b1.y, ahi[6] = b1.z,
                ahi[7] = b1.w, ahi[8] = thi[8];
            }
            bf16x8 fa[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const unsigned r0 = ybase + yoff + (unsigned)(ks * 32) * RS + a * 32;
                s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(smem + r0));
                s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(smem + r0 + 16 * RS));
                fa[a] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
#pragma unroll
            for (int j = 0; j < 9; ++j) {
                s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(smem + (alo[j] + xcol)));
                s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(smem + (ahi[j] + xcol)));
                const bf16x8 fb = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
                for (int a = 0; a < 2; ++a)
                    acc[j][a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[a], fb, acc[j][a], 0, 0, 0);
            }
        }
        // Pin the LDS-read / MFMA interleave of the whole chunk (one basic block): the reads run about three
        // taps ahead of the MFMAs that consume them.  Left alone the compiler issues each tap's two
        // transposed reads right before its two MFMAs and the in-order wave eats the LDS latency 36 times.
        __builtin_amdgcn_sched_group_barrier(0x100, 16, 0);  // k-step 0: 6 table + 4 dY^T + 3 taps
#pragma unroll
        for (int ks = 0; ks < CH / 32; ++ks)
#pragma unroll
            for (int j = 0; j < 9; ++j) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                if (j < 1) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                else __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
            }
        VC_W9_LAP(1);
        if (more) store_chunk(b ^ 1);  // the other buffer was last read before the previous barrier
        VC_W9_LAP(2);
        __syncthreads();
        VC_W9_LAP(3);
    }
    if (p.dbg != nullptr && tid == 0) {
        long long* d = p.dbg + ((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8;
        d[0] = tq[0], d[1] = tq[1], d[2] = tq[2], d[3] = tq[3], d[4] = t_prev - t_begin, d[5] = nchunks, d[6] = t_begin;
    }

    // ---- epilogue: acc[j][a][jj] = dW[co = coc*64 + ch*32 + a*16 + lg*4 + jj][tap = dtg*9 + j][ci = cic*64 + cib*16 + l15]
    float* dst = p.dw;
    int how = 1;  // 1: += (this workgroup owns the elements), 2: store into its slab, 0: atomics
    if (p.splits > 1) {
        if (p.partial != nullptr) {
            dst = p.partial + (size_t)blockIdx.z * p.partial_stride;
            how = 2;
        } else {
            how = 0;
        }
    }
#pragma unroll
    for (int j = 0; j < 9; ++j)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                float* e = dst + (size_t)(coc * 64 + ch * 32 + a * 16 + lg * 4 + jj) * p.Kw + (size_t)(dtg * 9 + j) * p.Cs +
                           cic * 64 + cib * 16 + l15;
                if (how == 0) atomicAdd(e, acc[j][a][jj]);
                else if (how == 1) *e += acc[j][a][jj];
                else *e = acc[j][a][jj];
            }
}

}  // namespace

// LDS bytes for image width W; 0 if the chunk does not fit (then the plain kernel is used)
static int wgrad9_lds(int W) {
    const int XR = CH + 2 * W + 2;
    if ((CH + XR) * 8 > NT * MAXSEG) return 0;
    const int lds = RS + 2 * ((CH + XR) * RS + CH * 48);
    return lds <= 160 * 1024 ? lds : 0;
}

int vc_wgrad9_plan(long M, int W, int Cs, int Co, int ndt, int* rows_per_split) {
    if (wgrad9_lds(W) == 0) return 0;
    const long tiles = (long)ndt * (Cs / 64) * (Co / 64);
    static const long target = getenv("VC_WGRAD9_TARGET") ? atol(getenv("VC_WGRAD9_TARGET")) : 256;
    long splits = (target + tiles / 2) / tiles;
    const long maxs = (M + 4 * CH - 1) / (4 * CH);  // at least 4 chunks per slice
    if (splits > maxs) splits = maxs;
    if (splits < 1) splits = 1;
    long rps = (M + splits - 1) / splits;
    rps = (rps + CH - 1) / CH * CH;
    splits = (M + rps - 1) / rps;
    *rows_per_split = (int)rps;
    return (int)splits;
}

static long long* g_w9_dbg = nullptr;
extern "C" void vc_debug_wgrad9_stamps(long long* buf) { g_w9_dbg = buf; }  // tuning aid

int vc_launch_wgrad9(VcWgrad9Params& p, hipStream_t stream) {
    p.dbg = g_w9_dbg;
    const int lds = wgrad9_lds(p.W);
    if (lds == 0 || p.Cs % 64 != 0 || p.Co % 64 != 0) {
        vc_set_error("wgrad9: unsupported geometry (W=%d Cs=%d Co=%d)", p.W, p.Cs, p.Co);
        return -2;
    }
    const int HW = p.H * p.W;
    p.dHW = vc_make_fastdiv(HW), p.dT = vc_make_fastdiv(p.T), p.dW = vc_make_fastdiv(p.W);
    static VcPerDeviceOnce once;  // thread-safe, once per device
    once.run([&]() { (void)hipFuncSetAttribute((const void*)wgrad9_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
    if (p.ndt < 1 || p.ndt > 3) p.dt0 = 0, p.ndt = 3;
    dim3 grid(p.ndt * (p.Cs / 64), p.Co / 64, p.splits);
    (void)hipGetLastError();
    hipLaunchKernelGGL(wgrad9_kernel, grid, dim3(NT), lds, stream, p);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        vc_set_error("wgrad9 launch failed: %s", hipGetErrorString(e));
        return -1;
    }
    return 0;
}
