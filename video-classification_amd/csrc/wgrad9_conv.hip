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
// Staging is register double-buffered and runs one chunk further ahead than the LDS buffers: during the MFMA phase
// of chunk c the registers loaded a phase ago (chunk c+1) go into the other LDS buffer first -- they landed long
// ago, no wait -- and the global loads of chunk c+2 are issued right after, with a whole phase to land.  One barrier
// per chunk.  The phases outside the MFMAs are exposed (the waves of a workgroup are in the same phase, and the two
// waves of a SIMD serialise their VALU work), so the staging code is kept lean: 32-bit offsets advanced by a constant
// per chunk, 32-bit chunk-relative validity tests, no branches at all (every thread stores all its segments, the
// buffers are sized for that; zeros made with a mask; the tail iterations stage harmlessly).  Placing the staging
// between the k-steps of the MFMA stream instead of in front of it was tried and is slower (195k against 185k
// clocks per layer1 workgroup: it breaks the pinned read / MFMA interleave).
//
// Two forms of the same kernel: eight waves with 128-row chunks (one workgroup per CU; wave = co half x ci quarter,
// 18 MFMA tiles) and four waves with 64-row chunks (two workgroups per CU, which drift apart so that one computes
// while the other stages; wave = co half x ci half, 36 tiles).
#include <stdlib.h>

#include "common.h"
#include "kernels.h"

namespace {

constexpr int RS = 160;   // bytes per LDS row: 128 data + 32 pad (conflict-free transposed reads)
constexpr int MAXSEG = 6; // 16-byte staging loads per thread per chunk

typedef s16x4 __attribute__((address_space(3))) * lds_s16x4_p;
typedef __attribute__((ext_vector_type(4))) unsigned int w9_u32x4;

template <int NW>
struct W9Cfg {
    static constexpr int NT = NW * 64;          // threads
    static constexpr int CH = NW * 16;          // rows of m per chunk (128 / 64)
    static constexpr int NCI = NW == 8 ? 1 : 2; // ci tiles (16 channels) per wave
    static constexpr int ROWS = NT * MAXSEG / 8;  // staged rows per buffer (every thread stores all its segments)
    static constexpr int BUF = ROWS * RS + CH * 48;  // + the per-(row, tap) address table [CH][12] dwords
    static constexpr int LDS = RS + 2 * BUF;
    static_assert(CH % 64 == 0, "segments 0 .. CH/(NT/8)-1 of a thread are dY rows");
};

template <int NW>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(2, 2))) void wgrad9_kernel(const VcWgrad9Params p) {
    using Cfg = W9Cfg<NW>;
    constexpr int NT = Cfg::NT, CH = Cfg::CH, NCI = Cfg::NCI;
    constexpr int NYSEG = CH / (NT / 8);  // a thread's first NYSEG segments are dY rows, the rest image rows (= 2)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nci = p.Cs >> 6;
    // Workgroups go round-robin to the 8 XCDs (linear id mod 8), each with its own L2, and the workgroups of one slice
    // of m -- three dt offsets, the ci and co chunks -- read the same dY rows and overlapping image rows: XCD k takes
    // the k-th contiguous share of the (z, y, x) order, so that a slice's workgroups meet in one L2.
    unsigned bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (p.xcd_map) {
        const unsigned total = gridDim.x * gridDim.y * gridDim.z;
        const unsigned lin = bx + gridDim.x * (by + gridDim.y * bz);
        const unsigned k = lin & 7u, base = total >> 3, rem = total & 7u;
        const unsigned logical = k * base + (k < rem ? k : rem) + (lin >> 3);
        bx = logical % gridDim.x;
        const unsigned t = logical / gridDim.x;
        by = t % gridDim.y;
        bz = t / gridDim.y;
    }
    const int dtg = p.dt0 + (int)bx / nci;  // 0..2  -> dt = dtg - 1
    const int cic = (int)bx % nci;
    const int coc = (int)by;
    const int HW = p.H * p.W;
    const int XR = CH + 2 * p.W + 2;            // image rows per chunk
    const int dt = dtg - 1;
    const long Mtot = (long)p.NP * HW;

    // LDS: [zero row][buf0: sY | sX ... | table][buf1: ...]
    char* zrow = smem;
    auto sY = [&](int b) { return smem + RS + b * Cfg::BUF; };
    auto sX = [&](int b) { return smem + RS + b * Cfg::BUF + CH * RS; };
    auto sT = [&](int b) { return (unsigned*)(smem + RS + b * Cfg::BUF + Cfg::ROWS * RS); };
    if (tid < RS / 4) ((unsigned*)zrow)[tid] = 0u;

    const long mb = (long)bz * p.rows_per_split;
    const long mend = mb + p.rows_per_split < Mtot ? mb + p.rows_per_split : Mtot;
    const int nchunks = (int)((mend - mb + CH - 1) / CH);

    // ---- staging: 16-byte segment e = tid + NT*i; rows [0, CH) are dY rows m = mb + c*CH + row, rows [CH, CH + XR)
    // image rows g = mb + c*CH + dt*HW - (W+1) + (row - CH)
    bf16x8 st[MAXSEG];
    unsigned sttab[3] = {0, 0, 0};  // address-table entries of (row tid % CH, taps 3*(tid / CH) ..), image-relative
    unsigned stvalid = 0;           // bit i: st[i] holds real data (else the segment is stored as zeros)
    unsigned soff[MAXSEG];          // byte offset of the NEXT chunk's element from its tensor base (< 2^32: see capi)
    const int row0 = tid >> 3, part = tid & 7;
    const long gb = mb + (long)dt * HW - (p.W + 1);  // image row 0 of chunk 0
    auto sat = [](long v) { return (int)(v < -(1l << 30) ? -(1l << 30) : (v > (1l << 30) ? (1l << 30) : v)); };
    const int ylim = sat(mend - mb);              // dY row v = row + c*CH is real iff v < ylim
    const int xlo = sat(-gb), xhi = sat(Mtot - gb);  // image row v = (row - CH) + c*CH is real iff xlo <= v < xhi
    unsigned inimg = 0;                           // bit i: segment i lies inside the chunk's CH + XR rows
#pragma unroll
    for (int i = 0; i < MAXSEG; ++i) {
        const int row = row0 + (NT / 8) * i;
        if (row < CH + XR) inimg |= 1u << i;
        const long r = i < NYSEG ? mb + row : gb + (row - CH);
        soff[i] = i < NYSEG ? (unsigned)((r * p.Co + coc * 64 + part * 8) * 2) : (unsigned)((r * p.Cs + cic * 64 + part * 8) * 2);
    }
    const char* const ybase_g = (const char*)p.dy;
    const char* const xbase_g = (const char*)p.x;
    const unsigned ystep = (unsigned)(CH * p.Co * 2), xstep = (unsigned)(CH * p.Cs * 2);
    int cnext = 0;  // chunk the next load_chunk() fetches (calls come in chunk order)
    auto load_chunk = [&]() {
        const int c = cnext++;
        // every load is issued unconditionally (invalid ones from offset 0) and zeroed at store time
        stvalid = 0;
#pragma unroll
        for (int i = 0; i < MAXSEG; ++i) {
            const int v = row0 + (NT / 8) * i - (i < NYSEG ? 0 : CH) + c * CH;
            bool ok = i < NYSEG ? v < ylim : (unsigned)(v - xlo) < (unsigned)(xhi - xlo);
            ok = ok && ((inimg >> i) & 1u);
            const unsigned off = ok ? soff[i] : 0u;
            st[i] = *(const bf16x8*)((i < NYSEG ? ybase_g : xbase_g) + off);
            soff[i] += i < NYSEG ? ystep : xstep;
            stvalid |= ok ? (1u << i) : 0u;
        }
        if (tid < 3 * CH) {  // (whole waves: 3 * CH is a multiple of 64)
            // Address table of the chunk: for row r = tid % CH and taps j = 3g .. 3g+2 (g = tid / CH, i.e. one
            // dh), the LDS byte offset the X^T fragment of (r, j) is read from: its image row, or the zero
            // row when (t+dt, h+dh, w+dw) lies outside the sample.  Computed once per chunk by the staging
            // threads, so the MFMA loop spends one add per read instead of a test and a select.
            const int r = tid & (CH - 1), g = tid / CH;
            const long m = mb + (long)c * CH + r;
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
        char* base = sY(b) + row0 * RS + part * 16;  // sY and sX are contiguous: the row index runs over all staged rows
#pragma unroll
        for (int i = 0; i < MAXSEG; ++i) {
            const unsigned m = 0u - ((stvalid >> i) & 1u);
            w9_u32x4 v = __builtin_bit_cast(w9_u32x4, st[i]);
            v &= m;
            *(w9_u32x4*)(base + (NT / 8) * i * RS) = v;
        }
        if (tid < 3 * CH) {
            const unsigned xb = (unsigned)(sX(b) - smem);
            unsigned* trow = sT(b) + (tid & (CH - 1)) * 12 + (tid / CH) * 3;
#pragma unroll
            for (int k = 0; k < 3; ++k) trow[k] = sttab[k] == 0xffffffffu ? 0u : xb + sttab[k];  // 0 = the zero row
        }
    };

    // ---- wave roles: co half (32 channels = 2 tiles) x NCI ci tiles of 16 x 9 taps
    const int ch = wave & 1;     // co half: channels [32ch, 32ch+32)
    const int ciw = wave >> 1;   // ci slice: channels [16 NCI ciw, 16 NCI (ciw+1))
    const int l15 = lane & 15;
    const int lg = lane >> 4;
    const int q = l15 >> 2, pp = l15 & 3;
    const int krow = 4 * lg + q;                       // this lane's row inside a 16-row block
    const unsigned yoff = (unsigned)krow * RS + (unsigned)(ch * 32 + 4 * pp) * 2;

    f32x4 acc[9][2][NCI];  // [tap][co tile][ci tile]
#pragma unroll
    for (int j = 0; j < 9; ++j)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int bc = 0; bc < NCI; ++bc) acc[j][a][bc] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if (nchunks > 0) {
        load_chunk();
        store_chunk(0);
        if (nchunks > 1) load_chunk();  // stays in registers until chunk 0's phase stores it
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
        // chunk c+1 (loaded during the previous phase) goes into the other buffer, which was last read before the
        // previous barrier; then the loads of chunk c+2 go out
#ifndef VC_W9_STAMPS
        // unconditional (no branches, one basic block with the MFMAs: 204k -> 185k clocks per workgroup at layer1)
        store_chunk(b ^ 1);  // past the last chunk: stale registers into a buffer nobody reads
        load_chunk();        // past the last chunk: every segment invalid, loads from offset 0
#else  // per-phase stamps (scripts/wgrad9_stamps.py; build with -DVC_W9_STAMPS)
        if (c + 1 < nchunks) store_chunk(b ^ 1);
        VC_W9_LAP(2);
        if (c + 2 < nchunks) load_chunk();
        VC_W9_LAP(0);
#endif
        const unsigned ybase = (unsigned)(sY(b) - smem);
        const unsigned* tab = sT(b);
        const unsigned xcol = (unsigned)(ciw * 16 * NCI + 4 * pp) * 2;  // this lane's column inside an image (or zero) row
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
            for (int j = 0; j < 9; ++j)
#pragma unroll
                for (int bc = 0; bc < NCI; ++bc) {
                    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(smem + (alo[j] + xcol + bc * 32)));
                    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(smem + (ahi[j] + xcol + bc * 32)));
                    const bf16x8 fb = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
                    for (int a = 0; a < 2; ++a)
                        acc[j][a][bc] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[a], fb, acc[j][a][bc], 0, 0, 0);
                }
        }
        // Pin the LDS-read / MFMA interleave of the whole chunk (one basic block): the reads run about three
        // fragments ahead of the MFMAs that consume them.  Left alone the compiler issues each fragment's two
        // transposed reads right before its two MFMAs and the in-order wave eats the LDS latency every time.
        __builtin_amdgcn_sched_group_barrier(0x100, 16, 0);  // k-step 0: 6 table + 4 dY^T + 3 fragments
#pragma unroll
        for (int ks = 0; ks < CH / 32; ++ks)
#pragma unroll
            for (int g2 = 0; g2 < 9 * NCI; ++g2) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                if (NCI == 1) {
                    if (g2 < 1) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                    else __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
                } else {
                    if (g2 % 3 == 0) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
                    else __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                }
            }
#ifdef VC_W9_STAMPS
        VC_W9_LAP(1);
#endif
        __syncthreads();
        VC_W9_LAP(3);
    }
    if (p.dbg != nullptr && tid == 0) {
        long long* d = p.dbg + ((size_t)(bz * gridDim.y + by) * gridDim.x + bx) * 8;
        d[0] = tq[0], d[1] = tq[1], d[2] = tq[2], d[3] = tq[3], d[4] = t_prev - t_begin, d[5] = nchunks, d[6] = t_begin;
    }

    // ---- epilogue: acc[j][a][bc][jj] =
    // dW[co = coc*64 + ch*32 + a*16 + lg*4 + jj][tap = dtg*9 + j][ci = cic*64 + (ciw*NCI + bc)*16 + l15]
    float* dst = p.dw;
    int how = 1;  // 1: += (this workgroup owns the elements), 2: store into its slab, 0: atomics
    if (p.splits > 1) {
        if (p.partial != nullptr) {
            dst = p.partial + (size_t)bz * p.partial_stride;
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
            for (int bc = 0; bc < NCI; ++bc)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    float* e = dst + (size_t)(coc * 64 + ch * 32 + a * 16 + lg * 4 + jj) * p.Kw + (size_t)(dtg * 9 + j) * p.Cs +
                               cic * 64 + (ciw * NCI + bc) * 16 + l15;
                    if (how == 0) atomicAdd(e, acc[j][a][bc][jj]);
                    else if (how == 1) *e += acc[j][a][bc][jj];
                    else *e = acc[j][a][bc][jj];
                }
}

}  // namespace

// LDS bytes of the NW-wave form for image width W; 0 if a chunk's rows do not fit its staging registers
template <int NW>
static int wgrad9_lds(int W) {
    using Cfg = W9Cfg<NW>;
    const int XR = Cfg::CH + 2 * W + 2;
    if (Cfg::CH + XR > Cfg::ROWS) return 0;
    return Cfg::LDS <= 160 * 1024 ? Cfg::LDS : 0;
}
// The four-wave form needs two workgroups per CU, i.e. twice the slices of m and twice the slabs.  Measured (layer1,
// 64 x 8 x 28 x 28 x 64): its chunk loop is 28 % shorter (147k against 204k clocks per workgroup: the two workgroups of a
// CU drift apart and one computes while the other stages) but the second set of slabs (75 MB more to write and
// reduce) takes it back: 0.121 against 0.123 ms with the reduce, and the deeper layers lose 5-10 %.  Opt-in
// (VC_WGRAD9_4WAVE=1) until the slabs of a CU's two workgroups can be combined before they leave it.
static bool wgrad9_four_waves(int W) {
    static const bool on = getenv("VC_WGRAD9_4WAVE") != nullptr;
    const int lds = wgrad9_lds<4>(W);
    return on && lds != 0 && 2 * lds <= 160 * 1024;
}

int vc_wgrad9_plan(long M, int W, int Cs, int Co, int ndt, int* rows_per_split) {
    if (wgrad9_lds<8>(W) == 0 && !wgrad9_four_waves(W)) return 0;
    const int CH = 128;  // slices are whole chunks of either form
    const long tiles = (long)ndt * (Cs / 64) * (Co / 64);
    static const long target_env = getenv("VC_WGRAD9_TARGET") ? atol(getenv("VC_WGRAD9_TARGET")) : 0;
    // 192 workgroups of eight waves, not one per CU: the weight gradients run on the side stream, and the quarter of
    // the CUs they leave free takes the main stream's kernels (measured over 96 .. 512: R3D-18 step 4.95 ms at 256,
    // 4.90 at 192, 4.98 at 128; R3D-50 10.45 / 10.39 / 10.50) -- and a slice fewer means a slab fewer to reduce
    const long target = target_env > 0 ? target_env : (wgrad9_four_waves(W) ? 512 : 192);
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

template <int NW>
static int launch_wgrad9(const VcWgrad9Params& p, hipStream_t stream) {
    static VcPerDeviceOnce once;  // thread-safe, once per device
    once.run([&]() { (void)hipFuncSetAttribute((const void*)wgrad9_kernel<NW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
    dim3 grid(p.ndt * (p.Cs / 64), p.Co / 64, p.splits);
    (void)hipGetLastError();
    hipLaunchKernelGGL(wgrad9_kernel<NW>, grid, dim3(NW * 64), wgrad9_lds<NW>(p.W), stream, p);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        vc_set_error("wgrad9 launch failed: %s", hipGetErrorString(e));
        return -1;
    }
    return 0;
}

int vc_launch_wgrad9(VcWgrad9Params& p, hipStream_t stream) {
    static const bool no_xcd_map = getenv("VC_NO_XCD_MAP") != nullptr;
    p.xcd_map = no_xcd_map ? 0 : 1;
    p.dbg = g_w9_dbg;
    const bool four = wgrad9_four_waves(p.W) && p.rows_per_split % 64 == 0;
    if ((!four && wgrad9_lds<8>(p.W) == 0) || p.Cs % 64 != 0 || p.Co % 64 != 0) {
        vc_set_error("wgrad9: unsupported geometry (W=%d Cs=%d Co=%d)", p.W, p.Cs, p.Co);
        return -2;
    }
    const int HW = p.H * p.W;
    p.dHW = vc_make_fastdiv(HW), p.dT = vc_make_fastdiv(p.T), p.dW = vc_make_fastdiv(p.W);
    if (p.ndt < 1 || p.ndt > 3) p.dt0 = 0, p.ndt = 3;
    return four ? launch_wgrad9<4>(p, stream) : launch_wgrad9<8>(p, stream);
}
