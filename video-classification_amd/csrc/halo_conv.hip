// Stride-1 Conv3d forward / data-gradient with the activation HALO kept in LDS (gfx950).
//
// The tap-gather implicit GEMM of igemm_conv.hip re-fetches every activation row once per tap
// (27x for 3x3x3): with Co = 64 per tile that is 64-80 bytes of global->LDS traffic per MFMA
// clock per CU against a measured fill rate of ~18, so it tops out near 22 % of the MFMA peak.
// Here a workgroup loads the activation rows its tile needs ONCE per 64-channel chunk -- TT
// consecutive (n,t) planes x SEG consecutive (h,w) rows, plus one plane and W+1 rows of halo on
// each side -- and walks the taps over that LDS image: a tap is a constant row shift
// (dt*HR + dh*W + dw), applied when the B fragments are read.  Rows of the image that a tap must
// not see (outside the sample: zero padding) are redirected per lane to a 128-byte zero row, from a
// 27-bit validity mask computed once per output position.  Only the 8 KiB weight tiles stream in
// during the tap loop: LDS-DMA into a 3-deep ring, two taps ahead of the MFMAs, retired with a counted
// s_waitcnt vmcnt(N) and a raw s_barrier (a __syncthreads would drain the DMA queue every tap; an
// LDS-DMA takes about 1 us to land under load, twice the MFMA time of a tap).
//
// GEMM view per tap:  D[co][m] += W_tap[co][ci] * X[m + shift(tap)][ci]
//   tile   : 64 co x (NW*64) position slots; wave w owns slots [64w, 64w+64): 4x4 accumulators of
//            v_mfma_f32_16x16x32_bf16, as in igemm_conv.hip (weights = A operand, so a lane ends
//            with 4 consecutive channels of one position: an 8-byte channels-last store)
//   slot p : plane tp = p / SEG, row s = p % SEG of the tile; slots >= TT*SEG are idle
//   LDS    : [W buf 0 | W buf 1 | zero row | halo image], 128-byte rows, 16-byte slots XOR-swizzled
//            by HaloCfg::swz(row) on the DMA source side so that ds_read_b128 is conflict free
// Dgrad of a stride-1 conv is the same kernel with the mirrored tap table and [Ci][taps*Co] weights.
#include <stdio.h>
#include <stdlib.h>

#include "common.h"
#include "kernels.h"

namespace {

constexpr int NWB = 3;  // weight tiles in the ring: the DMA runs 2 taps ahead of the MFMAs

// CK = channels per K chunk (64 or 32).  CK = 32 halves the image and the weight tiles: two 4-wave
// workgroups then fit in one CU's LDS and one loads / stores while the other runs its MFMAs.
template <int CK>
struct HaloCfg {
    static constexpr int ROWB = CK * 2;         // bytes per LDS row
    static constexpr int PR = 1024 / ROWB;      // rows per 1-KiB LDS-DMA piece (8 or 16)
    static constexpr int SLOTS = ROWB / 16;     // 16-byte slots per row (8 or 4)
    static constexpr int KS = CK / 32;          // MFMA k-steps per tap and chunk
    static constexpr int WBUF = 64 * ROWB;      // one weight tile: 64 co x CK ci
    static constexpr int WPIECES = WBUF / 1024; // 8 or 4
    static constexpr int ZERO_OFF = NWB * WBUF; // 256 bytes reserved, the first ROWB are the zero row
    static constexpr int HALO_OFF = NWB * WBUF + 256;
    // XOR swizzle of the 16-byte slot by the row, so that the 16 lanes ds_read_b128 serves in one LDS cycle fall into
    // 16 different 16-byte pieces of the 256-byte bank row.  Those 16 lanes are NOT 16 consecutive ones: the groups are
    // {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} (+32 for the upper half; MI355X_MICROARCH.md, LDS table), i.e. with
    // lane = lg*16 + l15 and row = r0 + l15: rows r0 + {0-3, 12-15} with slot lg and rows r0 + {4-11} with slot lg ^ 1.
    // 128-byte rows (two per bank row): slot ^= 2*bit1(row) ^ 4*bit2(row); 64-byte rows (four per bank row):
    // slot ^= 2*bit2(row) -- conflict-free for every r0 (exhaustive host check).  Rounds 1-2 used (row >> 1) & 7 and
    // (row >> 2) & 3, derived for 16 consecutive lanes: 1.75 / 1.88 LDS cycles per group (SQ_LDS_BANK_CONFLICT 27 / 49 %).
    static __device__ __host__ constexpr int swz(int row) { return CK == 64 ? ((row >> 1) & 3) << 1 : ((row >> 2) & 1) << 1; }
};

__device__ __attribute__((aligned(128))) unsigned int vc_halo_zero_page[32];

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// sum over the 16 lanes of a DPP row (lanes 16g .. 16g+15), result in every lane: full-rate VALU, no LDS
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));  // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));  // row_mirror
    return v;
}

#ifndef VC_HALO_ABLATE
#define VC_HALO_ABLATE 0  // timing experiments only (scripts/halo_ablate.sh): 1 no tap barrier, 2 no image reads, 4 no weight reads, 8 no weight DMA, 16 no MFMA, 64 / 128 image fill from the zero page with / without its address arithmetic (padded form)
#endif

// PAD: the LDS image is stored zero-padded -- (R + 2) x (W + 1) + 1 rows per plane, out-of-sample rows filled
// from the zero page by the loader -- so a tap is a pure row shift: no validity masks, no per-tap selects
// (the tap loop is VALU-issue bound otherwise).  Needs whole-h-row segments and TT | T.
// reads / MFMAs of one pipeline stage interleaved in program order: R fragment reads spread evenly over M MFMAs
template <int I, int R, int M>
struct HaloSched {
    static __device__ __forceinline__ void run() {
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        constexpr int n = ((I + 1) * M) / R - (I * M) / R;
        if constexpr (n > 0) __builtin_amdgcn_sched_group_barrier(0x008, n, 0);
        if constexpr (I + 1 < R) HaloSched<I + 1, R, M>::run();
    }
};

// NB = 16-position blocks per wave: the wave tile is 64 co x NB*16 positions (NB*4 accumulator quads).  NB = 4 with
// eight waves is the one-workgroup-per-CU form; NB = 7 / 8 with four waves and 32-channel chunks (64-byte image rows)
// halves the LDS of a workgroup, so TWO workgroups share a CU: one's image fill, barriers and epilogue run beside
// the other's MFMAs, and a wave reads 4 weight + NB activation fragments per 4*NB MFMAs instead of 4 + 4 per 16.
template <int NW, int CK, bool PAD, int NB>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(2, 2))) void halo_conv_kernel(const VcHaloParams p) {
    using Cfg = HaloCfg<CK>;
    constexpr int WS = NB * 16;  // position slots per wave
    constexpr int RSH = CK == 64 ? 7 : 6;  // log2 of the image row pitch in bytes
    constexpr int ROWB = Cfg::ROWB, WBUF = Cfg::WBUF, ZERO_OFF = Cfg::ZERO_OFF, HALO_OFF = Cfg::HALO_OFF, PR = Cfg::PR,
                  KS = Cfg::KS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15;
    const int lg = lane >> 4;

    // ---- tile coordinates
    // Workgroups are dealt round-robin to the 8 XCDs (workgroup id mod 8), each with its own L2.  Neighbouring tiles
    // share their halos (two of a tile's TT + 2 planes with each temporal neighbour, two of its R + 2 rows with each
    // segment neighbour: the image is 2-3x the tile), so XCD k takes the k-th contiguous eighth of the tiles and the
    // shared rows are L2 hits there instead of fabric traffic into eight L2s.
    unsigned bx = blockIdx.x;
    if (p.xcd_map) bx = (bx & 7u) * (gridDim.x >> 3) + (bx >> 3);
    const int tg = (int)bx / p.nseg;
    const int sg = (int)bx - tg * p.nseg;
    const int P0 = tg * p.TT;          // first (n,t) plane of the tile
    const int s0 = sg * p.SEG;         // first (h,w) row of the tile inside a plane
    const int co0 = blockIdx.y * 64;
    const int HW = p.H * p.W;

    if (tid < 32) ((unsigned*)(smem + ZERO_OFF))[tid] = 0u;
    // the tap table lives in two VGPRs (lane i = tap i) and is read with v_readlane: no scalar-memory
    // round trip (and no lgkmcnt(0) drain) per tap inside the loops
    const int tv_shift = p.taps[lane & 31].src_off;
    const int tv_wk = p.taps[lane & 31].wk_off;
    const int tv_off = ((int)p.taps[lane & 31].ot & 0xff) | (((int)p.taps[lane & 31].oh & 0xff) << 8) |
                       (((int)p.taps[lane & 31].ow & 0xff) << 16);
#define VC_STAMP(k)                                                                                          \
    do {                                                                                                     \
        if (p.dbg != nullptr && tid == 0 && blockIdx.y == 0 && blockIdx.z == 0)                              \
            p.dbg[(size_t)blockIdx.x * 8 + (k)] = (long long)__builtin_amdgcn_s_memtime();                   \
    } while (0)
    VC_STAMP(0);

    // ---- per-lane output slots: NB positions (b = 0..NB-1), row = wave*WS + b*16 + l15
    int lrow0[NB];        // LDS image row of the slot's own position (tap 0,0,0); PAD: its byte offset (row << RSH)
    unsigned mask[NB];    // bit i: tap i reads inside the sample (unused with PAD)
    int ooff[NB];         // element offset of the output row, -1 = idle slot
    if constexpr (PAD) {
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int slot = wave * WS + b * 16 + l15;
            const int tp = (int)vc_div((unsigned)slot, p.dSEG);
            const int s = slot - tp * p.SEG;
            const int hl = (int)vc_div((unsigned)s, p.dW);
            const int w = s - hl * p.W;
            const int P = P0 + tp;
            const bool ok = tp < p.TT && P < p.NP;
            // idle slot: the tile's first position (every tap stays inside the image; the result is dropped)
            lrow0[b] = (ok ? (tp + 1) * p.HR + (hl + 1) * p.Wp + (w + 1) : p.HR + p.Wp + 1) << RSH;
            ooff[b] = ok ? (P * HW + s0 + s) * p.Cn : -1;
            mask[b] = 0;
        }
    } else {
    int pt_[NB], ph_[NB], pw_[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int slot = wave * WS + b * 16 + l15;
        const int tp = (int)vc_div((unsigned)slot, p.dSEG);
        const int s = slot - tp * p.SEG;
        const int P = P0 + tp;
        const int hw = s0 + s;
        const bool ok = tp < p.TT && P < p.NP && hw < HW;
        lrow0[b] = ok ? (tp + 1) * p.HR + s + p.W + 1 : p.HR + p.W + 1;  // idle slot: any in-image row, all taps masked
        const int n = (int)vc_div((unsigned)P, p.dT);
        const int h = (int)vc_div((unsigned)hw, p.dW);
        pt_[b] = ok ? P - n * p.T : -4;  // -4: every tap falls outside
        ph_[b] = h;
        pw_[b] = hw - h * p.W;
        ooff[b] = ok ? (P * HW + hw) * p.Cn : -1;
        mask[b] = 0;
    }
    if (p.canon != 0) {
        // canonical 3x3x3 table (tap i = (a*3 + b)*3 + c with offsets a-1, b-1, c-1, or all negated):
        // the 27-bit mask is the outer product of three 3-bit per-axis validity masks
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            unsigned vt = 0, vh = 0, vw = 0;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int o = p.canon > 0 ? k - 1 : 1 - k;
                vt |= ((unsigned)(pt_[b] + o) < (unsigned)p.T) ? (1u << k) : 0u;
                vh |= ((unsigned)(ph_[b] + o) < (unsigned)p.H) ? (1u << k) : 0u;
                vw |= ((unsigned)(pw_[b] + o) < (unsigned)p.W) ? (1u << k) : 0u;
            }
            const unsigned hw9 = ((vh & 1u) ? vw : 0u) | ((vh & 2u) ? vw << 3 : 0u) | ((vh & 4u) ? vw << 6 : 0u);
            mask[b] = ((vt & 1u) ? hw9 : 0u) | ((vt & 2u) ? hw9 << 9 : 0u) | ((vt & 4u) ? hw9 << 18 : 0u);
        }
    } else {
        for (int i = 0; i < p.ntaps; ++i) {
            const int o3 = __builtin_amdgcn_readlane(tv_off, i);
            const int ot = (int)(signed char)(o3 & 0xff), oh = (int)(signed char)((o3 >> 8) & 0xff),
                      ow = (int)(signed char)((o3 >> 16) & 0xff);
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const bool in = (unsigned)(pt_[b] + ot) < (unsigned)p.T && (unsigned)(ph_[b] + oh) < (unsigned)p.H &&
                                (unsigned)(pw_[b] + ow) < (unsigned)p.W;
                mask[b] |= in ? (1u << i) : 0u;
            }
        }
    }
    }

    const int nchunks = p.Cs / CK;
    int cbeg = 0, cend = nchunks;
    if (p.splits > 1) {
        cbeg = blockIdx.z * p.cps;
        cend = min(nchunks, cbeg + p.cps);
    }

    // ---- loader roles
    // weight tile: 8 pieces of 8 rows x 128 B; wave w issues pieces w, w+NW, ...
    const int prow = lane / Cfg::SLOTS;   // row inside a piece
    const int pslot = lane % Cfg::SLOTS;  // physical 16-B slot written by this lane
    const gptr_t zero = (gptr_t)vc_halo_zero_page;
    const long Mtot = (long)p.NP * HW;
    const long gplane0 = (long)(P0 - 1) * HW + s0 - (p.W + 1);  // global row of image row 0

    auto issue_w = [&](int tap, int cch, int buf) {
        char* dst = smem + buf * WBUF;
        for (int pc = wave; pc < Cfg::WPIECES; pc += NW) {
            const int r = pc * PR + prow;
            const int ls = pslot ^ Cfg::swz(r);
            const bf16* src = p.wgt + ((size_t)(co0 + r) * p.Kw + __builtin_amdgcn_readlane(tv_wk, tap) + cch * CK + ls * 8);
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(dst + pc * 1024), 16, 0, 0);
        }
    };
    auto issue_halo = [&](int cch) {
        char* dst = smem + HALO_OFF;
        if constexpr (PAD) {
            // image row r = q*HR + hh*(W+1) + ww, HR = (R+2)*(W+1) + 1  ->  plane P0-1+q, row h0-1+hh, column ww-1; zero
            // outside.  ONE zero column per h-row: column 0 is the left neighbour of w = 0 and, being the element after
            // the previous row's w = W-1, that row's right neighbour too; the plane's last row gets one trailing element.
            //
            // Addressing: a row's source is a 32-bit offset from a tile base held in SGPRs, its validity four range tests
            // against tile-uniform bounds, and the rows that stay zero are not loaded from a zero page: their lanes are
            // masked off the DMA and the rows are zeroed by plain LDS stores, once per tile (no chunk ever writes them).
            // (What bounds the fill is the CU's LDS-DMA rate, ~14 B/clk with all of its waves issuing: cheaper address
            // arithmetic, an XCD-aware tile order and a VGPR-staged fill -- 60 % slower -- all left its 9-11k clocks
            // per 67-134 KB where they were; scripts/halo_ablate.sh, modes 64 / 128.)
            const int p0t = P0 - (int)vc_div((unsigned)P0, p.dT) * p.T;
            const int h0 = sg * p.R;
            const int q_lo = p0t == 0 ? 1 : 0;  // planes of the image that exist: [q_lo, q_hi)
            const int q_hi = min(p.TT + 2 - (p0t + p.TT == p.T ? 1 : 0), p.NP - P0 + 1);
            const int h_lo = h0 > 0 ? 0 : 1;    // h-rows: [h_lo, h_hi)
            const int h_hi = min(p.R + 2, p.H - h0 + 1);
            const long long base_e = ((long long)(P0 - 1) * HW + (long long)(h0 - 1) * p.W - 1) * p.Cs + cch * CK;
            const char* const tbase = (const char*)p.src + base_e * 2;  // uniform; may lie before the tensor (never read there)
            // A wave takes a contiguous run of pieces, so (plane q, h-row hh, column wc) of its row advance by PR rows per
            // piece with at most one wrap each (PR <= Wp; narrower images divide per piece): no division, and the one
            // multiply left (row offset x Cs) is a 24-bit one.
            const int ppw = (p.npieces + NW - 1) / NW;
            int pc = wave * ppw;
            const int pc_end = min(pc + ppw, p.npieces);
            int r = pc * PR + prow;
            int q = (int)vc_div((unsigned)r, p.dHR);
            int u = r - q * p.HR;
            int hh = (int)vc_div((unsigned)u, p.dWp);
            int wc = u - hh * p.Wp;  // image column: 0 is the zero column, 1 + w the tensor's column w
            int qHW = q * HW;
            int rowoff = qHW + hh * p.W + wc;  // global row of (q, hh, wc) relative to the tile base
            const bool stepwise = PR <= p.Wp;
            const unsigned nq = (unsigned)(q_hi - q_lo), nh = (unsigned)(h_hi - h_lo);
            auto step = [&](bool& ok, unsigned& rel) {  // this piece's row: loadable?, byte offset from tbase; then advance
                ok = r < p.halo_rows && (unsigned)(q - q_lo) < nq && (unsigned)(hh - h_lo) < nh && wc >= 1;
                const int ls = pslot ^ Cfg::swz(r);
                rel = (__umul24((unsigned)rowoff, (unsigned)p.Cs) + (unsigned)(ls * 8)) * 2u;
                r += PR;
                if (stepwise) {
                    u += PR, wc += PR, rowoff += PR;
                    const bool hwrap = wc >= p.Wp;  // next h-row: the zero column is not in the tensor, one row less
                    wc = hwrap ? wc - p.Wp : wc;
                    hh = hwrap ? hh + 1 : hh;
                    rowoff = hwrap ? rowoff - 1 : rowoff;
                    const bool pwrap = u >= p.HR;   // next plane: u < PR <= Wp, so h-row 0, column u
                    u = pwrap ? u - p.HR : u;
                    qHW = pwrap ? qHW + HW : qHW;
                    q = pwrap ? q + 1 : q;
                    hh = pwrap ? 0 : hh;
                    wc = pwrap ? u : wc;
                    rowoff = pwrap ? qHW + u : rowoff;
                } else {
                    q = (int)vc_div((unsigned)r, p.dHR);
                    u = r - q * p.HR;
                    hh = (int)vc_div((unsigned)u, p.dWp);
                    wc = u - hh * p.Wp;
                    rowoff = q * HW + hh * p.W + wc;
                }
            };
            for (; pc < pc_end; ++pc) {
                bool ok;
                unsigned rel;
                step(ok, rel);
                if (VC_HALO_ABLATE & 128) {
                    __builtin_amdgcn_global_load_lds(zero, (lptr_t)(dst + pc * 1024), 16, 0, 0);
                } else if (ok) {
                    const gptr_t src = (VC_HALO_ABLATE & 64) ? zero : (gptr_t)(tbase + rel);
                    __builtin_amdgcn_global_load_lds(src, (lptr_t)(dst + pc * 1024), 16, 0, 0);
                } else if (cch == cbeg) {
                    *(u32x4*)(dst + pc * 1024 + lane * 16) = (u32x4){0u, 0u, 0u, 0u};
                }
            }
        } else {
            // image row r = q*HR + u  ->  global row gplane0 + q*HW + u
            int r = wave * PR + prow;
            int q = r / p.HR;
            int u = r - q * p.HR;
            for (int pc = wave; pc < p.npieces; pc += NW) {
                const long grow = gplane0 + (long)q * HW + u;
                const int ls = pslot ^ Cfg::swz(r);
                const bool ok = grow >= 0 && grow < Mtot && r < p.halo_rows;
                const gptr_t src = ok ? (gptr_t)(p.src + (grow * p.Cs + cch * CK + ls * 8)) : zero;
                __builtin_amdgcn_global_load_lds(src, (lptr_t)(dst + pc * 1024), 16, 0, 0);
                r += NW * PR;
                u += NW * PR;
                while (u >= p.HR) {
                    u -= p.HR;
                    ++q;
                }
            }
        }
    };

    f32x4 acc[4][NB];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    VC_STAMP(1);

    const int aswz = Cfg::swz(l15);  // == swz(a*16 + l15)
    const unsigned halo_base = HALO_OFF;

    constexpr int WPT = (Cfg::WPIECES + NW - 1) / NW;  // weight DMA instructions per wave per tap
    static_assert(Cfg::WPIECES % NW == 0 || Cfg::WPIECES < NW, "every wave must issue the same number of weight pieces");

    // fragments of one tap: KS k-steps x (4 weight + NB activation) ds_read_b128
    auto load_frags = [&](int tap, bf16x8 (&fa)[KS][4], bf16x8 (&fb)[KS][NB]) {
        const char* sW = smem + (tap % NWB) * WBUF;
        const int shift = __builtin_amdgcn_readlane(tv_shift, tap);  // row shift of this tap inside the image
        unsigned baddr[KS][NB];
        if constexpr (PAD) {
            // byte offset of the row L = (row << RSH); its 16-byte slot is lg ^ swz(row) (+4 for k-step 1):
            // 128-byte rows: swz = ((row >> 1) & 3) << 1 = ((L >> 8) & 3) << 1;  64-byte rows: ((L >> 8) & 1) << 1
            const unsigned sh = (unsigned)shift << RSH;
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const unsigned L = (unsigned)lrow0[b] + sh;
                const unsigned sw = (L >> 3) & (CK == 64 ? 0x60u : 0x20u);  // swz(row) << 4, from the row's byte offset
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) baddr[ks][b] = halo_base + ((sw ^ (unsigned)((lg << 4) ^ (ks << 6))) + L);
            }
        } else {
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const int lrow = lrow0[b] + shift;
                const unsigned a_in = halo_base + (unsigned)lrow * ROWB + ((unsigned)(lg ^ Cfg::swz(lrow)) << 4);
                const unsigned a_zero = ZERO_OFF + ((unsigned)lg << 4);
                const unsigned a0 = ((mask[b] >> tap) & 1u) ? a_in : a_zero;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) baddr[ks][b] = a0 ^ (unsigned)(ks << 6);
            }
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (!(VC_HALO_ABLATE & 4)) {
#pragma unroll
                for (int a = 0; a < 4; ++a)
                    fa[ks][a] = *(const bf16x8*)(sW + (a * 16 + l15) * ROWB + ((((ks << 2) + lg) ^ aswz) << 4));
            }
            if (!(VC_HALO_ABLATE & 2)) {
#pragma unroll
                for (int b = 0; b < NB; ++b) fb[ks][b] = *(const bf16x8*)(smem + baddr[ks][b]);
            }
        }
    };
    auto mfma_all = [&](const bf16x8 (&fa)[KS][4], const bf16x8 (&fb)[KS][NB]) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    if (VC_HALO_ABLATE & 16) {
                        asm volatile("" : "+v"(acc[a][b]) : "v"(fa[ks][a]), "v"(fb[ks][b]));
                    } else {
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[ks][a], fb[ks][b], acc[a][b], 0, 0, 0);
                    }
                }
    };
    // W[t] has landed (for this wave) once at most the instructions of the younger tiles are in flight;
    // `younger` = how many of W[t+1], W[t+2] have been issued.  All LDS reads of this wave are retired too,
    // so after the barrier the buffer they came from may be overwritten.
    // (__builtin_amdgcn_s_waitcnt rather than inline asm: the compiler's own wait-count pass sees the drain and
    // does not add a second lgkmcnt(0) in front of the first MFMA of the stage.  imm = vmcnt[3:0] | expcnt 7 << 4
    // | lgkmcnt << 8 | vmcnt[5:4] << 14; 0xF in a field = no wait.)
    auto wait_w = [&](int younger) {
#if VC_HALO_ABLATE & 32
        if (younger >= 2) {
            if (WPT == 1) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        } else if (younger == 1) {
            if (WPT == 1) asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        return;
#endif
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0), unconditionally (the vmcnt part below may sit in branches)
        if (younger >= 2) {
            if (WPT == 1) __builtin_amdgcn_s_waitcnt(0x0F72);  // vmcnt(2)
            else __builtin_amdgcn_s_waitcnt(0x0F74);           // vmcnt(4)
        } else if (younger == 1) {
            if (WPT == 1) __builtin_amdgcn_s_waitcnt(0x0F71);  // vmcnt(1)
            else __builtin_amdgcn_s_waitcnt(0x0F72);           // vmcnt(2)
        } else {
            __builtin_amdgcn_s_waitcnt(0x0F70);                // vmcnt(0)
        }
        if (!(VC_HALO_ABLATE & 1)) __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    // one pipeline stage (t + 1 < ntaps): make W[t+1] visible, refill the ring, then fetch the fragments of
    // tap t+1 into `nfa/nfb` WHILE the MFMAs of tap t run from `cfa/cfb`.  The reads are interleaved with
    // the MFMAs in program order (2 ds_read_b128 per 4 MFMAs): issued all up front, the 16 reads would
    // overflow the 4-bit lgkmcnt and the compiler would have to drain them before the first MFMA.
    auto stage = [&](int t, int cch, bf16x8 (&cfa)[KS][4], bf16x8 (&cfb)[KS][NB], bf16x8 (&nfa)[KS][4],
                     bf16x8 (&nfb)[KS][NB]) {
        wait_w(t + 2 < p.ntaps ? 1 : 0);  // issued so far: up to W[t+2]
        if (!(VC_HALO_ABLATE & 8) && t + 3 < p.ntaps) issue_w(t + 3, cch, (t + 3) % NWB);  // overwrites W[t]: its reads are retired
        load_frags(t + 1, nfa, nfb);
        mfma_all(cfa, cfb);
        if constexpr (NB == 4) {
#pragma unroll
            for (int i = 0; i < 4 * KS; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);  // 2 DS reads
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);  // 4 MFMAs
            }
        } else {
            HaloSched<0, (4 + NB) * KS, 4 * NB * KS>::run();
        }
    };

    // NB > 4 (two workgroups per CU, KS == 1): 112-128 accumulator registers leave no room for two full fragment
    // sets.  The MFMAs run position block by position block (b outer, the four co blocks inner), so the activation
    // fragment of block b is dead after its four MFMAs and the same registers take tap t+1's fragment of block b
    // right there: ONE activation set refreshed in place, only the four weight fragments are double-buffered.
    auto stage2 = [&](int t, int cch, bf16x8 (&cfa)[KS][4], bf16x8 (&nfa)[KS][4], bf16x8 (&fb)[KS][NB]) {
        wait_w(t + 2 < p.ntaps ? 1 : 0);
        if (!(VC_HALO_ABLATE & 8) && t + 3 < p.ntaps) issue_w(t + 3, cch, (t + 3) % NWB);
        const char* sW = smem + ((t + 1) % NWB) * WBUF;
        const int shift = __builtin_amdgcn_readlane(tv_shift, t + 1);
        unsigned baddr[NB];
        if constexpr (PAD) {
            const unsigned sh = (unsigned)shift << RSH;
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const unsigned L = (unsigned)lrow0[b] + sh;
                const unsigned sw = (L >> 3) & (CK == 64 ? 0x60u : 0x20u);  // swz(row) << 4, from the row's byte offset
                baddr[b] = halo_base + ((sw ^ (unsigned)(lg << 4)) + L);
            }
        } else {
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const int lrow = lrow0[b] + shift;
                const unsigned a_in = halo_base + (unsigned)lrow * ROWB + ((unsigned)(lg ^ Cfg::swz(lrow)) << 4);
                const unsigned a_zero = ZERO_OFF + ((unsigned)lg << 4);
                baddr[b] = ((mask[b] >> (t + 1)) & 1u) ? a_in : a_zero;
            }
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) nfa[0][a] = *(const bf16x8*)(sW + (a * 16 + l15) * ROWB + ((lg ^ aswz) << 4));
#pragma unroll
        for (int b = 0; b < NB; ++b) {
#pragma unroll
            for (int a = 0; a < 4; ++a) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cfa[0][a], fb[0][b], acc[a][b], 0, 0, 0);
            fb[0][b] = *(const bf16x8*)(smem + baddr[b]);
        }
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);  // the four weight fragments of tap t+1
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);  // block b of tap t
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // its registers take block b of tap t+1
        }
    };

    for (int cch = cbeg; cch < cend; ++cch) {
        if (cch != cbeg) __syncthreads();  // every wave is done with the previous chunk's image and weights
        issue_halo(cch);
        issue_w(0, cch, 0);
        if (p.ntaps > 1) issue_w(1, cch, 1);
        if (p.ntaps > 2) issue_w(2, cch, 2);
        if (cch == cbeg) VC_STAMP(2);
        wait_w(p.ntaps > 2 ? 2 : (p.ntaps > 1 ? 1 : 0));
        VC_STAMP(cch == cbeg ? 3 : 5);
        if constexpr (NB == 4) {
            bf16x8 fa0[KS][4], fb0[KS][NB], fa1[KS][4], fb1[KS][NB];
            load_frags(0, fa0, fb0);
            int tap = 0;
            for (; tap + 2 < p.ntaps; tap += 2) {
                stage(tap, cch, fa0, fb0, fa1, fb1);
                stage(tap + 1, cch, fa1, fb1, fa0, fb0);
            }
            mfma_all(fa0, fb0);  // (odd tap count, see below)
        } else {
            static_assert(NB == 4 || KS == 1, "the in-place fragment refresh is written for one k-step per tap");
            bf16x8 fa0[KS][4], fa1[KS][4], fb[KS][NB];
            load_frags(0, fa0, fb);
            int tap = 0;
            for (; tap + 2 < p.ntaps; tap += 2) {
                stage2(tap, cch, fa0, fa1, fb);
                stage2(tap + 1, cch, fa1, fa0, fb);
            }
            // ntaps is odd (checked by the host; every axis of a "same"-padded 1- or 3-wide kernel has 1 or 3 live
            // taps), so the last tap's fragments are in fa0.  With the even-count tail as a second path, the
            // accumulators met in phi nodes at its merge that the register allocator did not coalesce: this form
            // spilled ~100 VGPRs, stores and reloads between the MFMAs, and the 64 x 64 forms sat at 250-256.
            mfma_all(fa0, fb);
        }
        VC_STAMP(cch == cbeg ? 4 : 6);
    }

    // ---- epilogue: acc[a][b][j] = D[co = co0 + a*16 + lg*4 + j][slot = wave*WS + b*16 + l15]
    const int cobase = co0 + lg * 4;
    if (p.partial != nullptr) {
        float* slab = p.partial + (size_t)blockIdx.z * p.partial_stride;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            if (ooff[b] < 0) continue;
#pragma unroll
            for (int a = 0; a < 4; ++a) *(f32x4*)(slab + (size_t)ooff[b] + cobase + a * 16) = acc[a][b];
        }
        return;
    }
    if (p.ep_params != nullptr) {
        // inference: BatchNorm with fixed (running) statistics is a per-channel scale and shift, applied with the
        // ReLU on the fp32 accumulators, so the conv output is rounded to bf16 once and no separate pass reads it back
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const f32x4 s4 = *(const f32x4*)(p.ep_params + 2 * p.Cn + cobase + a * 16);
            const f32x4 h4 = *(const f32x4*)(p.ep_params + 3 * p.Cn + cobase + a * 16);
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float v = acc[a][b][j] * s4[j] + h4[j];
                    if (p.ep_relu && p.ep_res == nullptr) v = v > 0.f ? v : 0.f;  // with a residual: after the add
                    acc[a][b][j] = v;
                }
        }
    }
    if (p.accumulate) {
        // dx += : the old values are added to the fp32 accumulators (one rounding), each wave on its own before the
        // barrier -- the loads' latency falls into the time the other waves still spend in their tap loops
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            if (ooff[b] < 0) continue;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const bf16x4 old = *(const bf16x4*)(p.out + (size_t)ooff[b] + cobase + a * 16);
                acc[a][b][0] += bf2f(old[0]), acc[a][b][1] += bf2f(old[1]);
                acc[a][b][2] += bf2f(old[2]), acc[a][b][3] += bf2f(old[3]);
            }
        }
    }
    __syncthreads();  // every wave has left the tap loop: the image area is free
    // The image area is reused: [NW][WS positions][64 co] bf16 with a 144-byte row pitch for the output
    // tile (so that it leaves as whole 128-byte channels-last rows, 16 bytes per lane), then
    // [NW][2][64] floats for the BatchNorm partial sums of the waves.
    constexpr int OPITCH = 144;
    char* otile = smem + HALO_OFF + wave * WS * OPITCH;
    float* sstat = (float*)(smem + HALO_OFF + NW * WS * OPITCH);
    int* sooff = (int*)(sstat + NW * 128);  // [NW][WS] output offsets of the wave's slots
    // Every path stages the tile in LDS and leaves as whole rows; sooff = the rows' output offsets.
#pragma unroll
    for (int b = 0; b < NB; ++b)
        if (lg == 0) sooff[wave * WS + b * 16 + l15] = ooff[b];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            bf16x4* q = (bf16x4*)(otile + (b * 16 + l15) * OPITCH + (a * 16 + lg * 4) * 2);
            const f32x4 v = acc[a][b];
            *q = (bf16x4){f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
        }
    }
    if (p.stats != nullptr) {
        // four channels at a time (packed fp32), without the per-element slot test when every slot of the wave is a real
        // position (wave-uniform): this accumulation is exposed -- it cost a layer1 forward 15 us of 134
        bool anybad = false;
#pragma unroll
        for (int b = 0; b < NB; ++b) anybad = anybad || ooff[b] < 0;
        const bool clean = __builtin_amdgcn_ballot_w64(anybad) == 0;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            f32x4 sh4 = {0.f, 0.f, 0.f, 0.f};  // shifted sums (bn_fin.h)
            if (p.stat_shift != nullptr) sh4 = *(const f32x4*)(p.stat_shift + cobase + a * 16);
            f32x4 ss4 = {0.f, 0.f, 0.f, 0.f}, qq4 = {0.f, 0.f, 0.f, 0.f};
            if (clean) {
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    const f32x4 v = acc[a][b] - sh4;
                    ss4 += v;
                    qq4 += v * v;
                }
            } else {
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    if (ooff[b] < 0) continue;
                    const f32x4 v = acc[a][b] - sh4;
                    ss4 += v;
                    qq4 += v * v;
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float ss = row16_sum(ss4[j]);
                const float qq = row16_sum(qq4[j]);
                if (l15 == 0) {
                    sstat[wave * 128 + a * 16 + lg * 4 + j] = ss;
                    sstat[wave * 128 + 64 + a * 16 + lg * 4 + j] = qq;
                }
            }
        }
    }
    __syncthreads();
    if (p.stats != nullptr && tid < 128) {  // one atomic per channel and moment per workgroup
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) s += sstat[w * 128 + tid];
        vc_stat_add(p.stats, blockIdx.x, p.Cn, (tid < 64 ? 0 : p.Cn) + co0 + (tid & 63), s);
    }
    {
        // each wave streams its own WS x 64 tile: 8 rows of 128 bytes per instruction
#pragma unroll
        for (int i = 0; i < NB * 2; ++i) {
            const int r = i * 8 + (lane >> 3);
            const int oo = sooff[wave * WS + r];
            if (oo >= 0) {
                bf16x8 v = *(const bf16x8*)(otile + r * OPITCH + (lane & 7) * 16);
                if (p.ep_res != nullptr) {
                    // residual add (+ ReLU) on the staged rows: whole 16-byte pieces of the residual, as the store
                    // (the staged value is already bf16: two roundings, as conv -> bf16 -> bn_apply has)
                    const bf16x8 rr = *(const bf16x8*)(p.ep_res + (size_t)oo + co0 + (lane & 7) * 8);
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        float f = bf2f(v[k]) + bf2f(rr[k]);
                        if (p.ep_relu) f = f > 0.f ? f : 0.f;
                        v[k] = f2bf(f);
                    }
                }
                *(bf16x8*)(p.out + (size_t)oo + co0 + (lane & 7) * 8) = v;
            }
        }
    }
    VC_STAMP(7);
    vc_bn_fin_last<4>(p.fin, p.stats, p.Cn);
}

// LDS bytes of a workgroup: ring + zero row + the larger of the image and the epilogue staging area
static int halo_lds(int NW, int CK, int npieces, int NB = 4) {
    const int off = CK == 32 ? HaloCfg<32>::HALO_OFF : HaloCfg<64>::HALO_OFF;
    const int img = off + npieces * 1024;
    const int epi = off + NW * NB * 16 * 144 + NW * 128 * 4 + NW * NB * 16 * 4;
    return img > epi ? img : epi;
}

template <int NW, int CK, bool PAD, int NB = 4>
int launch_halo(const VcHaloParams& p, hipStream_t stream) {
    const int lds = halo_lds(NW, CK, p.npieces, NB);
    static VcPerDeviceOnce once;  // thread-safe, once per device
    once.run([&]() { (void)hipFuncSetAttribute((const void*)halo_conv_kernel<NW, CK, PAD, NB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
    dim3 grid(p.ntg * p.nseg, p.Cn / 64, p.splits > 1 ? p.splits : 1);

    (void)hipGetLastError();
    hipLaunchKernelGGL((halo_conv_kernel<NW, CK, PAD, NB>), grid, dim3(NW * 64), lds, stream, p);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        vc_set_error("halo conv launch failed: %s", hipGetErrorString(e));
        return -1;
    }
    return 0;
}

}  // namespace

// Tile planning: NW waves (4 or 8), TT planes x SEG rows per tile, split-K over 64-channel chunks.
// Cost model in shader clocks per workgroup, constants measured on MI355X with per-phase s_memtime stamps
// (scripts/halo_stamps.py): ~2k prologue, ~11.5 clocks per image row of fill (the chip-wide fill rate when
// every CU loads its image at once), per tap and chunk ~1550 with two waves per SIMD (8 waves, or two
// 4-wave workgroups per CU) and ~2500 for a lone 4-wave workgroup, ~7k epilogue; workgroups run in rounds
// of (256 CUs x workgroups per CU); split-K adds the slab traffic and a finalize launch.
static const double vc_halo_k2 = getenv("VC_HALO_K2") ? atof(getenv("VC_HALO_K2")) : 1.0;  // tuning knob of the 2-WG/CU model

static const double vc_halo_kpad = getenv("VC_HALO_KPAD") ? atof(getenv("VC_HALO_KPAD")) : 1250.0;  // per-tap clocks, padded form

int vc_halo_plan(int NP, int T, int H, int W, int Cs, int Cn, int ntaps, VcHaloPlan* plan) {
    const int HW = H * W;
    if ((ntaps & 1) == 0) return -1;  // the kernel's tap pipeline is written for an odd count (what halo convs have)
    const char* env = getenv("VC_HALO_FORCE");  // "NW,TT,SEG,splits[,CK[,pad[,NB]]]" for experiments
    static const bool no_pad = getenv("VC_HALO_NO_PAD") != nullptr;
    // The two-workgroups-per-CU form (4 waves of 64 x 112, 32-channel chunks): one workgroup's image fill, barriers and
    // epilogue run beside the other's MFMAs.  Measured on MI355X, layer1 (64 x 8x28x28 x 64 -> 64): forward 0.131 ->
    // 0.110 ms, dgrad 0.126 -> 0.117; layer2 (128 ch, 4x14x14) stays with the eight-wave form (0.056 against 0.064).
    // VC_HALO_NO_2WG=1 takes it out of the planner.
    static const bool no_2wg = getenv("VC_HALO_NO_2WG") != nullptr;
    static const double k2wg = getenv("VC_HALO_K2WG") ? atof(getenv("VC_HALO_K2WG")) : 1.15;
    double best = 1e30;
    VcHaloPlan bp;
    bp.NW = 0;
    // forms: one workgroup per CU (8 or 4 waves of 64 x 64, 64-channel chunks), or two per CU (4 waves of 64 x NB*16,
    // 32-channel chunks: half the image bytes per chunk).  A 32-channel form with 64 x 64 wave tiles measured no
    // faster (doubled weight traffic per MFMA) and stays reachable through VC_HALO_FORCE only.
    struct Form { int NW, NB, CK; };
    static const Form forms[] = {{4, 4, 64}, {8, 4, 64}, {4, 7, 32}};
    for (const Form& f : forms) {
        if (f.CK == 32 && no_2wg) continue;
        const int CK = f.CK, NW = f.NW, NB = f.NB;
        const int nchunks = Cs / CK;
        const int PR = CK == 32 ? 16 : 8;
        for (int pad = 0; pad <= (no_pad ? 0 : 1); ++pad) {
            // the 32-channel forms exist in the padded variant only (the masked variants spilled 100+ registers)
            if (CK == 32 && !pad) continue;
            const int BM = NW * NB * 16;
            for (int SEG = 1; SEG <= (HW < BM ? HW : BM); ++SEG) {
                // segments that are whole (h) rows, whole planes, or an even split of a plane
                if (!(SEG == HW || SEG % W == 0 || HW % SEG == 0)) continue;
                if (pad && (SEG % W != 0 || HW % SEG != 0)) continue;  // padded form: whole h-rows, no ragged last segment
                // masked form: half-filled tiles as well -- on tiny planes (layer4: 64 planes of 4 x 4) more, smaller
                // workgroups fill the chip where full tiles leave half of it idle (VC_HALO_NO_SHRINK=1: full tiles only)
                static const bool no_shrink = getenv("VC_HALO_NO_SHRINK") != nullptr;
                for (int shrink = 1; shrink <= ((pad || no_shrink) ? 1 : 2); shrink *= 2) {
                    int TT = BM / SEG / shrink;
                    if (TT > NP) TT = NP;
                    if (pad) {
                        if (TT > T) TT = T;
                        while (TT > 1 && T % TT != 0) --TT;  // a tile must not straddle two samples
                    }
                    if (TT < 1) continue;
                    const int HR = pad ? (SEG / W + 2) * (W + 1) + 1 : SEG + 2 * W + 2;
                    const int rows = (TT + 2) * HR;
                    const int npieces = (rows + PR - 1) / PR;
                    const int lds = halo_lds(NW, CK, npieces, NB);
                    if (lds > 160 * 1024) continue;
                    if (CK == 32 && 2 * lds > 160 * 1024) continue;  // the 32-channel forms exist for two workgroups per CU
                    const int wgcu = (NW == 4 && 2 * lds <= 160 * 1024) ? 2 : 1;
                    const long tiles = (long)((NP + TT - 1) / TT) * ((HW + SEG - 1) / SEG);
                    for (int splits = 1; splits <= nchunks; splits *= 2) {
                        const int cps = (nchunks + splits - 1) / splits;
                        const long wgs = tiles * (Cn / 64) * splits;
                        const long rounds = (wgs + 256L * wgcu - 1) / (256L * wgcu);
                        double per_tap, fill, fixed;
                        if (CK == 32) {
                            // two workgroups share the CU's MFMA pipes: a tap step of one takes about twice its own MFMA
                            // time (NB*4 MFMAs of 16 clocks); the partner's MFMAs cover this one's fill and epilogue
                            per_tap = k2wg * 2.0 * NB * 4 * 16;
                            fill = 6.0 * rows;
                            fixed = 2000.0 + 4000.0;
                        } else {
                            per_tap = NW == 8 ? 1550.0 : (wgcu == 2 ? 1700.0 : 2500.0);
                            if (pad) per_tap *= vc_halo_kpad / 1550.0;
                            fill = 11.5 * rows;
                            fixed = 2000.0 + 7000.0;
                        }
                        const double t_wg = fixed + cps * (fill + ntaps * per_tap);
                        double cost = rounds * t_wg;
                        if (splits > 1) cost += 15000.0 + 2.4e3 * ((double)NP * HW * Cn * 4.0 * (splits + 0.5)) / 4.0e6;
                        if (cost < best) {
                            best = cost;
                            bp.NW = NW, bp.TT = TT, bp.SEG = SEG, bp.splits = splits, bp.CK = CK, bp.pad = pad, bp.NB = NB;
                        }
                    }
                }
            }
        }
    }
    if (env) {
        int a, b, c, d, e = 64, f = 0, g = 4;
        const int nf = sscanf(env, "%d,%d,%d,%d,%d,%d,%d", &a, &b, &c, &d, &e, &f, &g);
        if (nf >= 4 && (a == 4 || a == 8) && b >= 1 && c >= 1 && d >= 1 && (g == 4 || (a == 4 && e == 32 && g == 7)) &&
            b * c <= a * g * 16 && (e == 64 || (e == 32 && a == 4)) && d <= Cs / e &&
            (f == 0 || (c % W == 0 && HW % c == 0 && T % b == 0 && (e == 64 || g != 4))))
            bp.NW = a, bp.TT = b, bp.SEG = c, bp.splits = d, bp.CK = e, bp.pad = f, bp.NB = g;
    }
    if (bp.NW == 0) return -1;
    static const bool verbose = getenv("VC_HALO_VERBOSE") != nullptr;
    if (verbose)
        fprintf(stderr, "halo plan NP=%d T=%d H=%d W=%d Cs=%d Cn=%d taps=%d -> NW=%d NB=%d TT=%d SEG=%d splits=%d CK=%d pad=%d\n", NP, T,
                H, W, Cs, Cn, ntaps, bp.NW, bp.NB, bp.TT, bp.SEG, bp.splits, bp.CK, bp.pad);
    *plan = bp;
    return 0;
}

static long long* g_halo_dbg = nullptr;
extern "C" void vc_debug_halo_stamps(long long* buf) { g_halo_dbg = buf; }  // tuning aid: per-workgroup s_memtime stamps

int vc_launch_halo(VcHaloParams& p, const VcHaloPlan& plan, hipStream_t stream) {
    p.dbg = g_halo_dbg;
    p.stats = vc_det_tag(p.stats);
    if (p.Cs % 64 != 0 || p.Cn % 64 != 0 || p.ntaps < 1 || p.ntaps > 32 || (p.ntaps & 1) == 0) {
        vc_set_error("halo conv: bad channels/taps (Cs=%d Cn=%d taps=%d)", p.Cs, p.Cn, p.ntaps);
        return -2;
    }
    const int HW = p.H * p.W;
    p.TT = plan.TT, p.SEG = plan.SEG;
    p.pad = plan.pad;
    p.R = p.SEG / p.W, p.Wp = p.W + 1;
    if (p.pad && ((plan.CK != 64 && plan.NB == 4) || p.SEG % p.W != 0 || HW % p.SEG != 0 || p.T % p.TT != 0)) {
        vc_set_error("halo conv: padded form needs whole-row segments and TT | T (TT=%d SEG=%d)", p.TT, p.SEG);
        return -2;
    }
    p.HR = p.pad ? (p.R + 2) * p.Wp + 1 : p.SEG + 2 * p.W + 2;
    p.dHR = vc_make_fastdiv(p.HR), p.dWp = vc_make_fastdiv(p.Wp);
    p.halo_rows = (p.TT + 2) * p.HR;
    const int PR = plan.CK == 32 ? 16 : 8;
    p.npieces = (p.halo_rows + PR - 1) / PR;
    p.dSEG = vc_make_fastdiv(p.SEG), p.dT = vc_make_fastdiv(p.T), p.dW = vc_make_fastdiv(p.W);
    p.canon = 0;
    if (p.ntaps == 27) {
        bool fwd = true, rev = true;
        for (int i = 0; i < 27; ++i) {
            const int a = i / 9 - 1, b = (i / 3) % 3 - 1, c = i % 3 - 1;
            fwd = fwd && p.taps[i].ot == a && p.taps[i].oh == b && p.taps[i].ow == c;
            rev = rev && p.taps[i].ot == -a && p.taps[i].oh == -b && p.taps[i].ow == -c;
        }
        p.canon = fwd ? 1 : (rev ? -1 : 0);
    }
    p.ntg = (p.NP + p.TT - 1) / p.TT;
    p.nseg = (HW + p.SEG - 1) / p.SEG;
    static const bool no_xcd_map = getenv("VC_NO_XCD_MAP") != nullptr;
    p.xcd_map = ((p.ntg * p.nseg) % 8 == 0 && p.ntg * p.nseg >= 16 && !no_xcd_map) ? 1 : 0;
    if (p.TT * p.SEG > plan.NW * plan.NB * 16 || halo_lds(plan.NW, plan.CK, p.npieces, plan.NB) > 160 * 1024 ||
        (plan.CK != 32 && plan.CK != 64) || (plan.CK == 32 && plan.NW != 4) ||
        !(plan.NB == 4 || (plan.CK == 32 && plan.NB == 7))) {
        vc_set_error("halo conv: plan does not fit (TT=%d SEG=%d NW=%d)", p.TT, p.SEG, plan.NW);
        return -2;
    }
    for (int i = 0; i < p.ntaps; ++i) {  // a tap must stay inside the halo: one plane, one row, one column
        const VcTap& t = p.taps[i];
        if (t.ot < -1 || t.ot > 1 || t.oh < -1 || t.oh > 1 || t.ow < -1 || t.ow > 1) {
            vc_set_error("halo conv: tap offset outside the 1-deep halo");
            return -2;
        }
        p.taps[i].src_off = t.ot * p.HR + t.oh * (p.pad ? p.Wp : p.W) + t.ow;
    }
    if (plan.CK == 32) {
        if (plan.NB != 7 || !p.pad) {
            vc_set_error("halo conv: the 32-channel form exists as padded 64 x 112 wave tiles only");
            return -2;
        }
        return launch_halo<4, 32, true, 7>(p, stream);
    }
    if (p.pad) return plan.NW == 8 ? launch_halo<8, 64, true, 4>(p, stream) : launch_halo<4, 64, true, 4>(p, stream);
    return plan.NW == 8 ? launch_halo<8, 64, false, 4>(p, stream) : launch_halo<4, 64, false, 4>(p, stream);
}
