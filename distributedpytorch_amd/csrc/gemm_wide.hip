// Wide-tile bf16 GEMM for gfx950: PERSISTENT workgroups, 256 x 256 output tile, K step 64, 8 waves (512
// threads), one workgroup per CU with 128 KiB of LDS, operands by LDS-DMA (buffer_load ... lds) kept in flight
// ACROSS barriers and ACROSS output tiles behind counted s_waitcnt vmcnt(N).
//
//     y[m][n] = sum_k x[m][k] * w[n][k]          (both operands K-contiguous, fp32 accumulation)
//
// This is the R = S = 1, stride 1 case of ConvArgs (kernels.h): every Linear layer (forward and input
// gradient), every 1x1 convolution.  conv_igemm() routes a launch here when the grid fills the chip with
// 256-wide tiles (wide_gemm_eligible below); the 128 x 128 kernel of conv_igemm.hip keeps the rest.  Why a
// second main loop: the 128^2 / BK=32 loop stages 64 FLOP per byte through L2 -> LDS and pays a barrier every
// 32-deep K step; it tops out near 0.7 PFLOP/s (docs/DESIGN.md section 6).  This loop stages 128 FLOP per byte
// and never drains the load queue:
//
//   * wave (wr, wc), wr = wave >> 2, wc = wave & 3, owns 128 output channels x 64 pixels as 8 x 4 MFMA
//     tiles (v_mfma_f32_16x16x32_bf16; A operand = filter rows, B operand = pixel rows, so a lane ends up
//     with 4 consecutive channels of one pixel);
//   * a K step is FOUR phases of 16 MFMAs (one quadrant of the wave tile each); a phase is a LOAD segment
//     (the quadrant's fragment reads, two 1-KiB DMA pieces of a LATER K step, a counted vmcnt) and an MFMA
//     segment, each closed by a raw s_barrier;
//   * waves 4-7 run ONE barrier behind waves 0-3, so on every SIMD one wave is in its MFMA segment while
//     its partner is in its load segment (the two waves of a SIMD share the matrix pipe, so two MFMA
//     segments side by side would gain nothing; matrix beside memory does);
//   * each K tile is cut into four 16-KiB regions (filter rows of quadrants 0-1 / 2-3, pixel rows of
//     quadrants 0,3 / 1,2), each read from LDS in exactly ONE phase (the fragments then live in registers),
//     so a region of the buffer can be re-filled long before the whole tile is done: every DMA is issued
//     5-6 phases ahead of its first read and at least 2 phases after the last read of what it overwrites;
//     `s_waitcnt vmcnt(8)` at the end of every load segment retires exactly the region the NEXT phase reads
//     (8 = 2 pieces x the 4 regions issued since);
//   * LDS rows are 128 B (64 bf16); the 16-byte chunk index is XOR-ed with (row & 7): the ds_read_b128
//     fragment reads are bank-conflict free.  The DMA's LDS image is lane-linear, so the swizzle is applied
//     to the SOURCE address (lane -> chunk (lane & 7) ^ (lane >> 3));
//   * PERSISTENT: <= 256 workgroups each walk a list of work items (output tile, K range).  The four region
//     cursors that feed the DMA run ahead of the MFMAs THROUGH item boundaries, so the next tile's first
//     K steps are already in LDS when the current tile's last MFMA retires: no prologue latency per tile,
//     and the epilogue's stores drain behind the next tile's main loop instead of in front of an idle CU
//     (measured with one workgroup per tile: 11 us of fixed cost per 256 x 256 x 768 tile against 16 us of
//     main loop, scripts/one_gemm.py).  vmcnt counts stores too, in issue order, so the four load segments
//     that follow an epilogue allow its stores to stay outstanding (vmcnt(8 + stores)); every epilogue
//     issues EXACTLY the same number of store instructions (range-checked buffer stores, no branches);
//   * epilogue straight from registers: v_permlane16_swap pairs two adjacent channel tiles so that a lane
//     holds 8 consecutive channels (16 B) of one pixel - no LDS staging (the LDS belongs to the DMA running
//     ahead).  bias / activation / residual addend (+ bit mask) / GELU backward are applied there;
//   * split K (deep K, few tiles: the LM head's input gradient is 96 tiles x 786 K steps): an item is a
//     K range of a tile; partial tiles go to an fp32 workspace slab per split with plain 16-B stores and one
//     small kernel folds the slabs in a fixed order (deterministic, no atomics) and applies the epilogue;
//   * XCD-aware, grouped item order: the workgroups of one XCD (id mod 8) own a contiguous range of items,
//     walked 8 pixel tiles x 4.. channel tiles at a time: the 32 workgroups running on an XCD share 8 + 4
//     operand panels in its 4 MiB L2 instead of 1 + 32.
#include <stdlib.h>
#include <mutex>
#include <unordered_map>
#include "common.h"
#include "kernels.h"

namespace dpa {

namespace {

constexpr int WBK = 64;                    // K step (bf16 elements)
constexpr int ROWB = WBK * 2;              // bytes per LDS tile row
constexpr int TILE_B = 256 * ROWB;         // one operand tile: 32 KiB
constexpr int BUF_B = 2 * TILE_B;          // one K tile: filter rows, then pixel rows
constexpr int WIDE_LDS = 2 * BUF_B;        // 128 KiB
constexpr unsigned OOB = 0x80000000u;      // byte offset past every operand (operands are < 2^31 bytes)
#ifndef WIDE_EPI_JB
#define WIDE_EPI_JB 1
#endif
constexpr int GM = 8;                      // pixel tiles per group of the item order

typedef __attribute__((address_space(3))) unsigned char* lds_byte_ptr;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

// one 1-KiB LDS-DMA piece: 16 B per lane from buffer offset `voff` to LDS address `lds_addr` + 16 * lane
__device__ __forceinline__ void dma_piece(u32x4 rs4, unsigned lds_addr, unsigned voff) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds"
                 :: "v"(voff), "s"(rs4), "s"(lds_addr) : "memory", "m0");
}

struct WideArgs {
    int tiles_m, tiles_n, M, Kd;
    int splitk, nks;          // K steps per split (splitk == 1: nks = all)
    int items;                // tiles_m * tiles_n * splitk
    float* ws;                // splitk > 1: [splitk][M][N] fp32 partial sums
    int dbg;                  // timing experiments: bit 0 = drop the output stores (zero-record descriptor)
    int skew;                 // start-up skew: workgroup b sleeps b * skew / 256 periods of 8128 cycles before its first item
};

// NT = channel tiles (of 16) per wave row: the output tile is 256 pixels x (32 * NT) channels.  NT = 8 is the 256 x 256
// tile; NT = 6 (256 x 192) exists for tile QUANTISATION: a GEMM with 768 output channels over 16,384 tokens is 192 tiles
// of 256 x 256 - three quarters of one round on 256 CUs - but exactly 256 tiles of 256 x 192, and 2304 / 3072 channels
// are 12 / 16 such tiles (whole rounds again).  The LDS image, the DMA plan and every barrier / vmcnt count are those of
// NT = 8: LDS rows [wr * 128 + 16 * NT, wr * 128 + 128) of the filter tile are simply dead (their DMA pieces point past
// the operand and write zeros: no memory traffic), phases 3 and 4 of a K step issue 4 * (NT - 4) MFMAs instead of 16.
// GEN: the fused epilogues (bias / activation / residual addend / GELU backward / GELU twin) are compiled in.  The plain
// tile (LM head, plain input gradients) gets its own instantiation: with both epilogues in one kernel the general one's
// registers spilled into the plain path as well (fc 768 -> 3072 plain: 76 -> 104 us when the general epilogue grew).
// PREW: the filter fragments of channel tiles 0-3 are read ONE PHASE EARLIER - in phase 4 of the previous K step, which
// has no fragment reads of its own - into a second register set.  The four phases then read 4 / 4 / 8 / 8 fragments instead
// of 12 / 4 / 8 / 0: with one wave per SIMD in its load segment at a time, 12 x 1 KiB reads x 4 SIMDs are 384 LDS cycles
// against the 256 cycles of the MFMA segment they must hide behind; 8 are exactly 256.  (Legal under the refill
// discipline: filter region 0 of step k + 1 is retired by the vmcnt at the end of phase 3 of step k, read in phase 4, and
// refilled by the DMA of step k + 1's phase 3.)  Costs 32 registers: plain tiles only.
template <int NT, bool GEN, bool PREW>
__global__ __launch_bounds__(512, 2) void gemm_wide_kernel(ConvArgs a, WideArgs g) {
    static_assert(NT == 8 || NT == 6, "wave row = 4 + (NT - 4) channel tiles");
    constexpr int HALF = 16 * NT;                        // channels per wave row
    constexpr int BN = 2 * HALF;                         // channels per output tile
    extern __shared__ __attribute__((aligned(16))) unsigned char wsm[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int Nn = a.K;                                  // output channels = GEMM columns
    const int M = g.M, Kd = g.Kd, nk = Kd / WBK;

    // ---- this workgroup's items: the workgroups of one XCD own a contiguous range ----
    const int G = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3, nb = (G - xcd + 7) >> 3;
    const int iq = g.items >> 3, ir = g.items & 7;
    const int istart = xcd < ir ? xcd * (iq + 1) : ir * (iq + 1) + (xcd - ir) * iq;
    const int icount = iq + (xcd < ir ? 1 : 0);
    const int nseq = slot < icount ? (icount - slot + nb - 1) / nb : 0;
    if (nseq == 0) return;                               // workgroup-uniform
    // Start-up skew.  Items of one launch are equally long, so workgroups that start together reach their epilogues
    // together: 256 CUs x 128 KiB of stores every ~16 us arrive at HBM as one burst that takes ~5 us to drain, and vmcnt
    // retires in order, so each wave then sits on its next loads until its own stores are through.  Started a fraction
    // of an item apart, the CUs store at different times and the write stream to HBM is steady.
    for (int i = (bid * g.skew) >> 8; i > 0; --i) __builtin_amdgcn_s_sleep(127);
    auto item_coords = [&](int seq, int& tm, int& tn, int& z) {
        const int L = istart + slot + seq * nb;
        const int tile = L / g.splitk;
        z = L - tile * g.splitk;
        const int width = GM * g.tiles_n, grp = tile / width, first = grp * GM;
        const int gsz = min(g.tiles_m - first, GM), inner = tile - grp * width;
        tm = first + inner % gsz; tn = inner / gsz;
    };

    const unsigned xbytes = (unsigned)M * (unsigned)Kd * 2u, wbytes = (unsigned)Nn * (unsigned)Kd * 2u;
    const u32x4 xrs4 = buffer_words(a.x, xbytes), wrs4 = buffer_words(a.w, wbytes);
    const unsigned lds0 = (unsigned)(unsigned long long)(lds_byte_ptr)wsm;

    // ---- DMA plan of this wave: per region two pieces of 8 rows x 128 B ----
    // filter region R (0: channel tiles 0-3 of both wave rows, 1: tiles 4-7): 8-row group q = 2 * wave + u covers
    //   rows (q >> 3) * 128 + R * 64 + (q & 7) * 8 ..;  pixel region R (0: pixel tiles 0-1 of every wave column,
    //   1: tiles 2-3): rows (q >> 2) * 64 + R * 32 + (q & 3) * 8 ..
    const int lrow = lane >> 3;
    const unsigned src_chunk = (unsigned)((lane & 7) ^ lrow) * 16u;       // source-side swizzle (see header)
    // (formulas instead of arrays: an array indexed by a lambda parameter is demoted to scratch memory)
#define WG_WROW(R, U) ((((wave * 2 + (U)) >> 3) * 128) + (R) * 64 + (((wave * 2 + (U)) & 7) * 8))
#define WG_XROW(R, U) ((((wave * 2 + (U)) >> 2) * 64) + (R) * 32 + (((wave * 2 + (U)) & 3) * 8))
#define WG_WLDS(R, U) (lds0 + (unsigned)WG_WROW(R, U) * ROWB)
#define WG_XLDS(R, U) (lds0 + TILE_B + (unsigned)WG_XROW(R, U) * ROWB)
    // place cursor P (half R of both operands) at the first K step of item SEQ (plain locals and a macro: a struct
    // handed to a lambda by reference ended up in scratch memory)
    unsigned c0_w0, c0_w1, c0_x0, c0_x1, c1_w0, c1_w1, c1_x0, c1_x1, c0_par = 0u, c1_par = 0u;
    int c0_kt, c0_kend, c0_seq, c1_kt, c1_kend, c1_seq;
#define WG_SEEK(P, R, SEQ)                                                                        \
    {                                                                                             \
        P##_seq = (SEQ);                                                                          \
        if (P##_seq >= nseq) {                                                                    \
            P##_w0 = P##_w1 = P##_x0 = P##_x1 = OOB; P##_kt = 0; P##_kend = 0x7fffffff;           \
        } else {                                                                                  \
            int tm_, tn_, z_;                                                                     \
            item_coords(P##_seq, tm_, tn_, z_);                                                   \
            P##_kt = z_ * g.nks;                                                                  \
            P##_kend = min(nk, P##_kt + g.nks);                                                   \
            const int r0_ = WG_WROW(R, 0) + lrow, r1_ = WG_WROW(R, 1) + lrow;     /* LDS rows of the filter tile */ \
            const int w0_ = tn_ * BN + (r0_ >> 7) * HALF + (r0_ & 127), w1_ = tn_ * BN + (r1_ >> 7) * HALF + (r1_ & 127); \
            P##_w0 = ((r0_ & 127) < HALF && w0_ < Nn) ? (unsigned)w0_ * (unsigned)Kd * 2u + src_chunk : OOB; \
            P##_w1 = ((r1_ & 127) < HALF && w1_ < Nn) ? (unsigned)w1_ * (unsigned)Kd * 2u + src_chunk : OOB; \
            const int x0_ = tm_ * 256 + WG_XROW(R, 0) + lrow, x1_ = tm_ * 256 + WG_XROW(R, 1) + lrow; \
            P##_x0 = x0_ < M ? (unsigned)x0_ * (unsigned)Kd * 2u + src_chunk : OOB;               \
            P##_x1 = x1_ < M ? (unsigned)x1_ * (unsigned)Kd * 2u + src_chunk : OOB;               \
        }                                                                                         \
    }
    WG_SEEK(c0, 0, 0) WG_SEEK(c1, 1, 0)
#define WG_ADVANCE(P, R)                                                                          \
    { P##_par ^= 1u; if (++P##_kt == P##_kend) WG_SEEK(P, R, P##_seq + 1) }
#define WG_PIECES(P, RS4, LDS, R, OFF0, OFF1)                                                     \
    {                                                                                             \
        const unsigned kb_ = (unsigned)P##_kt * (unsigned)ROWB, bo_ = P##_par * (unsigned)BUF_B;  \
        dma_piece(RS4, LDS(R, 0) + bo_, P##OFF0 + kb_);                                           \
        dma_piece(RS4, LDS(R, 1) + bo_, P##OFF1 + kb_);                                           \
    }
#define WG_DMA_W0() WG_PIECES(c0, wrs4, WG_WLDS, 0, _w0, _w1)
#define WG_DMA_X0() WG_PIECES(c0, xrs4, WG_XLDS, 0, _x0, _x1)
#define WG_DMA_W1() WG_PIECES(c1, wrs4, WG_WLDS, 1, _w0, _w1)
#define WG_DMA_X1() WG_PIECES(c1, xrs4, WG_XLDS, 1, _x0, _x1)

    // ---- fragment addresses: lane reads row frow of a 16-row tile, chunk (ks * 4 + kq) ^ (frow & 7) ----
    const int frow = lane & 15, kq = lane >> 4;
    const unsigned fo0 = (unsigned)frow * ROWB + (unsigned)((kq ^ (frow & 7)) << 4);
    const unsigned fo1 = (unsigned)frow * ROWB + (unsigned)(((4 + kq) ^ (frow & 7)) << 4);
    const unsigned char* const wfb = wsm + (wr * 128) * ROWB;
    const unsigned char* const xfb = wsm + TILE_B + (wc * 64) * ROWB;

    floatx4 acc[NT][4];
    bf16x8 wf[4][2], wf0[PREW ? 4 : 1][2], xa[2][2], xb[2][2];
    const floatx4 zero4 = {0.f, 0.f, 0.f, 0.f};

#define WG_READ_WS(WF, WP, I0)                                                                    \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) if ((I0) + i < NT) {                            \
        WF[i][0] = *(const bf16x8*)((WP) + ((I0) + i) * (16 * ROWB) + fo0);                       \
        WF[i][1] = *(const bf16x8*)((WP) + ((I0) + i) * (16 * ROWB) + fo1);                       \
    }
#define WG_READ_W(I0) WG_READ_WS(wf, wp_, I0)
#define WG_READ_X(XF, J0)                                                                         \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                               \
        XF[j][0] = *(const bf16x8*)(xp_ + ((J0) + j) * (16 * ROWB) + fo0);                        \
        XF[j][1] = *(const bf16x8*)(xp_ + ((J0) + j) * (16 * ROWB) + fo1);                        \
    }
    // ZERO: first K step of an item - the ks = 0 MFMA starts from the inline constant 0 (no clearing pass)
#define WG_MMAS(WF, I0, J0, XF, ZERO)                                                             \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                              \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) if ((I0) + i < NT)                          \
            _Pragma("unroll") for (int j = 0; j < 2; ++j)                                         \
                acc[(I0) + i][(J0) + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(WF[i][ks], XF[j][ks], \
                    ((ZERO) && ks == 0) ? zero4 : acc[(I0) + i][(J0) + j], 0, 0, 0);
#define WG_MMA(I0, J0, XF, ZERO) WG_MMAS(wf, I0, J0, XF, ZERO)
    // end of a load segment: retire the region the next phase reads, meet, wait for this phase's fragments
#define WG_WAIT_8() asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    // first K step of an item: the previous item's epilogue stores (2 NT, or 4 NT partial-sum / twin stores) may stay in flight
#define WG_WAIT_SEL()                                                                             \
    if (vmsel == 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                              \
    else if (vmsel == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(8 + 2 * NT) : "memory");       \
    else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(8 + 4 * NT) : "memory");
#define WG_TO_MMA(WAIT)                                                                           \
    WAIT()                                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                            \
    __builtin_amdgcn_s_barrier();                                                                 \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                            \
    __builtin_amdgcn_sched_barrier(0);                                                            \
    __builtin_amdgcn_s_setprio(1);
#define WG_TO_LOAD()                                                                              \
    __builtin_amdgcn_s_setprio(0);                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                            \
    __builtin_amdgcn_s_barrier();                                                                 \
    __builtin_amdgcn_sched_barrier(0);
    // one K step = four phases; VM = operations that may stay outstanding at the end of a load segment
#define WG_KSTEP(ZERO, VM)                                                                        \
    {                                                                                             \
        const unsigned char* const wp_ = wfb + rpar * BUF_B;                                      \
        const unsigned char* const xp_ = xfb + rpar * BUF_B;                                      \
        if constexpr (!PREW) { WG_READ_W(0) }                                                     \
        WG_READ_X(xa, 0)                                                                          \
        WG_DMA_X1()                                                                               \
        WG_TO_MMA(VM)                                                                             \
        if constexpr (PREW) { WG_MMAS(wf0, 0, 0, xa, ZERO) } else { WG_MMA(0, 0, xa, ZERO) }      \
        WG_TO_LOAD()                                                                              \
        WG_READ_X(xb, 2)                                                                          \
        WG_DMA_W1()                                                                               \
        WG_TO_MMA(VM)                                                                             \
        if constexpr (PREW) { WG_MMAS(wf0, 0, 2, xb, ZERO) } else { WG_MMA(0, 2, xb, ZERO) }      \
        WG_TO_LOAD()                                                                              \
        WG_READ_W(4)                                                                              \
        WG_DMA_W0()                                                                               \
        WG_TO_MMA(VM)                                                                             \
        WG_MMA(4, 2, xb, ZERO)                                                                    \
        WG_TO_LOAD()                                                                              \
        if constexpr (PREW) { WG_READ_WS(wf0, wfb + (rpar ^ 1u) * BUF_B, 0) }  /* tiles 0-3 of the NEXT K step */ \
        WG_DMA_X0()                                                                               \
        WG_TO_MMA(VM)                                                                             \
        WG_MMA(4, 0, xa, ZERO)                                                                    \
        WG_TO_LOAD()                                                                              \
        rpar ^= 1u;                                                                               \
        WG_ADVANCE(c1, 1) WG_ADVANCE(c0, 0)                                                       \
    }

    // ---- prologue: all of the first K step, and the two regions of the second that phases 2-3 of a step issue ----
    WG_DMA_W0() WG_DMA_X0() WG_DMA_X1() WG_DMA_W1()
    WG_ADVANCE(c0, 0) WG_ADVANCE(c1, 1)
    WG_DMA_W0() WG_DMA_X0()
    WG_ADVANCE(c0, 0)
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");     // filter region 0 and pixel region 0 of step 0 have landed
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();           // waves 4-7 run one barrier behind
    if constexpr (PREW) { WG_READ_WS(wf0, wfb, 0) }      // tiles 0-3 of step 0 (every later step: phase 4 of its predecessor)

    const bool gelu_bwd = a.mode == 1 && a.epi == 5;
    const void* pre_src = gelu_bwd ? a.ep_src : a.addend;
    const unsigned ybytes = (unsigned)M * (unsigned)Nn * 2u;
    const int q4 = kq * 4;
    unsigned rpar = 0u;
    int vmsel = 0;

    const int row = kq;                                               // 16-lane row of this lane
    const bool twin = GEN && a.mode == 0 && a.y_act != nullptr;       // fprop: second output gelu(y) (32 stores per item)
    // The fused operands are requested in as few round trips as the ~64 registers freed by the dead fragments allow: the
    // bias slice of the tile ONCE per item as eight 16-byte loads (it used to be 4 x 32 dword loads per item and lane),
    // the residual / GELU operand of TWO pixel tiles at a time (two round trips per item instead of four).  Measured on
    // GPT-2's GEMMs at 16,384 tokens (scripts/bench_gemm.py EPI=1; before -> after in profiles/r3/gemm_epilogue_ab.txt).
    constexpr int JB = WIDE_EPI_JB;                                    // pixel tiles whose fused operands are requested together
#define WG_EPILOGUE(GENERAL)                                                                      \
    {                                                                                             \
        const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, (g.dbg & 1) ? 0u : ybytes, 0x00020000); \
        const __amdgpu_buffer_rsrc_t trs = __builtin_amdgcn_make_buffer_rsrc(twin ? a.y_act : a.y, 0, (g.dbg & 1) ? 0u : ybytes, 0x00020000); \
        _Pragma("unroll") for (int jh = 0; jh < 4 / JB; ++jh) {                                        \
            unsigned eo[JB][NT / 2];                                                               \
            u32x4 pre[JB][NT / 2];                                                                 \
            unsigned pmk[JB][NT / 2];                                                              \
            floatx4 bias4[NT];     /* the tile's bias slice: eight 16-byte loads (L1 hits after the first pixel tile) */ \
            if ((GENERAL) && a.bias != nullptr) {                                                 \
                const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc((void*)a.bias, 0, (unsigned)Nn * 4u, 0x00020000); \
                _Pragma("unroll") for (int i = 0; i < NT; ++i)    /* (Nn % 8 == 0: 4 channels are in range together) */ \
                    bias4[i] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(brs, (unsigned)(c0 + i * 16 + q4) * 4u, 0, 0)); \
            }                                                                                     \
            _Pragma("unroll") for (int jj = 0; jj < JB; ++jj) {                                    \
                const int px = m0 + (jh * JB + jj) * 16 + frow;                                    \
                _Pragma("unroll") for (int ip = 0; ip < NT / 2; ++ip) {                           \
                    const int ch = c0 + 16 * (2 * ip + (row & 1)) + 8 * (row >> 1);               \
                    eo[jj][ip] = (px < M && ch < Nn) ? (unsigned)px * (unsigned)Nn + (unsigned)ch : ~0u; \
                }                                                                                 \
            }                                                                                     \
            if ((GENERAL) && pre_src != nullptr) {                                                \
                const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc((void*)pre_src, 0, ybytes, 0x00020000); \
                _Pragma("unroll") for (int jj = 0; jj < JB; ++jj)                                  \
                    _Pragma("unroll") for (int ip = 0; ip < NT / 2; ++ip)                         \
                        pre[jj][ip] = __builtin_amdgcn_raw_buffer_load_b128(srs, eo[jj][ip] != ~0u ? eo[jj][ip] * 2u : 0xFFFFFFF0u, 0, 0); \
                if (!gelu_bwd && a.addend_mask != nullptr) {                                      \
                    const __amdgpu_buffer_rsrc_t mrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.addend_mask, 0, ybytes >> 4, 0x00020000); \
                    _Pragma("unroll") for (int jj = 0; jj < JB; ++jj)                              \
                        _Pragma("unroll") for (int ip = 0; ip < NT / 2; ++ip)                     \
                            pmk[jj][ip] = __builtin_amdgcn_raw_buffer_load_b8(mrs, eo[jj][ip] != ~0u ? eo[jj][ip] >> 3 : 0xFFFFFFF0u, 0, 0); \
                } else {                                                                          \
                    _Pragma("unroll") for (int jj = 0; jj < JB; ++jj)                              \
                        _Pragma("unroll") for (int ip = 0; ip < NT / 2; ++ip) pmk[jj][ip] = 0xFFu; \
                }                                                                                 \
            }                                                                                     \
            _Pragma("unroll") for (int jj = 0; jj < JB; ++jj)                                      \
            _Pragma("unroll") for (int ip = 0; ip < NT / 2; ++ip) {                               \
                const int j = jh * JB + jj;                                                        \
                unsigned d[2][2];                                                                 \
                _Pragma("unroll") for (int h = 0; h < 2; ++h) {                                   \
                    const int i = 2 * ip + h;                                                     \
                    float v[4];                                                                   \
                    _Pragma("unroll") for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r];            \
                    if (GENERAL) {                                                                \
                        if (a.bias != nullptr) {                                                  \
                            _Pragma("unroll") for (int r = 0; r < 4; ++r) v[r] += bias4[i][r];    \
                        }                                                                         \
                        if (a.act == 1) {                                                         \
                            _Pragma("unroll") for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f); \
                        } else if (a.act == 2) {                                                  \
                            _Pragma("unroll") for (int r = 0; r < 4; ++r) v[r] = gelu_tanh_fast(v[r]); \
                        }                                                                         \
                    }                                                                             \
                    d[h][0] = pack_bf16x2(v[0], v[1]);                                            \
                    d[h][1] = pack_bf16x2(v[2], v[3]);                                            \
                }                                                                                 \
                /* odd rows of the first operand <-> even rows of the second: afterwards a lane in row r holds channels */ \
                /* 16 * (2 ip + (r & 1)) + 8 * (r >> 1) .. + 7 of its pixel as (s0[0], s1[0], s0[1], s1[1]) */ \
                const u32x2 s0 = __builtin_amdgcn_permlane16_swap(d[0][0], d[1][0], false, false); \
                const u32x2 s1 = __builtin_amdgcn_permlane16_swap(d[0][1], d[1][1], false, false); \
                uint4 v = make_uint4(s0[0], s1[0], s0[1], s1[1]);                                 \
                if ((GENERAL) && pre_src != nullptr) {                                            \
                    float f[8], gx[8];                                                            \
                    unpack8(v, f); unpack8(__builtin_bit_cast(uint4, pre[jj][ip]), gx);           \
                    if (gelu_bwd) {                                                               \
                        _Pragma("unroll") for (int e = 0; e < 8; ++e) f[e] *= gelu_tanh_grad_fast(gx[e]); \
                    } else {                                                                      \
                        const unsigned mk = pmk[jj][ip];                                          \
                        _Pragma("unroll") for (int e = 0; e < 8; ++e) f[e] += ((mk >> e) & 1u) ? gx[e] : 0.f; \
                    }                                                                             \
                    v = pack8(f);                                                                 \
                }                                                                                 \
                const unsigned so_ = eo[jj][ip] != ~0u ? eo[jj][ip] * 2u : 0xFFFFFFF0u;           \
                __builtin_amdgcn_raw_buffer_store_b128(u32x4{v.x, v.y, v.z, v.w}, yrs, so_, 0, 0); \
                if ((GENERAL) && twin) {   /* second output gelu(y), from the bf16 values just stored (what the GELU's backward sees) */ \
                    float f[8];                                                                   \
                    unpack8(v, f);                                                                \
                    _Pragma("unroll") for (int e = 0; e < 8; ++e) f[e] = gelu_tanh_fast(f[e]);    \
                    const uint4 t = pack8(f);                                                     \
                    __builtin_amdgcn_raw_buffer_store_b128(u32x4{t.x, t.y, t.z, t.w}, trs, so_, 0, 0); \
                }                                                                                 \
            }                                                                                     \
        }                                                                                         \
    }

    for (int seq = 0; seq < nseq; ++seq) {
        int tm, tn, z;
        item_coords(seq, tm, tn, z);
        const int k0 = z * g.nks, steps = min(nk, k0 + g.nks) - k0;
        WG_KSTEP(true, WG_WAIT_SEL)
        for (int s = 1; s < steps; ++s) WG_KSTEP(false, WG_WAIT_8)
        vmsel = (g.splitk == 1 && !twin) ? 1 : 2;                    // stores the next item's first K step may leave in flight: 2 NT or 4 NT

        // Both wave groups write their tile out in the SAME barrier interval: waves 0-3 (one barrier ahead) sit out the
        // interval in which waves 4-7 issue their last MFMAs, and waves 4-7 drop back behind afterwards.  Left staggered,
        // each group's write-out (~4k cycles of VALU and store issue) held the other group at its next barrier: twice
        // the cost per item.
        if (wr == 0) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const int m0 = tm * 256 + wc * 64, c0 = tn * BN + wr * HALF;
        if (g.splitk > 1) {
            // ---- partial sums: this lane's 4 consecutive channels of a pixel = one 16-byte fp32 store ----
            float* const slab = g.ws + (size_t)z * (size_t)M * (size_t)Nn;
            const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(slab, 0, (unsigned)M * (unsigned)Nn * 4u, 0x00020000);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int px = m0 + j * 16 + frow;
#pragma unroll
                for (int i = 0; i < NT; ++i) {
                    const int ch = c0 + i * 16 + q4;
                    const unsigned off = (px < M && ch < Nn) ? ((unsigned)px * (unsigned)Nn + (unsigned)ch) * 4u : 0xFFFFFFF0u;
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[i][j]), prs, off, 0, 0);
                }
            }
        } else
        // ---- epilogue from registers: bias / activation in fp32, pack, pair adjacent channel tiles so that a
        //      lane holds 8 consecutive channels: 16 range-checked 16-byte stores per lane, no branches.
        //      Expanded once for the plain tile (BatchNorm-free GEMMs: the LM head) and once in general form ----
        WG_EPILOGUE(GEN)
        __builtin_amdgcn_sched_barrier(0);
        if (wr == 1) __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the run-ahead pieces (zeros) must land before the LDS is released
    if (wr == 0) __builtin_amdgcn_s_barrier();           // waves 0-3 catch up: barrier counts are equal again
    __syncthreads();

#undef WG_ADVANCE
#undef WG_EPILOGUE
#undef WG_SEEK
#undef WG_WROW
#undef WG_XROW
#undef WG_WLDS
#undef WG_XLDS
#undef WG_PIECES
#undef WG_WAIT_8
#undef WG_WAIT_SEL
#undef WG_DMA_W0
#undef WG_DMA_W1
#undef WG_DMA_X0
#undef WG_DMA_X1
#undef WG_READ_W
#undef WG_READ_WS
#undef WG_MMAS
#undef WG_READ_X
#undef WG_MMA
#undef WG_TO_MMA
#undef WG_TO_LOAD
#undef WG_KSTEP
}

// Fold the split-K slabs in a fixed order and apply the epilogue: one thread = 8 consecutive channels of a pixel.
__global__ __launch_bounds__(256) void gemm_splitk_fold_kernel(ConvArgs a, const float* __restrict__ ws, int splitk, long MN, int Nn) {
    const bool gelu_bwd = a.mode == 1 && a.epi == 5;
    const void* pre_src = gelu_bwd ? a.ep_src : a.addend;
    for (long e = ((long)blockIdx.x * 256 + threadIdx.x) * 8; e < MN; e += (long)gridDim.x * 256 * 8) {
        float f[8];
        {
            const float4 p0 = *(const float4*)(ws + e), p1 = *(const float4*)(ws + e + 4);
            f[0] = p0.x; f[1] = p0.y; f[2] = p0.z; f[3] = p0.w; f[4] = p1.x; f[5] = p1.y; f[6] = p1.z; f[7] = p1.w;
        }
        for (int z = 1; z < splitk; ++z) {
            const float4 p0 = *(const float4*)(ws + z * MN + e), p1 = *(const float4*)(ws + z * MN + e + 4);
            f[0] += p0.x; f[1] += p0.y; f[2] += p0.z; f[3] += p0.w; f[4] += p1.x; f[5] += p1.y; f[6] += p1.z; f[7] += p1.w;
        }
        const int ch = (int)(e % Nn);
        if (a.bias != nullptr) {
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] += a.bias[ch + j];
        }
        if (a.act == 1) {
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = fmaxf(f[j], 0.f);
        } else if (a.act == 2) {
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = gelu_tanh(f[j]);
        }
        uint4 v = pack8(f);
        if (pre_src != nullptr) {
            float gx[8];
            unpack8(v, f); unpack8(*(const uint4*)((const unsigned short*)pre_src + e), gx);
            if (gelu_bwd) {
#pragma unroll
                for (int j = 0; j < 8; ++j) f[j] *= gelu_tanh_grad_fast(gx[j]);
            } else {
                const unsigned mk = a.addend_mask != nullptr ? a.addend_mask[e >> 3] : 0xFFu;
#pragma unroll
                for (int j = 0; j < 8; ++j) f[j] += ((mk >> j) & 1u) ? gx[j] : 0.f;
            }
            v = pack8(f);
        }
        *(uint4*)((unsigned short*)a.y + e) = v;
    }
}

int g_wide_min_tiles = -1;      // -1: not yet read from the environment
int g_wide_splitk = -1;         // -1: not yet read; 0 = choose per launch; n > 0 = force

// Split-K workspace.  With an allocator registered (bindings.cpp hands in torch's caching allocator), every split-K launch
// takes a block for the launch's stream and gives it back right after the fold kernel has been enqueued: the caching
// allocator only re-issues it to later work of the same stream, so the memory is visible to torch.cuda.memory_*, shared
// with every other temporary of the step, and never held while idle.  Without one (stand-alone harnesses): a grow-only
// block per stream from hipMalloc.
WorkspaceAlloc g_ws_alloc = nullptr;
WorkspaceFree g_ws_free = nullptr;
struct Workspace { float* ptr = nullptr; size_t bytes = 0; };
std::mutex g_ws_mutex;
std::unordered_map<hipStream_t, Workspace> g_ws;

float* workspace_for(hipStream_t st, size_t bytes) {
    if (g_ws_alloc != nullptr) {
        float* p = (float*)g_ws_alloc(bytes, st);
        if (p == nullptr) {
            fprintf(stderr, "gemm_wide: the registered allocator returned no %zu-byte split-K workspace\n", bytes);
            abort();
        }
        return p;
    }
    std::lock_guard<std::mutex> lock(g_ws_mutex);
    Workspace& w = g_ws[st];
    if (w.bytes < bytes) {
        if (w.ptr != nullptr) { (void)hipStreamSynchronize(st); (void)hipFree(w.ptr); }
        w.ptr = nullptr; w.bytes = 0;
        if (hipMalloc((void**)&w.ptr, bytes) != hipSuccess) {
            fprintf(stderr, "gemm_wide: cannot allocate %zu bytes of split-K workspace\n", bytes);
            abort();
        }
        w.bytes = bytes;
    }
    return w.ptr;
}

// Split factor of a launch: a cost model in microseconds calibrated on scripts/one_gemm.py (1.35 us per K step per
// workgroup on the 256 x 256 tile - `kstep_us` - and ~1 us per item, slabs written and read back at ~4 TB/s); 1 = no
// split.  *model_us receives the modelled time of the choice.
int choose_splitk(long tiles, int nk, long MN, double kstep_us = 1.35, double* model_us = nullptr, double item_us = 1.0) {
    if (g_wide_splitk < 0) {
        const char* e = getenv("DPA_WIDE_SPLITK");
        g_wide_splitk = e ? atoi(e) : 0;
        if (g_wide_splitk < 0) g_wide_splitk = 0;
    }
    int lo = 1, hi = 16;
    if (g_wide_splitk > 0) lo = hi = g_wide_splitk;
    double best = 1e30;
    int best_s = 1;
    for (int s = lo; s <= hi; ++s) {
        const int nks = (nk + s - 1) / s;
        if (s > 1 && ((long)(s - 1) * nks >= nk || nks < 4)) continue;     // every split non-empty and worth an item
        if (s > 1 && (double)s * MN * 4.0 >= 4.0e9) continue;
        const long rounds = (tiles * s + 255) / 256;
        // K ranges of one tile do not share operand panels in L2 the way neighbouring tiles at the SAME K offset do: a split
        // K step was measured at 1.7 - 2.5 us (LM-head input gradient, 8,192 / 16,384 tokens) against 1.2 - 1.35 unsplit
        double t = rounds * (nks * kstep_us * (s > 1 ? 1.6 : 1.0) + (s > 1 ? 1.0 : item_us));
        if (s > 1) t += (s + 0.5) * MN * 4.0 / 4.0e6;
        if (t < best * 0.97) { best = t; best_s = s; }
    }
    if (model_us != nullptr) *model_us = best;
    return best_s;
}

int g_wide_bn = -1;             // -1: not yet read; 0 = choose per launch; 256 / 192 = force

// Tile width (output channels per tile) and split factor of a launch: whichever of 256 x 256 and 256 x 192 the cost model
// prices lower - the narrow tile's K step is 48 MFMAs per wave instead of 64 but only ~0.9 of the time (measured), so what decides
// is the number of ROUNDS the tiles make on 256 CUs (768 channels x 16,384 tokens: 0.75 round of wide tiles, one whole
// round of narrow ones).  DPA_WIDE_BN=256|192 forces one.
// `epi`: 0 = plain tile, 1 = bias / activation, 2 = a second operand or output (residual, GELU backward, GELU twin).  The
// fused epilogue of the 256-wide tile runs out of registers (128 accumulators + the operands in flight: spill code whose
// scratch accesses drain vmcnt) and costs about twice the narrow tile's per item - measured at 16,384 tokens, K = 768:
// bias +8.4 vs +3.4 us per item, residual +17 vs +6.5, GELU backward +18.6 vs +9.3 (profiles/r3/gemm_epilogue_ab.txt).
struct WidePlan { int bn, tiles_m, tiles_n, splitk; };
WidePlan plan_wide(long M, int Nn, int nk, int epi) {
    if (g_wide_bn < 0) {
        const char* e = getenv("DPA_WIDE_BN");
        g_wide_bn = e ? atoi(e) : 0;
        if (g_wide_bn != 192 && g_wide_bn != 256) g_wide_bn = 0;
    }
    WidePlan best{};
    double best_us = 1e30;
    for (int bn = 256; bn >= 192; bn -= 64) {
        if (g_wide_bn != 0 && bn != g_wide_bn) continue;
        WidePlan p{bn, (int)((M + 255) / 256), (Nn + bn - 1) / bn, 1};
        double us = 0.0;
        const double item_us = 1.0 + (epi == 0 ? 0.0 : epi == 1 ? (bn == 256 ? 8.0 : 3.5) : (bn == 256 ? 17.0 : 8.0));
        p.splitk = choose_splitk((long)p.tiles_m * p.tiles_n, nk, M * Nn, bn == 256 ? 1.30 : 1.18, &us, item_us);
        if (us < best_us * 0.97) { best_us = us; best = p; }       // (ties go to the wide tile)
    }
    return best;
}

}  // namespace

// Smallest number of 256 x 256 work items for which conv_igemm() prefers this kernel (DPA_WIDE_MIN_TILES; 0 = never).
void set_wide_gemm_min_tiles(int n) { g_wide_min_tiles = n < 0 ? 0 : n; }
int wide_gemm_min_tiles() {
    if (g_wide_min_tiles < 0) {
        const char* e = getenv("DPA_WIDE_MIN_TILES");
        g_wide_min_tiles = e ? atoi(e) : 256;
        if (g_wide_min_tiles < 0) g_wide_min_tiles = 0;
    }
    return g_wide_min_tiles;
}
void set_wide_gemm_splitk(int n) { g_wide_splitk = n < 0 ? 0 : n; }
void set_wide_gemm_bn(int n) { g_wide_bn = (n == 192 || n == 256) ? n : 0; }

// A launch the wide kernel can take: a plain GEMM (1x1, stride 1, no padding) with K a multiple of 64, one of the
// epilogues implemented above, operands addressable by 31-bit byte offsets.
bool wide_gemm_supported(const ConvArgs& a) {
    if (!(a.R == 1 && a.S == 1 && a.pad == 0 && a.stride == 1)) return false;
    if (a.mode != 0 && a.mode != 1) return false;
    if (a.C % WBK != 0 || a.K % 8 != 0 || a.y == nullptr) return false;
    if (a.stats != nullptr) return false;
    if (a.y_act != nullptr && (a.mode != 0 || a.act != 0)) return false;      // the GELU twin: forward, linear epilogue
    if (!(a.epi == 0 || (a.mode == 1 && a.epi == 5))) return false;
    const long M = (long)a.N * a.Ho * a.Wo;
    if (M * a.C * 2 >= (1l << 31) || (long)a.K * a.C * 2 >= (1l << 31) || M * a.K >= (1l << 31) - 16) return false;
    return true;
}

// Where the wide kernel wins (scripts/bench_gemm.py, one process, both kernels on the same operands): deep K
// (>= 768: the per-item cost - a 128-KiB store burst per CU that in-order vmcnt cannot hide - is amortised over >= 12
// K steps) and a grid that fills WHOLE rounds of the 256 CUs; split K only for very deep K.  The thresholds are lifted
// by set_wide_gemm_min_tiles(1) (tests: every supported launch).
// History of the default: 700 items (round 2, batch 8 x 1024: GPT-2's 288-384-tile qkv / fc launches made 1.1-1.5
// rounds of 256 x 256 tiles, and a persistent one-workgroup-per-CU kernel then runs two full tile times - 621k vs 662k
// tok/s).  Round 3: with the 256 x 192 tile (plan_wide) the six mid-size GEMMs of a block are whole rounds at 16 x 1024
// tokens (proj / fc2 / their input gradients: exactly 256 tiles) and run at 0.81-1.25 PFLOP/s alone against 0.63-0.86
// on the 128 x 128 kernel (profiles/r3/gemm_bench_r3.md); inside the step, one box, interleaved: 789k (700 items) vs
// 822k tok/s (256 items) at 16 x 1024 and 798k vs 832k at 32 x 1024.  Hence 256: at least one item per CU.
int wide_epilogue_kind(const ConvArgs& a) {
    const bool gelu_bwd = a.mode == 1 && a.epi == 5;
    if (a.y_act != nullptr || (gelu_bwd ? a.ep_src : a.addend) != nullptr) return 2;
    return (a.bias != nullptr || a.act != 0) ? 1 : 0;
}

bool wide_gemm_eligible(const ConvArgs& a) {
    const int min_tiles = wide_gemm_min_tiles();
    if (min_tiles <= 0 || !wide_gemm_supported(a)) return false;
    const long M = (long)a.N * a.Ho * a.Wo;
    const int nk = a.C / WBK;
    const WidePlan plan = plan_wide(M, a.K, nk, wide_epilogue_kind(a));
    const long tiles = (long)plan.tiles_m * plan.tiles_n;
    const int s = plan.splitk;
    if (a.y_act != nullptr && s > 1) return false;          // the twin is written by the tile epilogue, not by the split-K fold
    if (min_tiles == 1) return true;
    if (nk < 12) return false;
    if (s > 1 && nk < 96) return false;
    // (scripts/bench_gemm.py times launches BACK TO BACK: the idle CUs of a partial last round are filled by the next
    // launch of the loop; inside a step the next kernel depends on this one.  DPA_WIDE_FWD=1: forward launches from 192
    // tiles, the round-2 experiment.)
    static const bool fwd_wide = [] { const char* e = getenv("DPA_WIDE_FWD"); return e && atoi(e) != 0; }();   // experiment switch
    if (fwd_wide && a.mode == 0 && s == 1 && tiles >= 192) return true;
    // enough work for several items per workgroup, counted in tile-K-steps so that the answer does not depend on the
    // split factor the cost model happens to pick (the LM head's input gradient is 96 tiles x 786 K-steps: at split 5
    // it is 480 items and fell back to the 128 x 128 kernel at 1000 us instead of 554 us)
    return tiles * s >= min_tiles || tiles * (long)nk >= 40L * min_tiles;
}

void gemm_wide(const ConvArgs& a, hipStream_t st) {
    if (!wide_gemm_supported(a)) {
        fprintf(stderr, "gemm_wide: unsupported launch (host dispatch error)\n");
        abort();
    }
    static bool allowed = false;
    if (!allowed) {
        for (const void* fn : {(const void*)gemm_wide_kernel<8, false, false>, (const void*)gemm_wide_kernel<6, false, false>,
                               (const void*)gemm_wide_kernel<8, false, true>, (const void*)gemm_wide_kernel<6, false, true>,
                               (const void*)gemm_wide_kernel<8, true, false>, (const void*)gemm_wide_kernel<6, true, false>}) {
            const hipError_t err = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, WIDE_LDS);
            if (err != hipSuccess) {
                fprintf(stderr, "gemm_wide: cannot allow %d bytes of LDS: %s\n", WIDE_LDS, hipGetErrorString(err));
                abort();
            }
        }
        allowed = true;
    }
    const long M = (long)a.N * a.Ho * a.Wo;
    WideArgs g{};
    const int nk = a.C / WBK;
    const WidePlan plan = plan_wide(M, a.K, nk, wide_epilogue_kind(a));
    g.tiles_m = plan.tiles_m; g.tiles_n = plan.tiles_n;
    g.M = (int)M; g.Kd = a.C;
    const long tiles = (long)g.tiles_m * g.tiles_n;
    g.splitk = plan.splitk;
    if (a.y_act != nullptr && g.splitk > 1) {
        fprintf(stderr, "gemm_wide: the GELU twin is written by the tile epilogue and cannot be combined with split K\n");
        abort();
    }
    g.nks = (nk + g.splitk - 1) / g.splitk;
    g.items = (int)(tiles * g.splitk);
    {
        static const int skew_on = [] { const char* e = getenv("DPA_WIDE_SKEW"); return e ? atoi(e) : 1; }();
        // one item = nks K steps of ~2,700 cycles; spread the starts over one item when every workgroup has several
        g.skew = (skew_on && g.items >= 3 * 256) ? (int)((long)g.nks * 2700 * skew_on / 8128) : 0;
    }
    { static const int dbg = [] { const char* e = getenv("DPA_WIDE_DBG"); return e ? atoi(e) : 0; }(); g.dbg = dbg; }
    g.ws = g.splitk > 1 ? workspace_for(st, (size_t)g.splitk * (size_t)M * (size_t)a.K * 4) : nullptr;
    const int grid = g.items < 256 ? g.items : 256;
    const bool plain = wide_epilogue_kind(a) == 0;
    // measured: LM-head forward 1166 -> 1148 us, every other shape and the GPT-2 step within noise (profiles/r3/gemm_prew_ab.txt):
    // the load segment is not bound by its LDS read volume.  Off by default.
    static const int prew = [] { const char* e = getenv("DPA_WIDE_PREW"); return e ? atoi(e) : 0; }();
#define WG_LAUNCH(NT_, GEN_, PREW_) hipLaunchKernelGGL((gemm_wide_kernel<NT_, GEN_, PREW_>), dim3(grid), dim3(512), WIDE_LDS, st, a, g)
    if (plan.bn == 256) {
        if (!plain) WG_LAUNCH(8, true, false); else if (prew) WG_LAUNCH(8, false, true); else WG_LAUNCH(8, false, false);
    } else {
        if (!plain) WG_LAUNCH(6, true, false); else if (prew) WG_LAUNCH(6, false, true); else WG_LAUNCH(6, false, false);
    }
#undef WG_LAUNCH
    if (g.splitk > 1) {
        const long MN = M * a.K;
        long blocks = (MN / 8 + 255) / 256;
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(gemm_splitk_fold_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a, (const float*)g.ws, g.splitk, MN, a.K);
        if (g_ws_alloc != nullptr && g_ws_free != nullptr) g_ws_free(g.ws);      // stream-ordered: reusable behind the fold
    }
}

void* launcher_workspace(size_t bytes, hipStream_t st) { return workspace_for(st, bytes); }
void launcher_workspace_release(void* p) {
    if (g_ws_alloc != nullptr && g_ws_free != nullptr) g_ws_free(p);      // stream-ordered: reusable behind what was enqueued
}

void set_workspace_allocator(WorkspaceAlloc alloc, WorkspaceFree free_) {
    g_ws_alloc = alloc;
    g_ws_free = free_;
}

}  // namespace dpa
