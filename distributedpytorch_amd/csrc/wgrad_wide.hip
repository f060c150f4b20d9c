// Wide-tile weight gradient of Linear layers for gfx950, several layers per launch:
//
//     dw_p[n][k] += sum_m dy_p[m][n] * x_p[m][k]        p = 0 .. nprob-1, all over the same M tokens
//     dbias_p[n] += sum_m dy_p[m][n]                    (optional)
//
// Both operands are TOKEN-major (the reduction index m is the slow axis), so this is the "TN" twin of gemm_wide.hip:
// the same persistent 8-wave workgroup with 128 KiB of LDS, the same four-phase K step with LDS-DMA pieces kept in
// flight behind counted vmcnt and waves 4-7 one barrier behind waves 0-3 - but the tiles lie in LDS as [token][column]
// and the MFMA fragments come out of gfx950's transposing LDS read (ds_read_b64_tr_b16), as in conv_wgrad.hip.
//
// Why it exists: conv_wgrad.hip (128 x 128 tiles, 4 waves, one launch per layer, float atomics over a token split)
// runs GPT-2's four weight gradients of a block at ~0.6 PFLOP/s - 4.6 ms of a 20 ms step at 16 x 1024 tokens.  A
// 256 x 256 tile halves the LDS fragment traffic per MFMA, and ONE launch for the four layers of a block
// (27 + 9 + 36 + 36 = 108 tiles) needs only a 2-way token split to give every CU one item of 128 K steps - a launch per
// layer needs 7-9-way splits, and every split flushes a 256 KiB fp32 tile per CU.
//
//   * K step = 64 tokens.  LDS per K step and operand: 64 token rows x 256 columns bf16 = 32 KiB, as two REGIONS of
//     16 KiB (dy: the columns of channel tiles 0-3 / 4-7 of both wave rows; x: column tiles 0-1 / 2-3 of every wave
//     column), each read in exactly one phase - the refill discipline of gemm_wide.hip carries over unchanged.
//   * a region is 16 DMA pieces of 1 KiB = 4 token rows x 256 B; wave w issues pieces 2w, 2w+1 of every region (2 per
//     region per K step, 8 per K step: the vmcnt arithmetic of gemm_wide.hip).  Inside a token row the 32-byte window
//     (= the 16 columns of one MFMA tile) index is XOR-ed with f(row) = (row & 3) | ((row >> 3) & 1) << 2, applied on
//     the SOURCE side (the DMA's LDS image is lane-linear): the 8 token rows one half-wave's transposed read touches
//     ({t, t+8}, t = 0..3 or 4..7) land in 8 different 32-byte windows of the 256-byte bank row.
//   * MFMA 16x16x32: A operand = x fragment (rows of D = k), B operand = dy fragment (columns of D = n), so a lane ends
//     up with 4 CONSECUTIVE k of one n: one 16-byte fp32 access of dw.
//   * items = (layer, token range z, tile); an item that covers all tokens adds its tile into dw itself (one writer per
//     element: plain 16-byte loads and stores, no atomics); with a token split the partial tiles go to an fp32 slab in
//     register order (8 KiB contiguous per store instruction and workgroup) and wgrad_wide_fold_kernel adds them into dw
//     in a fixed order - deterministic, unlike the float atomics of conv_wgrad.hip.
//   * dbias: in items of the first k tile, wave (wr, wc) also multiplies dy tiles 2 wc, 2 wc + 1 by a fragment of ones
//     (4 extra MFMAs per K step and wave); 16 lanes per tile add the sums with float atomics (N floats per item).
//   * XCD-aware item order as in gemm_wide.hip: the workgroups of one XCD own a contiguous item range, tiles of one
//     (layer, z) walked k tile fastest, so the 27-32 items running behind one L2 share a handful of operand panels.
#include <stdlib.h>
#include <stdio.h>
#include "common.h"
#include "kernels.h"

namespace dpa {

namespace {

constexpr int WBK = 64;                    // tokens per K step
constexpr int REG_B = 16384;               // one region: 16 pieces of 1 KiB
constexpr int OP_B = 2 * REG_B;            // one operand tile
constexpr int BUF_B = 2 * OP_B;            // dy tile, then x tile
constexpr int WW_LDS = 2 * BUF_B;          // 128 KiB
constexpr unsigned OOB = 0x80000000u;      // byte offset past every operand (operands are < 2^31 bytes)
constexpr int TILE_F = 256 * 256;          // floats per output tile

typedef __attribute__((address_space(3))) unsigned char* lds_byte_ptr;
typedef __attribute__((address_space(3))) short4v* lds_s4_ptr;

__device__ __forceinline__ void dma_piece(u32x4 rs4, unsigned lds_addr, unsigned voff) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds"
                 :: "v"(voff), "s"(rs4), "s"(lds_addr) : "memory", "m0");
}

__device__ __forceinline__ bf16x8 tr_pair(const unsigned char* p) {
    const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(p));
    const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(p + 1024));     // token rows + 4: the next piece
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

struct WwProb {
    const void* dy; const void* x; float* dw; float* dbias;
    int N, K;                 // dw [N][K]; dy [M][N]; x [M][K]
    int tiles_n, tiles_k;
    int tile0, ntiles;        // the tiles [tile0, tile0 + ntiles) of the layer (n tile major) belong to this launch
    int item0;                // first item of this layer in the launch's item order
};

struct WwArgs {
    WwProb p[WGRAD_WIDE_MAX];
    int nprob, M, nk, splitk, nks, items;
    float* slab;              // splitk > 1: [items][TILE_F] partial tiles in register order
};

// The launch's table is read where it lies - in the kernel-argument segment, through scalar loads with a computed
// offset.  (Touching the by-value parameter itself with a run-time layer index made hipcc copy all of it to scratch
// memory and read it back through VECTOR loads: every cursor field became "divergent", 70 spilled VGPRs.)
typedef const __attribute__((address_space(4))) WwArgs* ww_args_ptr;

__device__ __forceinline__ WwProb pick_prob(ww_args_ptr gp, int pi) {
    WwProb r;
    r.dy = gp->p[pi].dy; r.x = gp->p[pi].x; r.dw = gp->p[pi].dw; r.dbias = gp->p[pi].dbias;
    r.N = gp->p[pi].N; r.K = gp->p[pi].K; r.tiles_n = gp->p[pi].tiles_n; r.tiles_k = gp->p[pi].tiles_k;
    r.tile0 = gp->p[pi].tile0; r.ntiles = gp->p[pi].ntiles;
    r.item0 = gp->p[pi].item0;
    return r;
}

struct WwHead { int nprob, M, nk, splitk, nks, items; float* slab; };
__device__ __forceinline__ WwHead load_head(ww_args_ptr gp) {
    return WwHead{gp->nprob, gp->M, gp->nk, gp->splitk, gp->nks, gp->items, gp->slab};
}

__global__ __launch_bounds__(512, 2) void wgrad_wide_kernel(WwArgs) {
    const ww_args_ptr gp = (ww_args_ptr)__builtin_amdgcn_kernarg_segment_ptr();
    const WwHead g = load_head(gp);
    extern __shared__ __attribute__((aligned(16))) unsigned char wsm[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int M = g.M, nk = g.nk;

    // ---- this workgroup's items: the workgroups of one XCD own a contiguous range ----
    const int G = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3, nb = (G - xcd + 7) >> 3;
    const int iq = g.items >> 3, ir = g.items & 7;
    const int istart = xcd < ir ? xcd * (iq + 1) : ir * (iq + 1) + (xcd - ir) * iq;
    const int icount = iq + (xcd < ir ? 1 : 0);
    const int nseq = slot < icount ? (icount - slot + nb - 1) / nb : 0;
    if (nseq == 0) return;                               // workgroup-uniform
    auto item_coords = [&](int seq, int& pi, int& tn, int& tk, int& z) {
        const int L = istart + slot + seq * nb;
        pi = 0;
#pragma unroll
        for (int q = 1; q < WGRAD_WIDE_MAX; ++q)
            if (q < g.nprob && L >= gp->p[q].item0) pi = q;
        const WwProb pr = pick_prob(gp, pi);
        const int l = L - pr.item0;
        z = l / pr.ntiles;
        const int t = pr.tile0 + (l - z * pr.ntiles);
        tn = t / pr.tiles_k; tk = t - tn * pr.tiles_k;
    };

    const unsigned lds0 = (unsigned)(unsigned long long)(lds_byte_ptr)wsm;

    // ---- DMA plan of this lane: token row 8 * wave + 4 * u + (lane >> 4) of a K step, LDS slot (lane & 15) of the row ----
    const int drow = lane >> 4;
    const int dslot = lane & 15;
    const int dkey = drow | ((wave & 1) << 2);                                   // f(row) of both pieces
    const int dchunk = ((((dslot >> 1) ^ dkey) << 1) | (dslot & 1));             // the 16-byte chunk of the row this lane fetches
    // dy row = [wave row 0: 64 columns of the region | wave row 1: ...]; x row = [wave column 0: 32 columns | ... 3]
#define WW_ACOL(R) (((dchunk >> 3) * 128) + (R) * 64 + ((dchunk & 7) * 8))
#define WW_BCOL(R) (((dchunk >> 2) * 64) + (R) * 32 + ((dchunk & 3) * 8))
#define WW_ALDS(R, U) (lds0 + (unsigned)(R) * REG_B + (unsigned)(wave * 2 + (U)) * 1024u)
#define WW_BLDS(R, U) (lds0 + OP_B + (unsigned)(R) * REG_B + (unsigned)(wave * 2 + (U)) * 1024u)

    // cursor P feeds region R of both operands; it runs ahead of the MFMAs through item boundaries
    unsigned c0_a0, c0_a1, c0_b0, c0_b1, c1_a0, c1_a1, c1_b0, c1_b1, c0_par = 0u, c1_par = 0u;
    unsigned c0_sa = 0u, c0_sb = 0u, c1_sa = 0u, c1_sb = 0u;                     // byte step of one K step (64 rows)
    u32x4 c0_ars, c0_brs, c1_ars, c1_brs;
    int c0_kt, c0_kend, c0_seq, c1_kt, c1_kend, c1_seq;
#define WW_SEEK(P, R, SEQ)                                                                        \
    {                                                                                             \
        P##_seq = (SEQ);                                                                          \
        if (P##_seq >= nseq) {                                                                    \
            P##_a0 = P##_a1 = P##_b0 = P##_b1 = OOB; P##_kt = 0; P##_kend = 0x7fffffff;           \
            P##_sa = P##_sb = 0u;                                                                 \
        } else {                                                                                  \
            int pi_, tn_, tk_, z_;                                                                \
            item_coords(P##_seq, pi_, tn_, tk_, z_);                                              \
            const WwProb pr_ = pick_prob(gp, pi_);                                                 \
            P##_kt = z_ * g.nks;                                                                  \
            P##_kend = min(nk, P##_kt + g.nks);                                                   \
            P##_ars = buffer_words(pr_.dy, (unsigned)M * (unsigned)pr_.N * 2u);                   \
            P##_brs = buffer_words(pr_.x, (unsigned)M * (unsigned)pr_.K * 2u);                    \
            P##_sa = (unsigned)pr_.N * (unsigned)(WBK * 2);                                       \
            P##_sb = (unsigned)pr_.K * (unsigned)(WBK * 2);                                       \
            const unsigned row_ = (unsigned)(P##_kt * WBK + wave * 8 + drow);                     \
            const int ca_ = tn_ * 256 + WW_ACOL(R), cb_ = tk_ * 256 + WW_BCOL(R);                 \
            P##_a0 = ca_ < pr_.N ? (row_ * (unsigned)pr_.N + (unsigned)ca_) * 2u : OOB;           \
            P##_a1 = ca_ < pr_.N ? ((row_ + 4u) * (unsigned)pr_.N + (unsigned)ca_) * 2u : OOB;    \
            P##_b0 = cb_ < pr_.K ? (row_ * (unsigned)pr_.K + (unsigned)cb_) * 2u : OOB;           \
            P##_b1 = cb_ < pr_.K ? ((row_ + 4u) * (unsigned)pr_.K + (unsigned)cb_) * 2u : OOB;    \
        }                                                                                         \
    }
    WW_SEEK(c0, 0, 0) WW_SEEK(c1, 1, 0)
#define WW_ADVANCE(P, R)                                                                          \
    {                                                                                             \
        P##_par ^= 1u;                                                                            \
        P##_a0 += P##_sa; P##_a1 += P##_sa; P##_b0 += P##_sb; P##_b1 += P##_sb;                   \
        if (++P##_kt == P##_kend) WW_SEEK(P, R, P##_seq + 1)                                      \
    }
#define WW_DMA_A(P, R)                                                                            \
    {                                                                                             \
        const unsigned bo_ = P##_par * (unsigned)BUF_B;                                           \
        dma_piece(P##_ars, WW_ALDS(R, 0) + bo_, P##_a0);                                          \
        dma_piece(P##_ars, WW_ALDS(R, 1) + bo_, P##_a1);                                          \
    }
#define WW_DMA_B(P, R)                                                                            \
    {                                                                                             \
        const unsigned bo_ = P##_par * (unsigned)BUF_B;                                           \
        dma_piece(P##_brs, WW_BLDS(R, 0) + bo_, P##_b0);                                          \
        dma_piece(P##_brs, WW_BLDS(R, 1) + bo_, P##_b1);                                          \
    }
#define WW_DMA_A0() WW_DMA_A(c0, 0)
#define WW_DMA_B0() WW_DMA_B(c0, 0)
#define WW_DMA_A1() WW_DMA_A(c1, 1)
#define WW_DMA_B1() WW_DMA_B(c1, 1)

    // ---- transposed fragment reads: lane (gi, tq, tp) addresses token row 8 gi + tq (+ 4: second read, + 32: second
    //      k slice) and columns 4 tp .. 4 tp + 3 of a 16-column tile; the hardware hands lane li of the group column li ----
    const int gi = lane >> 4, li = lane & 15, tq = li >> 2, tp = li & 3;
    const int rkey = tq | ((gi & 1) << 2);
    const unsigned lbase = (unsigned)(gi * 2048 + tq * 256 + tp * 8);
    unsigned la[4], lb[2];
#pragma unroll
    for (int il = 0; il < 4; ++il) la[il] = lbase + (unsigned)((((wr * 4 + il) ^ rkey)) << 5);
#pragma unroll
    for (int jl = 0; jl < 2; ++jl) lb[jl] = lbase + (unsigned)((((wc * 2 + jl) ^ rkey)) << 5);

    floatx4 acc[8][4];
    floatx4 accb[2];                                     // dbias: column sums of dy tiles 2 wc, 2 wc + 1
    bf16x8 af[4][2], ba[2][2], bb[2][2], ones;
    const floatx4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;
    bool do_bias = false;                                // workgroup-uniform, per item

#define WW_READ_A(I0)                                                                             \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                               \
        af[i][0] = tr_pair(ap_ + ((I0) / 4) * REG_B + la[i]);                                     \
        af[i][1] = tr_pair(ap_ + ((I0) / 4) * REG_B + 8192 + la[i]);                              \
    }
#define WW_READ_B(BF, J0)                                                                         \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                               \
        BF[j][0] = tr_pair(bp_ + ((J0) / 2) * REG_B + lb[j]);                                     \
        BF[j][1] = tr_pair(bp_ + ((J0) / 2) * REG_B + 8192 + lb[j]);                              \
    }
    // ZERO: first K step of an item - the ks = 0 MFMA starts from the inline constant 0 (no clearing pass)
#define WW_MMA(I0, J0, BF, ZERO)                                                                  \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                              \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                             \
            _Pragma("unroll") for (int j = 0; j < 2; ++j)                                         \
                acc[(I0) + i][(J0) + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(BF[j][ks], af[i][ks], \
                    ((ZERO) && ks == 0) ? zero4 : acc[(I0) + i][(J0) + j], 0, 0, 0);
    // dbias of the dy tiles in af[U0], af[U0 + 1] (waves WCSEL of both wave rows)
#define WW_BIAS(WCSEL, U0, ZERO)                                                                  \
    if (do_bias && wc == (WCSEL)) {                                                               \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                          \
            _Pragma("unroll") for (int u = 0; u < 2; ++u)                                         \
                accb[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, af[(U0) + u][ks],         \
                    ((ZERO) && ks == 0) ? zero4 : accb[u], 0, 0, 0);                              \
    }
#define WW_WAIT_8() asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    // first K step of an item: the previous item's flush may stay in flight (8 stores of the direct form's last batch,
    // 32 slab stores; the dbias atomics are older than either)
#define WW_WAIT_SEL()                                                                             \
    if (vmsel == 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                              \
    else if (vmsel == 1) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");                        \
    else asm volatile("s_waitcnt vmcnt(40)" ::: "memory");
#define WW_TO_MMA(WAIT)                                                                           \
    WAIT()                                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                            \
    __builtin_amdgcn_s_barrier();                                                                 \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                            \
    __builtin_amdgcn_sched_barrier(0);                                                            \
    __builtin_amdgcn_s_setprio(1);
#define WW_TO_LOAD()                                                                              \
    __builtin_amdgcn_s_setprio(0);                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                            \
    __builtin_amdgcn_s_barrier();                                                                 \
    __builtin_amdgcn_sched_barrier(0);
#define WW_KSTEP(ZERO, VM)                                                                        \
    {                                                                                             \
        const unsigned char* const ap_ = wsm + rpar * BUF_B;                                      \
        const unsigned char* const bp_ = ap_ + OP_B;                                              \
        WW_READ_A(0) WW_READ_B(ba, 0)                                                             \
        WW_DMA_B1()                                                                               \
        WW_TO_MMA(VM)                                                                             \
        WW_MMA(0, 0, ba, ZERO)                                                                    \
        WW_BIAS(0, 0, ZERO) WW_BIAS(1, 2, ZERO)                                                   \
        WW_TO_LOAD()                                                                              \
        WW_READ_B(bb, 2)                                                                          \
        WW_DMA_A1()                                                                               \
        WW_TO_MMA(VM)                                                                             \
        WW_MMA(0, 2, bb, ZERO)                                                                    \
        WW_TO_LOAD()                                                                              \
        WW_READ_A(4)                                                                              \
        WW_DMA_A0()                                                                               \
        WW_TO_MMA(VM)                                                                             \
        WW_MMA(4, 2, bb, ZERO)                                                                    \
        WW_BIAS(2, 0, ZERO) WW_BIAS(3, 2, ZERO)                                                   \
        WW_TO_LOAD()                                                                              \
        WW_DMA_B0()                                                                               \
        WW_TO_MMA(VM)                                                                             \
        WW_MMA(4, 0, ba, ZERO)                                                                    \
        WW_TO_LOAD()                                                                              \
        rpar ^= 1u;                                                                               \
        WW_ADVANCE(c1, 1) WW_ADVANCE(c0, 0)                                                       \
    }

    // ---- prologue: all of the first K step, and the two regions of the second that phases 3-4 of a step issue ----
    WW_DMA_A0() WW_DMA_B0() WW_DMA_B1() WW_DMA_A1()
    WW_ADVANCE(c0, 0) WW_ADVANCE(c1, 1)
    WW_DMA_A0() WW_DMA_B0()
    WW_ADVANCE(c0, 0)
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");     // dy region 0 and x region 0 of step 0 have landed
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();           // waves 4-7 run one barrier behind

    unsigned rpar = 0u;
    int vmsel = 0;

    for (int seq = 0; seq < nseq; ++seq) {
        int pi, tn, tk, z;
        item_coords(seq, pi, tn, tk, z);
        const WwProb pr = pick_prob(gp, pi);
        const int k0 = z * g.nks, steps = min(nk, k0 + g.nks) - k0;
        do_bias = pr.dbias != nullptr && tk == 0;
        WW_KSTEP(true, WW_WAIT_SEL)
        for (int s = 1; s < steps; ++s) WW_KSTEP(false, WW_WAIT_8)
        vmsel = g.splitk == 1 ? 1 : 2;

        // both wave groups flush in the SAME barrier interval (see gemm_wide.hip)
        if (wr == 0) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const int n0 = tn * 256 + wr * 128, kk0 = tk * 256 + wc * 64;
        if (do_bias && gi == 0) {                        // rows of D are all equal: lanes 0-15 hold column li
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int n = n0 + (wc * 2 + u) * 16 + li;
                if (n < pr.N) atomicAdd(pr.dbias + n, accb[u][0]);
            }
        }
        if (g.splitk > 1) {
            // ---- partial tile in register order: store (i, j) of the workgroup is 512 lanes x 16 B = 8 KiB contiguous ----
            const long L = (long)(istart + slot + seq * nb);
            float* const slab = g.slab + L * TILE_F;
            const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(slab, 0, (unsigned)TILE_F * 4u, 0x00020000);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[i][j]), prs,
                                                           (unsigned)(((i * 4 + j) * 512 + tid) * 16), 0, 0);
        } else {
            // ---- the only writer of this tile: dw += acc with 16-byte loads and stores, 8 at a time ----
            const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc(pr.dw, 0, (unsigned)pr.N * (unsigned)pr.K * 4u, 0x00020000);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = kk0 + j * 16 + gi * 4;
                unsigned off[8];
                u32x4 old[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int n = n0 + i * 16 + li;
                    off[i] = (n < pr.N && k < pr.K) ? ((unsigned)n * (unsigned)pr.K + (unsigned)k) * 4u : 0xFFFFFFF0u;
                    old[i] = __builtin_amdgcn_raw_buffer_load_b128(drs, off[i], 0, 0);
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const floatx4 o = __builtin_bit_cast(floatx4, old[i]);
                    const floatx4 v = {o[0] + acc[i][j][0], o[1] + acc[i][j][1], o[2] + acc[i][j][2], o[3] + acc[i][j][3]};
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), drs, off[i], 0, 0);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (wr == 1) __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the run-ahead pieces (zeros) must land before the LDS is released
    if (wr == 0) __builtin_amdgcn_s_barrier();           // waves 0-3 catch up: barrier counts are equal again
    __syncthreads();

#undef WW_ACOL
#undef WW_BCOL
#undef WW_ALDS
#undef WW_BLDS
#undef WW_SEEK
#undef WW_ADVANCE
#undef WW_DMA_A
#undef WW_DMA_B
#undef WW_DMA_A0
#undef WW_DMA_B0
#undef WW_DMA_A1
#undef WW_DMA_B1
#undef WW_READ_A
#undef WW_READ_B
#undef WW_MMA
#undef WW_BIAS
#undef WW_WAIT_8
#undef WW_WAIT_SEL
#undef WW_TO_MMA
#undef WW_TO_LOAD
#undef WW_KSTEP
}

// dw += sum_z slab[item(layer, z, tile)]: one block = 1 KiB-wide strips of one tile (256 threads x 16 B), fixed order.
__global__ __launch_bounds__(256) void wgrad_wide_fold_kernel(WwArgs, int total_tiles) {
    const ww_args_ptr gp = (ww_args_ptr)__builtin_amdgcn_kernarg_segment_ptr();
    const WwHead g = load_head(gp);
    const int T = blockIdx.x >> 6, part = blockIdx.x & 63;
    if (T >= total_tiles) return;
    int pi = 0, tbase = 0;
    {
        int acc_t = 0;
#pragma unroll
        for (int q = 0; q < WGRAD_WIDE_MAX; ++q) {
            if (q < g.nprob) {
                const int t = gp->p[q].ntiles;
                if (T >= acc_t) { pi = q; tbase = acc_t; }
                acc_t += t;
            }
        }
    }
    const WwProb pr = pick_prob(gp, pi);
    const int tiles = pr.ntiles, tl = T - tbase, t = pr.tile0 + tl;
    const int tn = t / pr.tiles_k, tk = t - tn * pr.tiles_k;
    const int e = part * 256 + threadIdx.x;              // 16-byte element of the tile in register order
    const int ij = e >> 9, tid = e & 511;
    const int i = ij >> 2, j = ij & 3;
    const int wave = tid >> 6, lane = tid & 63, wr = wave >> 2, wc = wave & 3, gi = lane >> 4, li = lane & 15;
    const int n = tn * 256 + wr * 128 + i * 16 + li, k = tk * 256 + wc * 64 + j * 16 + gi * 4;
    const float* src = g.slab + ((long)pr.item0 + tl) * TILE_F + (long)e * 4;
    float4 s = *(const float4*)src;
    for (int z = 1; z < g.splitk; ++z) {
        const float4 v = *(const float4*)(src + (long)z * tiles * TILE_F);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    if (n < pr.N && k < pr.K) {
        float4* d = (float4*)(pr.dw + (long)n * pr.K + k);
        float4 o = *d;
        o.x += s.x; o.y += s.y; o.z += s.z; o.w += s.w;
        *d = o;
    }
}

int g_ww_splitk = -1;            // -1: not yet read; 0 = choose per launch; n > 0 = force

// Token split of a launch: cost model in microseconds (1.4 us per K step and workgroup, a direct flush ~10 us, a slab
// flush ~3 us, the fold pass reads every slab tile and reads + writes dw at ~4 TB/s); 1 = no split.
int choose_split(long tiles, int nk, double dw_bytes) {
    if (g_ww_splitk < 0) {
        const char* e = getenv("DPA_WGRAD_WIDE_SPLITK");
        g_ww_splitk = e ? atoi(e) : 0;
        if (g_ww_splitk < 0) g_ww_splitk = 0;
    }
    int lo = 1, hi = 16;
    if (g_ww_splitk > 0) lo = hi = g_ww_splitk;
    double best = 1e30;
    int best_s = 1;
    for (int s = lo; s <= hi; ++s) {
        const int nks = (nk + s - 1) / s;
        if (s > 1 && ((long)(s - 1) * nks >= nk || nks < 8)) continue;
        const long items = tiles * s, rounds = (items + 255) / 256;
        double t = rounds * (nks * 1.4 + (s > 1 ? 3.0 : 10.0));
        if (s > 1) t += ((double)items * TILE_F * 4.0 + 2.0 * dw_bytes) / 4.0e6 + 5.0;
        if ((double)items * TILE_F * 4.0 > 2.0e9) continue;
        if (t < best * 0.97) { best = t; best_s = s; }
    }
    return best_s;
}

}  // namespace

void set_wgrad_wide_splitk(int n) { g_ww_splitk = n < 0 ? 0 : n; }

// One layer's problem as this kernel can take it: a Linear-shaped weight gradient (1x1, stride 1, no padding) with
// 16-byte-aligned rows and operands addressable by 31-bit byte offsets.
bool wgrad_wide_supported(const WgradArgs& a) {
    if (!(a.R == 1 && a.S == 1 && a.pad == 0 && a.stride == 1 && a.asym == 0)) return false;
    if (a.K % 8 != 0 || a.C % 8 != 0) return false;
    // 16-byte accesses everywhere (DMA chunks, float4 read-modify-write of dw); a shape-only query passes null pointers
    if (((uintptr_t)a.x | (uintptr_t)a.dy | (uintptr_t)a.dw) % 16 != 0) return false;
    const long M = (long)a.N * a.Ho * a.Wo;
    const long wide = a.K > a.C ? a.K : a.C;
    if ((M + 2 * WBK) * wide * 2 >= (1l << 31)) return false;
    if ((long)a.K * a.C * 4 >= (1l << 32) - 16) return false;
    return true;
}

// ... and as it PAYS: both dimensions of dw at least one wide tile's worth (the small filters of the ResNets stay on
// conv_wgrad.hip) and enough tokens for a main loop
bool wgrad_wide_eligible(const WgradArgs& a) {
    static const int on = [] { const char* e = getenv("DPA_WGRAD_WIDE"); return e ? atoi(e) : 1; }();
    if (!on || deterministic() || !wgrad_wide_supported(a)) return false;
    const long M = (long)a.N * a.Ho * a.Wo;
    return a.K >= 192 && a.C >= 192 && M >= 2048;
}

long wgrad_wide_tiles(const WgradArgs& a) { return (long)((a.K + 255) / 256) * ((a.C + 255) / 256); }

// A launch of its own pays when a token split of at most 8 fills most of the chip: GPT-2's qkv / fc / fc2 (27-36 tiles) and
// the LM head (591 tiles; 1.19 vs 1.51 ms at 16,384 tokens), not its 768 x 768 projection (9 tiles - that one gains from a
// shared launch only: ops/wgroup.py) and not the ResNets' filters (4-16 tiles: 62 vs 61 us at best).
bool wgrad_wide_alone_pays(const WgradArgs& a) {
    if (!wgrad_wide_eligible(a)) return false;
    const long M = (long)a.N * a.Ho * a.Wo, nk = (M + WBK - 1) / WBK;
    const long s = nk / 8 < 8 ? nk / 8 : 8;
    return wgrad_wide_tiles(a) * (s < 1 ? 1 : s) >= 192;
}

namespace {

// one launch over the tile ranges in g.p[0 .. nprob) (tile0 / ntiles filled in), token split `splitk` (0 = choose)
void launch_ranges(WwArgs& g, double dw_bytes, int splitk, hipStream_t st) {
    long tiles = 0;
    for (int q = 0; q < g.nprob; ++q) tiles += g.p[q].ntiles;
    g.splitk = splitk > 0 ? splitk : choose_split(tiles, g.nk, dw_bytes);
    g.nks = (g.nk + g.splitk - 1) / g.splitk;
    int item0 = 0;
    for (int q = 0; q < g.nprob; ++q) {
        g.p[q].item0 = item0;
        item0 += g.p[q].ntiles * g.splitk;
    }
    g.items = item0;
    g.slab = g.splitk > 1 ? (float*)launcher_workspace((size_t)g.items * TILE_F * 4, st) : nullptr;
    const int grid = g.items < 256 ? g.items : 256;
    hipLaunchKernelGGL(wgrad_wide_kernel, dim3(grid), dim3(512), WW_LDS, st, g);
    if (g.splitk > 1) {
        hipLaunchKernelGGL(wgrad_wide_fold_kernel, dim3((unsigned)(tiles * 64)), dim3(256), 0, st, g, (int)tiles);
        launcher_workspace_release(g.slab);
    }
}

}  // namespace

void wgrad_wide_group(const WgradArgs* probs, int nprob, hipStream_t st) {
    if (nprob < 1 || nprob > WGRAD_WIDE_MAX) {
        fprintf(stderr, "wgrad_wide_group: %d layers in one launch (1..%d)\n", nprob, WGRAD_WIDE_MAX);
        abort();
    }
    static bool allowed = false;
    if (!allowed) {
        const hipError_t err = hipFuncSetAttribute((const void*)wgrad_wide_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, WW_LDS);
        if (err != hipSuccess) {
            fprintf(stderr, "wgrad_wide: cannot allow %d bytes of LDS: %s\n", WW_LDS, hipGetErrorString(err));
            abort();
        }
        allowed = true;
    }
    WwArgs g{};
    const long M = (long)probs[0].N * probs[0].Ho * probs[0].Wo;
    long tiles = 0;
    double dw_bytes = 0.0;
    for (int q = 0; q < nprob; ++q) {
        const WgradArgs& a = probs[q];
        if (!wgrad_wide_supported(a) || (long)a.N * a.Ho * a.Wo != M) {
            fprintf(stderr, "wgrad_wide_group: layer %d is not a Linear-shaped problem over the launch's %ld tokens (host dispatch error)\n", q, M);
            abort();
        }
        WwProb& p = g.p[q];
        p.dy = a.dy; p.x = a.x; p.dw = a.dw; p.dbias = a.dbias;
        p.N = a.K; p.K = a.C;                        // WgradArgs: K = out channels (rows of dw), C = in channels
        p.tiles_n = (a.K + 255) / 256; p.tiles_k = (a.C + 255) / 256;
        p.tile0 = 0; p.ntiles = p.tiles_n * p.tiles_k;
        tiles += p.ntiles;
        dw_bytes += (double)a.K * a.C * 4.0;
    }
    g.nprob = nprob; g.M = (int)M; g.nk = (int)((M + WBK - 1) / WBK);
    // More tiles than CUs and a ragged last round (the LM head: 591 tiles = 2.3 rounds): the whole rounds run UNSPLIT - each
    // tile added into dw by its one writer - and only the remaining tiles are split over the tokens, in a second launch
    // (591 tiles, 16,384 tokens: 512 x 256 K steps + 79 x 3 x 86 instead of 591 x 3 x 86 with 3 x the slab traffic).
    static const int hybrid = [] { const char* e = getenv("DPA_WGRAD_WIDE_HYBRID"); return e ? atoi(e) : 1; }();
    if (g_ww_splitk < 0) (void)choose_split(1, 8, 0.0);      // reads DPA_WGRAD_WIDE_SPLITK once
    const long full = (tiles / 256) * 256, rest = tiles - full;
    if (hybrid && g_ww_splitk <= 0 && full > 0 && rest > 0 && rest < 224) {
        WwArgs head = g, tail = g;
        long left = full;
        int nh = 0, nt = 0;
        for (int q = 0; q < nprob; ++q) {
            const int n = g.p[q].ntiles;
            const int take = (int)(left < n ? left : n);
            if (take > 0) { head.p[nh] = g.p[q]; head.p[nh].tile0 = 0; head.p[nh].ntiles = take; ++nh; }
            if (take < n) {
                tail.p[nt] = g.p[q]; tail.p[nt].tile0 = take; tail.p[nt].ntiles = n - take;
                // the bias sums of a layer are produced by the items of its first k tile: once, in whichever launch holds them
                ++nt;
            }
            left -= take;
        }
        head.nprob = nh; tail.nprob = nt;
        launch_ranges(head, dw_bytes, 1, st);
        launch_ranges(tail, dw_bytes * (double)rest / (double)tiles, 0, st);
        return;
    }
    launch_ranges(g, dw_bytes, 0, st);
}

}  // namespace dpa
