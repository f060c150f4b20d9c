// Implicit-GEMM convolution (fprop + dgrad) for gfx950: NHWC bf16 activations,
// KRSC bf16 filters, fp32 accumulation on v_mfma_f32_16x16x32_bf16.
//
// GEMM view (per output pixel p = (n,ho,wo) and output channel k):
//     y[p][k] = sum_{kk} im2col(x)[p][kk] * w[k][kk],   kk = (r,s,c), c fastest
// Both operands are K-contiguous, which is exactly the bf16 MFMA fragment shape
// (8 consecutive k per lane), so no transposes are needed.  The MFMA A operand is
// the FILTER tile (rows = output channels) and the B operand the gathered
// activation tile (cols = pixels): each lane then holds 4 consecutive output
// channels of one pixel, i.e. 8 contiguous bytes of the NHWC output row.
//
// Block = 4 waves (64-wide), each wave a 64(pixel) x 64(channel) tile as 4x4 MFMA
// tiles; block tile 128x128 (2x2 waves) or 256x64 (4x1 waves, for 64-channel
// layers).  K step 32, one barrier per K step.  Operand tiles go global -> LDS by DMA
// (buffer_load ... lds; descriptor range checks supply the zeros of padding and tile
// edges), double- or triple-buffered; LDS tiles are XOR-swizzled so the ds_read_b128
// fragment reads are bank-conflict free.  Register-staged variants are kept selectable
// (STG below; measurements in docs/DESIGN.md section 6).  Epilogue goes through LDS so that
// global stores are 16 B per lane over full NHWC rows.  Optional fused epilogue work:
//   * per-channel (sum, sum of squares) of the output for a following BatchNorm, added
//     with fp32 atomics into a small replicated buffer (removes one full read of the
//     activation tensor per layer);
//   * `addend`: y = conv(...) + addend, which folds the residual-branch gradient add of
//     a bottleneck block into the dgrad kernel.
//
// MODE 0 fprop; MODE 1 dgrad at stride 1; MODE 2 dgrad at stride > 1, decomposed by
// output-pixel parity class: a pixel (h, w) only receives taps r = (h + pad) mod stride
// (+ k*stride), so each class is a dense small convolution and no MFMA work is spent on
// structurally-zero taps (a masked stride-2 3x3 dgrad would waste 75 % of it; a strided
// 1x1 dgrad is one dense GEMM on a quarter of the pixels plus zero fill).
#include <stdlib.h>
#include "common.h"
#include "kernels.h"

namespace dpa {

namespace {

constexpr int BK = 32;          // K step (bf16 elements)
constexpr int CHUNKS = BK / 8;  // 16-byte chunks per tile row

typedef __attribute__((address_space(3))) short4v* lds_s4_ptr;

__device__ __forceinline__ int swz4(int row) {  // conflict-free chunk swizzle for 64-byte rows
    return (0x78 >> (2 * ((row >> 2) & 3))) & 3;  // {0,2,3,1}[(row>>2)&3]
}

// ONE: 1x1 filter, no padding (a plain GEMM over gathered rows): no tap arithmetic in the loop.
// STG selects how the operand tiles reach LDS (host: staging(); DPA_IGEMM_STG forces one).  Accumulators stay in
// VGPRs in all of them (no AGPR copies):
//   STG 0: one register set, 4 waves/SIMD;  STG 1: register ring, 3 waves/SIMD;
//   STG 2: LDS-DMA (buffer_load ... lds): the gather lands in LDS without passing through VGPRs or
//          ds_write_b128 (13 LDS-pipe cycles per KiB) - with register staging the LDS pipe, not the MFMA pipe,
//          is the busiest unit of this kernel (PMC: SQ_ACTIVE_INST_LDS 71 % on a 768-deep GEMM).  The lane ->
//          LDS slot map of the DMA is linear, so each lane fetches the chunk that belongs in its slot
//          (the XOR swizzle moves to the source address); padding / tile edges still read zeros by range check.
//          Default for large grids (4 workgroups resident per CU).
//   STG 3: the same DMA issued from inline assembly with counted s_waitcnt vmcnt(N): two tiles in flight over
//          three LDS buffers (3 workgroups per CU).  Default for small grids.  (4, 5: depth 4 / 6, not instantiated.)
constexpr int dma_depth(int stg) { return stg == 3 ? 2 : stg == 4 ? 4 : stg == 5 ? 6 : 1; }
constexpr int igemm_lds_bytes(int wp, int wc, int stg) {
    const int stage = (wp + wc) * 64 * BK * 2, epi = wp * 64 * (wc * 64 + 8) * 2;
    const int ring = (stg >= 3 ? dma_depth(stg) + 1 : 2) * stage;
    return ring > epi ? ring : epi;
}
constexpr int igemm_waves_per_simd(int wp, bool one, int stg) {     // __launch_bounds__ occupancy target
    return stg == 1 ? ((wp == 4 && !one) ? 2 : 3) : stg == 3 ? 3 : stg >= 4 ? 2 : 4;
}

template <int WP, int WC, int MODE, bool ONE, int STG>
__global__ __launch_bounds__(256, igemm_waves_per_simd(WP, ONE, STG)) void conv_igemm_kernel(ConvArgs a, int tiles_c, long M_all, int Kg_all) {
    constexpr int BP = WP * 64, BC = WC * 64;
    constexpr int XR = BP * CHUNKS / 256;  // x rows per thread
    constexpr int WR = BC * CHUNKS / 256;  // w rows per thread
    constexpr int CS = BC + 8;             // epilogue row stride (elements)
    constexpr int STAGE_BYTES = (BP + BC) * BK * 2;
    constexpr int EPI_BYTES = BP * CS * 2;
    constexpr int DEPTH = dma_depth(STG);  // tiles in flight ahead of the one being multiplied (counted DMA variants)
    constexpr int NSTAGE = STG >= 3 ? DEPTH + 1 : 2;
    constexpr int LDS_BYTES = igemm_lds_bytes(WP, WC, STG);
    static_assert(LDS_BYTES >= NSTAGE * STAGE_BYTES && LDS_BYTES >= EPI_BYTES, "LDS plan");
    constexpr bool DYN_LDS = LDS_BYTES > 65536;   // above the static limit the launch passes the size
    __shared__ __attribute__((aligned(16))) unsigned char smem_static[DYN_LDS ? 16 : LDS_BYTES];
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_dynamic[];
    unsigned char* const smem = DYN_LDS ? smem_dynamic : smem_static;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wp = wave / WC, wc = wave % WC;

    // ---- geometry of this launch / parity class ----
    // pixel grid of the GEMM rows: [N][PH][PW]; MODE 2 walks every `stride`-th output pixel
    int PH = a.Ho, PW = a.Wo, ph0 = 0, pw0 = 0;       // class origin in the output image
    int r0 = 0, s0 = 0, nr = a.R, ns = a.S, tstep = 1; // taps r = r0 + tstep * kr, kr < nr
    if (MODE == 2) {
        const int st = a.stride;
        ph0 = blockIdx.y / st; pw0 = blockIdx.y % st;
        PH = (a.Ho - ph0 + st - 1) / st; PW = (a.Wo - pw0 + st - 1) / st;
        r0 = (ph0 + a.pad) % st; s0 = (pw0 + a.pad) % st;
        nr = r0 < a.R ? (a.R - r0 + st - 1) / st : 0;
        ns = s0 < a.S ? (a.S - s0 + st - 1) / st : 0;
        tstep = st;
    }
    const long M = (MODE == 2) ? (long)a.N * PH * PW : M_all;
    const int Kg = (MODE == 2) ? nr * ns * a.C : Kg_all;

    // XCD-aware remap: blocks that share an XCD (bid % 8) get a contiguous range of
    // tiles, so the channel tiles of one pixel tile hit the same L2.  Bijective.
    int lid;
    {
        const int nwg = gridDim.x, bid = blockIdx.x;
        const int q = nwg >> 3, r8 = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        lid = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + idx;
    }
    const int tile_c = lid % tiles_c, tile_p = lid / tiles_c;
    const int stat_slot = tile_p * (int)gridDim.y + (int)blockIdx.y;   // replica row of the statistics epilogues (parity classes apart)
    const long m0 = (long)tile_p * BP;
    if (m0 >= M) return;                   // MODE 2: classes differ in size; block-uniform exit
    const int c0 = tile_c * BC;

    // Buffer descriptors (wave-uniform, straight from kernel arguments): 32-bit byte offsets, and a
    // lane whose tap falls into the padding / past the tile edge is given an offset past the end
    // of the buffer, for which the hardware returns zeros - the gather needs no branch, no select
    // on the loaded data and no 64-bit address arithmetic.
    const unsigned xbytes = (unsigned)a.N * a.H * a.W * a.C * 2u;
    const int KgW = a.R * a.S * a.C;       // row pitch of the filter matrix
    const unsigned wbytes = (unsigned)a.K * KgW * 2u;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, wbytes, 0x00020000);
    const unsigned long long xaddr = (unsigned long long)a.x, waddr = (unsigned long long)a.w;
    const u32x4 xrs4 = {(unsigned)xaddr, (unsigned)(xaddr >> 32), xbytes, 0x00020000u};   // the same descriptors as plain words
    const u32x4 wrs4 = {(unsigned)waddr, (unsigned)(waddr >> 32), wbytes, 0x00020000u};

    // ---- per-thread gather state ----
    const int chunk = tid & 3;
    const int lrow = tid >> 2;  // 0..63
    int hi0[XR], wi0[XR];
    unsigned xb[XR];            // ONE: byte offset of the row's pixel (xbytes when the row is past M); else pixel index of its image
    unsigned wb[WR];            // byte offset of the filter row (rows past K read zeros)
    const int PHW = PH * PW;
#pragma unroll
    for (int i = 0; i < XR; ++i) {
        const long m = m0 + lrow + i * 64;
        const bool rv = m < M;
        if (ONE && a.stride == 1) {              // GEMM row m is input pixel m: no index arithmetic at all
            hi0[i] = 0; wi0[i] = 0;
            xb[i] = rv ? (unsigned)m * (unsigned)a.C * 2u : xbytes;
            continue;
        }
        const unsigned mm = rv ? (unsigned)m : 0u;      // M < 2^31 (host check): 32-bit divisions
        const unsigned n = mm / (unsigned)PHW;
        const unsigned rem = mm - n * (unsigned)PHW;
        const int ho = (int)(rem / (unsigned)PW), wo = (int)(rem - (rem / (unsigned)PW) * (unsigned)PW);
        if (MODE == 0)      { hi0[i] = ho * a.stride - a.pad; wi0[i] = a.asym ? wo * a.stride_w - a.pad_w : wo * a.stride - a.pad; }
        else if (MODE == 1) { hi0[i] = ho + a.pad;            wi0[i] = wo + a.pad; }
        else                { hi0[i] = (ho * a.stride + ph0 + a.pad - r0) / a.stride;   // exact division
                              wi0[i] = (wo * a.stride + pw0 + a.pad - s0) / a.stride; }
        if (ONE) xb[i] = rv ? (unsigned)(((int)n * a.H + hi0[i]) * a.W + wi0[i]) * (unsigned)a.C * 2u : xbytes;
        else {
            xb[i] = n * (unsigned)(a.H * a.W);   // pixel index of the image origin
            if (!rv) hi0[i] = -(1 << 24);        // never inside the image: the row gathers zeros
        }
    }
#pragma unroll
    for (int i = 0; i < WR; ++i) {
        const int co = c0 + lrow + i * 64;
        wb[i] = co < a.K ? (unsigned)co * (unsigned)KgW * 2u : wbytes;
    }
    const int wsw = chunk ^ swz4(lrow);  // (row>>2)&3 is the same for lrow + 64*i
    constexpr bool RING = STG == 1;
    // register staging: load chunk `chunk`, store it to slot `wsw`;  DMA: slot = lane order = `chunk`, so load chunk `wsw`
    int kr = 0, ks = 0, kc = (STG >= 2 ? wsw : chunk) * 8;  // (tap row, tap col, channel) of this thread's chunk
    if (!ONE) while (kc >= a.C) { kc -= a.C; if (++ks == ns) { ks = 0; ++kr; } }

    // Two register sets form a ring: the loads of tile t+2 are issued before the MFMAs of tile
    // t, and tile t+1 (issued one whole iteration earlier) is written to LDS after them, so a
    // load has two iterations of MFMA work (times the resident blocks) to cover its latency.
    u32x4 xr0[XR], wr0[WR], xr1[XR], wr1[WR];

#define DPA_LOAD_TILE(XREG, WREG)                                                                \
    {                                                                                            \
        if (ONE) {                                                                               \
            const unsigned kadd = kc < Kg ? (unsigned)kc * 2u : xbytes;                          \
            _Pragma("unroll") for (int i = 0; i < XR; ++i)                                       \
                XREG[i] = __builtin_amdgcn_raw_buffer_load_b128(xrs, xb[i] + kadd, 0, 0);        \
            _Pragma("unroll") for (int i = 0; i < WR; ++i)                                       \
                WREG[i] = __builtin_amdgcn_raw_buffer_load_b128(wrs, wb[i] + (unsigned)kc * 2u, 0, 0); \
            kc += BK;                                                                            \
        } else {                                                                                 \
            const bool kok = kr < nr;                                                            \
            _Pragma("unroll") for (int i = 0; i < XR; ++i) {                                     \
                int hi, wi;                                                                      \
                if (MODE == 0) { hi = hi0[i] + kr; wi = wi0[i] + ks; }                           \
                else           { hi = hi0[i] - kr; wi = wi0[i] - ks; }                           \
                const bool ok = kok && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W; \
                const unsigned off = ((xb[i] + (unsigned)(hi * a.W + wi)) * (unsigned)a.C + (unsigned)kc) * 2u; \
                XREG[i] = __builtin_amdgcn_raw_buffer_load_b128(xrs, ok ? off : xbytes, 0, 0);      \
            }                                                                                    \
            const unsigned wk = (unsigned)(((r0 + tstep * kr) * a.S + (s0 + tstep * ks)) * a.C + kc) * 2u; \
            _Pragma("unroll") for (int i = 0; i < WR; ++i)                                       \
                WREG[i] = __builtin_amdgcn_raw_buffer_load_b128(wrs, kok ? wb[i] + wk : wbytes, 0, 0); \
            kc += BK;                                                                            \
            while (kc >= a.C) { kc -= a.C; if (++ks == ns) { ks = 0; ++kr; } }                   \
        }                                                                                        \
    }

#define DPA_STORE_TILE(BUF, XREG, WREG)                                                          \
    {                                                                                            \
        unsigned char* xs_ = smem + (BUF) * STAGE_BYTES;                                         \
        unsigned char* ws_ = xs_ + BP * BK * 2;                                                  \
        _Pragma("unroll") for (int i = 0; i < XR; ++i)                                           \
            *(u32x4*)(xs_ + ((lrow + i * 64) * CHUNKS + wsw) * 16) = XREG[i];                    \
        _Pragma("unroll") for (int i = 0; i < WR; ++i)                                           \
            *(u32x4*)(ws_ + ((lrow + i * 64) * CHUNKS + wsw) * 16) = WREG[i];                    \
    }

    // LDS-DMA of one K tile: wave w fills rows 16w .. 16w+15 (+ 64 i) of each operand, 1 KiB per instruction
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
#define DPA_DMA_TILE(BUF)                                                                        \
    {                                                                                            \
        unsigned char* xs_ = smem + (BUF) * STAGE_BYTES + wave_u * (16 * BK * 2);                \
        unsigned char* ws_ = xs_ + BP * BK * 2;                                                  \
        if (ONE) {                                                                               \
            const unsigned kadd = kc < Kg ? (unsigned)kc * 2u : xbytes;                          \
            _Pragma("unroll") for (int i = 0; i < XR; ++i)                                       \
                dma16<(STG >= 3)>(xrs, xrs4, xs_ + i * (64 * BK * 2), xb[i] + kadd); \
            _Pragma("unroll") for (int i = 0; i < WR; ++i)                                       \
                dma16<(STG >= 3)>(wrs, wrs4, ws_ + i * (64 * BK * 2), wb[i] + (unsigned)kc * 2u); \
            kc += BK;                                                                            \
        } else {                                                                                 \
            const bool kok = kr < nr;                                                            \
            _Pragma("unroll") for (int i = 0; i < XR; ++i) {                                     \
                int hi, wi;                                                                      \
                if (MODE == 0) { hi = hi0[i] + kr; wi = wi0[i] + ks; }                           \
                else           { hi = hi0[i] - kr; wi = wi0[i] - ks; }                           \
                const bool ok = kok && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W; \
                const unsigned off = ((xb[i] + (unsigned)(hi * a.W + wi)) * (unsigned)a.C + (unsigned)kc) * 2u; \
                dma16<(STG >= 3)>(xrs, xrs4, xs_ + i * (64 * BK * 2), ok ? off : xbytes); \
            }                                                                                    \
            const unsigned wk = (unsigned)(((r0 + tstep * kr) * a.S + (s0 + tstep * ks)) * a.C + kc) * 2u; \
            _Pragma("unroll") for (int i = 0; i < WR; ++i)                                       \
                dma16<(STG >= 3)>(wrs, wrs4, ws_ + i * (64 * BK * 2), kok ? wb[i] + wk : wbytes); \
            kc += BK;                                                                            \
            while (kc >= a.C) { kc -= a.C; if (++ks == ns) { ks = 0; ++kr; } }                   \
        }                                                                                        \
    }

    floatx4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

    const int nk = (Kg + BK - 1) / BK;
    const int frow = lane & 15;
    const int fsw = (lane >> 4) ^ swz4(frow);

#define DPA_COMPUTE(BUF)                                                                         \
    {                                                                                            \
        const unsigned char* xs = smem + (BUF) * STAGE_BYTES;                                    \
        const unsigned char* ws = xs + BP * BK * 2;                                              \
        bf16x8 af[4], bfr[4];                                                                    \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                            \
            af[i] = *(const bf16x8*)(ws + ((wc * 64 + i * 16 + frow) * CHUNKS + fsw) * 16);      \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                            \
            bfr[j] = *(const bf16x8*)(xs + ((wp * 64 + j * 16 + frow) * CHUNKS + fsw) * 16);     \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                            \
            _Pragma("unroll") for (int j = 0; j < 4; ++j)                                        \
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0); \
    }

    // Loads past the last K tile fall outside the buffers (zeros) and their LDS image is never
    // read, so the loop body carries no conditions.
    if (STG >= 3) {
        // DMA `DEPTH` tiles ahead over DEPTH + 1 LDS buffers: when tile kt is multiplied, tiles kt+1 .. kt+DEPTH-1
        // have had whole iterations to land and tile kt+DEPTH is being issued; the wait before the barrier
        // leaves the DEPTH-1 newest tiles' loads outstanding (vmcnt counts DMA loads in issue order).  The
        // depth is picked by the host from the number of workgroups per CU: few resident workgroups cannot hide
        // the ~1 us load latency by occupancy, but then have the LDS for a deep ring.
        constexpr int LPT = XR + WR;                                       // loads per tile per lane
        constexpr int OUTSTANDING = (DEPTH - 1) * LPT;
        static_assert(OUTSTANDING < 64, "vmcnt is a 6-bit counter");
        constexpr int WAIT_OLDEST_TILE = (OUTSTANDING & 15) | ((OUTSTANDING >> 4) << 14) | (7 << 4) | (15 << 8);   // vmcnt only
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) DPA_DMA_TILE(d)
        __builtin_amdgcn_s_waitcnt(WAIT_OLDEST_TILE);
        __syncthreads();
        int cur = 0, nxt = DEPTH;
        for (int kt = 0; kt < nk; ++kt) {
            DPA_DMA_TILE(nxt)                  // tile kt + DEPTH (its buffer was last read in iteration kt - 1)
            __builtin_amdgcn_sched_barrier(0);
            DPA_COMPUTE(cur)
            __builtin_amdgcn_s_waitcnt(WAIT_OLDEST_TILE);
            __syncthreads();
            cur = cur == DEPTH ? 0 : cur + 1;
            nxt = nxt == DEPTH ? 0 : nxt + 1;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the tail tiles' (zero) DMA must land before LDS is reused
        __syncthreads();
    } else if (STG == 2) {
        // tile kt+1 is DMA-ed into the other buffer while tile kt is multiplied; the DMA counts on vmcnt
        DPA_DMA_TILE(0)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            DPA_DMA_TILE((kt + 1) & 1)         // tile kt + 1 (its buffer was last read before the barrier)
            __builtin_amdgcn_sched_barrier(0);
            DPA_COMPUTE(kt & 1)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    } else if (RING) {
        DPA_LOAD_TILE(xr0, wr0)
        DPA_LOAD_TILE(xr1, wr1)
        DPA_STORE_TILE(0, xr0, wr0)
        __syncthreads();
        for (int kt = 0; kt + 1 < nk; kt += 2) {
            DPA_LOAD_TILE(xr0, wr0)            // tile kt + 2
            __builtin_amdgcn_sched_barrier(0); // keep the loads ahead of the MFMAs and of the older tile's LDS writes
            DPA_COMPUTE(0)                     // tile kt
            DPA_STORE_TILE(1, xr1, wr1)        // tile kt + 1
            __syncthreads();
            DPA_LOAD_TILE(xr1, wr1)            // tile kt + 3
            __builtin_amdgcn_sched_barrier(0);
            DPA_COMPUTE(1)                     // tile kt + 1
            DPA_STORE_TILE(0, xr0, wr0)        // tile kt + 2
            __syncthreads();
        }
        if (nk & 1) {
            DPA_COMPUTE(0)
            __syncthreads();
        }
    } else {
        DPA_LOAD_TILE(xr0, wr0)
        DPA_STORE_TILE(0, xr0, wr0)
        DPA_LOAD_TILE(xr0, wr0)                // tile 1 in flight across the barrier
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            DPA_STORE_TILE((kt + 1) & 1, xr0, wr0)   // tile kt + 1 (its buffer was last read before the barrier)
            DPA_LOAD_TILE(xr0, wr0)                   // tile kt + 2
            __builtin_amdgcn_sched_barrier(0);
            DPA_COMPUTE(kt & 1)
            __syncthreads();
        }
    }

    // ---- epilogue: bias/activation, stage the bf16 tile in LDS ----
    unsigned short* cs = (unsigned short*)smem;
    const int q4 = (lane >> 4) * 4;
    if (a.bias == nullptr && a.act == 0) {          // every BatchNorm-ed conv and every dgrad: pack and go
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                uint2 pk;
                pk.x = pack_bf16x2(acc[i][j][0], acc[i][j][1]);
                pk.y = pack_bf16x2(acc[i][j][2], acc[i][j][3]);
                *(uint2*)(cs + (wp * 64 + j * 16 + frow) * CS + wc * 64 + i * 16 + q4) = pk;
            }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int col = wc * 64 + i * 16 + q4;  // channel inside the block tile (4 consecutive)
            float b[4] = {0.f, 0.f, 0.f, 0.f};
            if (a.bias != nullptr) {
#pragma unroll
                for (int r = 0; r < 4; ++r) b[r] = a.bias[min(c0 + col + r, a.K - 1)];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int pix = wp * 64 + j * 16 + frow;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float t = acc[i][j][r] + b[r];
                    if (a.act == 1) t = fmaxf(t, 0.f);
                    else if (a.act == 2) t = gelu_tanh(t);
                    v[r] = t;
                }
                uint2 pk;
                pk.x = pack_bf16x2(v[0], v[1]);
                pk.y = pack_bf16x2(v[2], v[3]);
                *(uint2*)(cs + pix * CS + col) = pk;
            }
        }
    }
    // ---- write-out geometry, and an early issue of everything the write-out READS from memory
    //      (residual / gradient operand and its bit mask).  Issued here - the accumulators are in
    //      LDS, their registers free - the loads fly across the barrier and the statistics; issued
    //      inside the write-out loop, each of the 8 passes paid a full memory round trip (measured:
    //      the fused epilogues streamed at < 4 TB/s where the stand-alone BatchNorm kernels reach 5). ----
    constexpr int CPR = BC / 8;        // chunks per row
    constexpr int RPP = 256 / CPR;     // rows per pass
    constexpr int NPASS = BP / RPP;
    const int och = tid % CPR, orow = tid / CPR;
    const int co = c0 + och * 8;       // this thread's 8 channels are the same in every pass
    unsigned eoff[NPASS];              // element offset of the 8 channels in the output; ~0u = nothing to write
#pragma unroll
    for (int pss = 0; pss < NPASS; ++pss) {
        const long m = m0 + pss * RPP + orow;
        unsigned pix = (unsigned)m;
        if (MODE == 2) {
            const unsigned mm = m < M ? (unsigned)m : 0u;
            const unsigned n = mm / (unsigned)PHW;
            const unsigned rem = mm - n * (unsigned)PHW;
            const unsigned hh = rem / (unsigned)PW, ww = rem - hh * (unsigned)PW;
            pix = (n * (unsigned)a.Ho + hh * (unsigned)a.stride + (unsigned)ph0) * (unsigned)a.Wo + ww * (unsigned)a.stride + (unsigned)pw0;
        }
        eoff[pss] = (m < M && co < a.K) ? pix * (unsigned)a.K + (unsigned)co : ~0u;   // output < 2^31 elements (host check)
    }
    const bool epi_on = MODE == 0 && a.epi != 0;
    const bool epi_dgrad_stats = MODE != 0 && a.epi == 4;       // dgrad + reduction pass of the next BatchNorm backward
    const bool epi_gelu_bwd = MODE == 1 && a.epi == 5;          // dgrad through a GELU: y = tile * gelu'(ep_src)
    const void* pre_src = (epi_on || epi_gelu_bwd) ? a.ep_src : a.addend;
    const unsigned char* pre_msk = epi_on ? (a.epi == 1 ? nullptr : a.ep_mask_in) : a.addend_mask;
    u32x4 pre[NPASS];
    unsigned pmk[NPASS];
    if (pre_src != nullptr) {
        const unsigned ybytes = (unsigned)a.N * a.Ho * a.Wo * a.K * 2u;
        const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc((void*)pre_src, 0, ybytes, 0x00020000);
#pragma unroll
        for (int pss = 0; pss < NPASS; ++pss)
            pre[pss] = __builtin_amdgcn_raw_buffer_load_b128(srs, eoff[pss] != ~0u ? eoff[pss] * 2u : 0xFFFFFFF0u, 0, 0);
        if (pre_msk != nullptr) {
            const __amdgpu_buffer_rsrc_t mrs = __builtin_amdgcn_make_buffer_rsrc((void*)pre_msk, 0, ybytes >> 4, 0x00020000);
#pragma unroll
            for (int pss = 0; pss < NPASS; ++pss)
                pmk[pss] = __builtin_amdgcn_raw_buffer_load_b8(mrs, eoff[pss] != ~0u ? eoff[pss] >> 3 : 0xFFFFFFF0u, 0, 0);
        } else {
#pragma unroll
            for (int pss = 0; pss < NPASS; ++pss) pmk[pss] = 0xFFu;
        }
    } else {
#pragma unroll
        for (int pss = 0; pss < NPASS; ++pss) { pre[pss] = u32x4{0u, 0u, 0u, 0u}; pmk[pss] = 0u; }
    }

    __syncthreads();

    // ---- optional BatchNorm statistics of the bf16-rounded tile, on the matrix cores: for a block
    //      of 16 channels the transposed LDS read delivers F[c][p] (8 consecutive pixels per lane),
    //      which is at once the A and the B fragment of G = F F^T (diagonal = sum of squares), and
    //      F . ones gives the plain sums.  Two MFMAs per 16 channels x 32 pixels replace 1.5 k
    //      LDS/VALU instructions per wave; the 16 diagonal lanes add (sum, sumsq) into replica
    //      (tile_p mod stats_rows) of a zero-initialised [rows][2][K] buffer. ----
    if (a.stats != nullptr) {
        constexpr int CB_PER_WAVE = BC / 16 / 4;
        const int kg = lane >> 4, li = lane & 15, tq = li >> 2, tp = li & 3;
        bf16x8 ones;
#pragma unroll
        for (int j = 0; j < 8; ++j) ones[j] = (__bf16)1.0f;
        const unsigned char* csb = (const unsigned char*)cs;
#pragma unroll
        for (int u = 0; u < CB_PER_WAVE; ++u) {
            const int cb = wave * CB_PER_WAVE + u;
            floatx4 G = {0.f, 0.f, 0.f, 0.f}, Sm = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int pb = 0; pb < BP / 32; ++pb) {
                const int row = pb * 32 + 8 * kg + tq;
                const unsigned char* ptr = csb + row * (CS * 2) + (cb * 16 + 4 * tp) * 2;
                const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(ptr));
                const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(ptr + 4 * CS * 2));
                const bf16x8 f = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
                G = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f, f, G, 0, 0, 0);
                Sm = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f, ones, Sm, 0, 0, 0);
            }
            // C layout: column n = lane & 15, rows 4 * (lane >> 4) + r; the diagonal element of channel
            // n sits in the lane whose row group contains n, in register n & 3
            const int r = li & 3;
            const float q = r == 0 ? G[0] : r == 1 ? G[1] : r == 2 ? G[2] : G[3];
            const float sm = r == 0 ? Sm[0] : r == 1 ? Sm[1] : r == 2 ? Sm[2] : Sm[3];
            const int ch = c0 + cb * 16 + li;
            if ((li >> 2) == kg && ch < a.K) {
                float* dst = a.stats + (long)(stat_slot % a.stats_rows) * 2 * a.K + ch;
                atomicAdd(dst, sm);
                atomicAdd(dst + a.K, q);
            }
        }
    }

    // ---- coalesced write-out: 16 B per lane over full output rows ----
    unsigned short* yg = (unsigned short*)a.y;

    // Fused BatchNorm epilogues (a.epi != 0; MODE 0 only).  They let a cheap 1x1 conv in front of a
    // BatchNorm be RECOMPUTED instead of stored: the raw conv output never goes to memory.
    //   epi 1: y = relu(c * ep_a[ch] + ep_b[ch] + ep_src) and its 1-bit mask      (BN apply + residual + ReLU)
    //   epi 2: no output; adds sum(g), sum(g * c) per channel into ep_stats        (BN backward, reduction pass)
    //   epi 3: y = ep_a[ch] * g + ep_b[ch] * c + ep_c[ch]                          (BN backward, apply pass)
    // with c = the bf16-rounded conv output of this tile, g = ep_src under the bit mask ep_mask_in.
    // per-thread (sum, sum-product) partials of 8 channels -> ep_stats: fold the RPP row groups through LDS (the
    // staged tile is dead once everyone has read it), then one atomic per channel and statistic
    auto fold_stats = [&](const float (&s8)[8], const float (&q8)[8]) {
        __syncthreads();
        float* red = (float*)smem;                       // [2][RPP][BC]
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            red[(0 * RPP + orow) * BC + och * 8 + j] = s8[j];
            red[(1 * RPP + orow) * BC + och * 8 + j] = q8[j];
        }
        __syncthreads();
        for (int o = tid; o < 2 * BC; o += 256) {
            const int st_ = o / BC, ch = o - st_ * BC;
            float t = 0.f;
#pragma unroll 8
            for (int r = 0; r < RPP; ++r) t += red[(st_ * RPP + r) * BC + ch];
            if (c0 + ch < a.K)
                atomicAdd(a.ep_stats + ((long)(stat_slot % a.ep_stats_rows) * 2 + st_) * a.K + c0 + ch, t);
        }
    };
    if (epi_on) {
        float ea[8], eb[8], ec[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int ch = min(co + j, a.K - 1);
            ea[j] = a.epi != 2 ? a.ep_a[ch] : 0.f;
            eb[j] = a.epi != 2 ? a.ep_b[ch] : 0.f;
            ec[j] = a.epi == 3 ? a.ep_c[ch] : 0.f;
        }
        float s8[8], q8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { s8[j] = 0.f; q8[j] = 0.f; }
#pragma unroll
        for (int pss = 0; pss < NPASS; ++pss) {
            const int row = pss * RPP + orow;
            if (eoff[pss] != ~0u) {
                const size_t e0 = eoff[pss];
                float c[8], g[8];
                { const uint4 v = *(const uint4*)(cs + row * CS + och * 8); unpack8(v, c); }
                unpack8(__builtin_bit_cast(uint4, pre[pss]), g);
                if (a.epi == 1) {
                    unsigned mk = 0;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float t = fmaxf(c[j] * ea[j] + eb[j] + g[j], 0.f);
                        c[j] = t;
                        mk |= (t > 0.f ? 1u : 0u) << j;
                    }
                    *(uint4*)(yg + e0) = pack8(c);
                    a.ep_mask_out[e0 >> 3] = (unsigned char)mk;
                } else {
                    const unsigned mk = pmk[pss];
#pragma unroll
                    for (int j = 0; j < 8; ++j) g[j] = ((mk >> j) & 1u) ? g[j] : 0.f;
                    if (a.epi == 2) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) { s8[j] += g[j]; q8[j] += g[j] * c[j]; }
                    } else {
#pragma unroll
                        for (int j = 0; j < 8; ++j) c[j] = ea[j] * g[j] + eb[j] * c[j] + ec[j];
                        *(uint4*)(yg + e0) = pack8(c);
                    }
                }
            }
        }
        if (a.epi == 2) fold_stats(s8, q8);
        return;
    }
    if (yg == nullptr) return;          // statistics-only pass

    // epi 4 (dgrad; the parity classes of a strided dgrad partition the output, so every element is seen once):
    // besides writing the tile g = dL/d(activation), accumulate what the BatchNorm in front
    // of that activation needs for ITS backward - sum(g'), sum(g' * c) with g' = g under the ReLU bit mask
    // ep_mask_in and c = ep_src the BatchNorm's raw input - so no separate reduction pass re-reads g.
    if (epi_dgrad_stats) {
        // the BatchNorm input and its ReLU bits for this thread's 8 passes (the accumulators are in LDS by now, their
        // registers free); the addend operands, when present, were issued before the staging barrier
        const unsigned ybytes = (unsigned)a.N * a.Ho * a.Wo * a.K * 2u;
        const __amdgpu_buffer_rsrc_t crs = __builtin_amdgcn_make_buffer_rsrc((void*)a.ep_src, 0, ybytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc((void*)a.ep_mask_in, 0, ybytes >> 4, 0x00020000);
        u32x4 cx[NPASS];
        unsigned kx[NPASS];
#pragma unroll
        for (int pss = 0; pss < NPASS; ++pss) {
            cx[pss] = __builtin_amdgcn_raw_buffer_load_b128(crs, eoff[pss] != ~0u ? eoff[pss] * 2u : 0xFFFFFFF0u, 0, 0);
            kx[pss] = __builtin_amdgcn_raw_buffer_load_b8(krs, eoff[pss] != ~0u ? eoff[pss] >> 3 : 0xFFFFFFF0u, 0, 0);
        }
        float s8[8], q8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { s8[j] = 0.f; q8[j] = 0.f; }
#pragma unroll
        for (int pss = 0; pss < NPASS; ++pss) {
            const int row = pss * RPP + orow;
            if (eoff[pss] != ~0u) {
                uint4 v = *(const uint4*)(cs + row * CS + och * 8);
                float g[8], c[8];
                unpack8(v, g);
                if (pre_src != nullptr) {                  // residual / shortcut gradient joins first (as in the plain write-out)
                    float ad[8];
                    unpack8(__builtin_bit_cast(uint4, pre[pss]), ad);
                    const unsigned am = pmk[pss];
#pragma unroll
                    for (int j = 0; j < 8; ++j) g[j] += ((am >> j) & 1u) ? ad[j] : 0.f;
                    v = pack8(g);
                    unpack8(v, g);                         // statistics of the bf16 values that are stored
                }
                *(uint4*)(yg + (size_t)eoff[pss]) = v;
                unpack8(__builtin_bit_cast(uint4, cx[pss]), c);
                const unsigned mk = kx[pss];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float gm = ((mk >> j) & 1u) ? g[j] : 0.f;
                    s8[j] += gm; q8[j] += gm * c[j];
                }
            }
        }
        fold_stats(s8, q8);
        return;
    }

#pragma unroll
    for (int pss = 0; pss < NPASS; ++pss) {
        const int row = pss * RPP + orow;
        if (eoff[pss] != ~0u) {
            uint4 v = *(const uint4*)(cs + row * CS + och * 8);
            if (epi_gelu_bwd) {                    // g = the GELU's input: the activation's backward in the epilogue
                float f[8], g[8];
                unpack8(v, f); unpack8(__builtin_bit_cast(uint4, pre[pss]), g);
#pragma unroll
                for (int j = 0; j < 8; ++j) f[j] *= gelu_tanh_grad_fast(g[j]);
                v = pack8(f);
            } else if (pre_src != nullptr) {
                float f[8], g[8];
                unpack8(v, f); unpack8(__builtin_bit_cast(uint4, pre[pss]), g);
                const unsigned mk = pmk[pss];
#pragma unroll
                for (int j = 0; j < 8; ++j) f[j] += ((mk >> j) & 1u) ? g[j] : 0.f;
                v = pack8(f);
            }
            *(uint4*)(yg + (size_t)eoff[pss]) = v;
            if (MODE == 0 && a.y_act != nullptr) {     // the activated twin, from the bf16 values just stored
                float f[8];
                unpack8(v, f);
#pragma unroll
                for (int j = 0; j < 8; ++j) f[j] = gelu_tanh(f[j]);
                *(uint4*)((unsigned short*)a.y_act + (size_t)eoff[pss]) = pack8(f);
            }
        }
    }
}

#undef DPA_LOAD_TILE
#undef DPA_DMA_TILE
#undef DPA_STORE_TILE
#undef DPA_COMPUTE

// Dynamic LDS of an instantiation (0 when the static array is used); above 64 KiB the kernel must be
// allowed the size once per process.
template <typename KernelT>
size_t dynamic_lds(KernelT kernel, int bytes) {
    if (bytes <= 65536) return 0;
    static bool allowed = false;        // one flag per KernelT instantiation = per kernel
    if (!allowed) {
        const hipError_t err = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (err != hipSuccess) {
            fprintf(stderr, "conv_igemm: cannot allow %d bytes of LDS: %s\n", bytes, hipGetErrorString(err));
            abort();
        }
        allowed = true;
    }
    return (size_t)bytes;
}

template <int WP, int WC, int STG>
void launch_by_parity(const ConvArgs& a, hipStream_t st) {      // strided dgrad: one grid row per pixel parity class
    const long M = (long)a.N * a.Ho * a.Wo;
    const int Kg = a.R * a.S * a.C;
    const int tiles_c = (a.K + WC * 64 - 1) / (WC * 64);
    const int st_ = a.stride;
    const long mc = (long)a.N * ((a.Ho + st_ - 1) / st_) * ((a.Wo + st_ - 1) / st_);   // largest parity class
    const int tiles_p = (int)((mc + WP * 64 - 1) / (WP * 64));
    auto kernel = conv_igemm_kernel<WP, WC, 2, false, STG>;
    const size_t lds = dynamic_lds(kernel, igemm_lds_bytes(WP, WC, STG));
    hipLaunchKernelGGL(kernel, dim3(tiles_p * tiles_c, st_ * st_), dim3(256), lds, st, a, tiles_c, M, Kg);
}

template <int WP, int WC, bool ONE, int STG>
void launch(const ConvArgs& a, hipStream_t st) {
    if (a.mode != 0 && a.stride != 1) { launch_by_parity<WP, WC, STG>(a, st); return; }
    const long M = (long)a.N * a.Ho * a.Wo;
    const int Kg = a.R * a.S * a.C;
    const int tiles_c = (a.K + WC * 64 - 1) / (WC * 64);
    const int tiles_p = (int)((M + WP * 64 - 1) / (WP * 64));
    if (a.mode == 0) {
        auto kernel = conv_igemm_kernel<WP, WC, 0, ONE, STG>;
        const size_t lds = dynamic_lds(kernel, igemm_lds_bytes(WP, WC, STG));
        hipLaunchKernelGGL(kernel, dim3(tiles_p * tiles_c), dim3(256), lds, st, a, tiles_c, M, Kg);
    } else {
        auto kernel = conv_igemm_kernel<WP, WC, 1, ONE, STG>;
        const size_t lds = dynamic_lds(kernel, igemm_lds_bytes(WP, WC, STG));
        hipLaunchKernelGGL(kernel, dim3(tiles_p * tiles_c), dim3(256), lds, st, a, tiles_c, M, Kg);
    }
}

}  // namespace

int conv_stats_rows(long M) { (void)M; return 32; }

// Staging variant of a launch.  DPA_IGEMM_STG=0..3 forces one (experiments; 4 and 5 = DMA depth 4 / 6 exist in
// the template but are not instantiated: measured slower than depth 2 on every shape, few-workgroup GEMMs included).
// Measured per ResNet-50 layer and on the GPT-2 GEMMs (scripts/bench_conv.py, bench_gemm.py under each value):
// sum over the network's fprop + dgrad 7.95 ms LDS-DMA (2) / 8.07 counted DMA (3) / 8.43 register ring (1) /
// 8.56 single register set (0).  The counted depth-2 DMA wins where a CU holds only 1-3 workgroups (layer-4 3x3:
// 68.7 vs 78.3 us; the 8192-token GPT-2 GEMMs: +5-9 %), the 4-resident-workgroup plain DMA everywhere else.
static int staging(const ConvArgs& a, bool one) {
    static const int forced = [] { const char* e = getenv("DPA_IGEMM_STG"); return e ? atoi(e) : -1; }();
    if (forced >= 0) return forced > 3 ? 3 : forced;
    const long M = (long)a.N * a.Ho * a.Wo;
    const bool wide = a.K <= 64;
    const long bp = wide ? 256 : 128, bc = wide ? 64 : 128;
    long blocks = ((M + bp - 1) / bp) * ((a.K + bc - 1) / bc);
    if (a.mode != 0 && a.stride > 1) blocks /= (a.stride * a.stride);   // per parity class
    const long depth = (long)a.R * a.S * a.C;
    if (blocks <= 512) return 3;
    if (one && depth >= 768 && blocks <= 2048) return 3;
    return 2;
}

void conv_igemm(const ConvArgs& a, hipStream_t st) {
    if (wide_gemm_eligible(a)) { gemm_wide(a, st); return; }     // large plain GEMMs: the 256 x 256 main loop
    if (conv3x3_halo_ok(a)) { conv3x3_halo(a, st); return; }     // stage-1 3 x 3 layers: halo-tiled, filter in registers
    // 1x1 / no padding: plain GEMM gather (strided fprop included; strided dgrad goes by parity class)
    const bool one = a.R == 1 && a.S == 1 && a.pad == 0 && (a.mode == 0 || a.stride == 1);
    const int stg = staging(a, one);
#define DPA_DISPATCH(WP_, WC_, ONE_)                                         \
    switch (stg) {                                                           \
        case 0: launch<WP_, WC_, ONE_, 0>(a, st); break;                     \
        case 1: launch<WP_, WC_, ONE_, 1>(a, st); break;                     \
        case 2: launch<WP_, WC_, ONE_, 2>(a, st); break;                     \
        default: launch<WP_, WC_, ONE_, 3>(a, st); break;                    \
    }
    if (a.K <= 64) { if (one) DPA_DISPATCH(4, 1, true) else DPA_DISPATCH(4, 1, false) }
    else           { if (one) DPA_DISPATCH(2, 2, true) else DPA_DISPATCH(2, 2, false) }
#undef DPA_DISPATCH
}

// ------------------------------------------------------------------ filter transpose
__global__ void transpose_filter_kernel(const unsigned short* __restrict__ w, unsigned short* __restrict__ wt,
                                        int K, int RS, int C) {
    // tile 32x32 over (k, c) for a fixed rs; LDS transpose
    __shared__ unsigned short t[32][33];
    const int rs = blockIdx.z;
    const int k0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int i = ty; i < 32; i += 8) {
        int k = k0 + i, c = c0 + tx;
        t[i][tx] = (k < K && c < C) ? w[((long)k * RS + rs) * C + c] : (unsigned short)0;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        int c = c0 + i, k = k0 + tx;
        if (c < C && k < K) wt[((long)c * RS + rs) * K + k] = t[tx][i];
    }
}

// Every conv filter of a model in ONE launch: table row = (src offset, dst offset, K, RS, C, first tile), offsets in
// elements of the flat bf16 parameter shadow / its transposed twin.  Replaces one ~5 us launch per layer per step.
// 64 x 64 tiles.  Filters whose K, C and offsets are multiples of 8 (every layer of the three model families) move as
// 16-byte chunks: a wave reads 8 rows x 128 contiguous bytes and writes 8 rows x 128 contiguous bytes (the 32 x 32
// scalar version touched 64-byte row pieces with 2-byte accesses: 1.9 TB/s on GPT-2's 124M weights, 264 us per step).
constexpr int TR_TILE = 64;
__global__ __launch_bounds__(256) void transpose_filters_batched_kernel(const unsigned short* __restrict__ src,
                                                                        unsigned short* __restrict__ dst,
                                                                        const long* __restrict__ table, int nconv) {
    __shared__ unsigned short t[TR_TILE][TR_TILE + 2];   // 33-word rows: the column gather below spreads over the banks
    int lo = 0, hi = nconv - 1;                       // last conv whose first tile <= blockIdx.x
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[mid * 6 + 5] <= (long)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const long* e = table + lo * 6;
    const unsigned short* w = src + e[0];
    unsigned short* wt = dst + e[1];
    const int K = (int)e[2], RS = (int)e[3], C = (int)e[4];
    int tile = (int)((long)blockIdx.x - e[5]);
    const int tc = (C + TR_TILE - 1) / TR_TILE, tk = (K + TR_TILE - 1) / TR_TILE;
    const int rs = tile / (tc * tk); tile -= rs * tc * tk;
    const int k0 = (tile / tc) * TR_TILE, c0 = (tile % tc) * TR_TILE;
    const int tid = threadIdx.x;
    if ((((long)K | (long)C | e[0] | e[1]) & 7) == 0) {
        const int ch = tid & 7;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = (tid >> 3) + 32 * i, k = k0 + row, c = c0 + ch * 8;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (k < K && c < C) v = *(const uint4*)(w + ((long)k * RS + rs) * C + c);
            unsigned* d = (unsigned*)&t[row][ch * 8];
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int crow = (tid >> 3) + 32 * i, c = c0 + crow, k = k0 + ch * 8;
            if (c < C && k < K) {
                unsigned short g[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) g[j] = t[ch * 8 + j][crow];
                uint4 v;
                v.x = g[0] | ((unsigned)g[1] << 16); v.y = g[2] | ((unsigned)g[3] << 16);
                v.z = g[4] | ((unsigned)g[5] << 16); v.w = g[6] | ((unsigned)g[7] << 16);
                *(uint4*)(wt + ((long)c * RS + rs) * K + k) = v;
            }
        }
        return;
    }
    for (int idx = tid; idx < TR_TILE * TR_TILE; idx += 256) {
        const int i = idx >> 6, x = idx & 63, k = k0 + i, c = c0 + x;
        t[i][x] = (k < K && c < C) ? w[((long)k * RS + rs) * C + c] : (unsigned short)0;
    }
    __syncthreads();
    for (int idx = tid; idx < TR_TILE * TR_TILE; idx += 256) {
        const int i = idx >> 6, x = idx & 63, c = c0 + i, k = k0 + x;
        if (c < C && k < K) wt[((long)c * RS + rs) * K + k] = t[x][i];
    }
}

int transpose_tile() { return TR_TILE; }

void transpose_filters_batched(const void* src, void* dst, const long* table, int nconv, long total_tiles, hipStream_t st) {
    hipLaunchKernelGGL(transpose_filters_batched_kernel, dim3((unsigned)total_tiles), dim3(256), 0, st,
                       (const unsigned short*)src, (unsigned short*)dst, table, nconv);
}

void transpose_filter(const void* w, void* wt, int K, int RS, int C, hipStream_t st) {
    dim3 grid((C + 31) / 32, (K + 31) / 32, RS), block(256);
    hipLaunchKernelGGL(transpose_filter_kernel, grid, block, 0, st, (const unsigned short*)w, (unsigned short*)wt, K, RS, C);
}

// ------------------------------------------------------------------ column sums (bias gradient)
__global__ __launch_bounds__(256) void colsum_kernel(const unsigned short* __restrict__ x, float* __restrict__ out,
                                                     long M, int K, long rows_per_block) {
    // thread -> (8-channel chunk, row lane); K % 8 == 0
    const int cpr = K / 8;
    const int ch = threadIdx.x % cpr, rl = threadIdx.x / cpr, nrl = 256 / cpr;
    __shared__ float red[256 * 8];
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (rl < nrl) {
        const long r0 = blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
        for (long r = r0 + rl; r < r1; r += nrl) {
            uint4 v = *(const uint4*)(x + r * K + ch * 8);
            float f[8]; unpack8(v, f);
#pragma unroll
            for (int j = 0; j < 8; ++j) s[j] += f[j];
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) red[threadIdx.x * 8 + j] = (rl < nrl) ? s[j] : 0.f;
    __syncthreads();
    // first cpr*8 threads... each output channel summed over row lanes
    for (int c = threadIdx.x; c < K; c += 256) {
        float t = 0.f;
        const int cch = c / 8, cj = c % 8;
        for (int l = 0; l < nrl; ++l) t += red[(l * cpr + cch) * 8 + cj];
        atomicAdd(out + c, t);
    }
}

__global__ __launch_bounds__(256) void colsum_wide_kernel(const unsigned short* __restrict__ x, float* __restrict__ part,
                                                          long M, int K, long rows_per_block) {
    // wide matrices (K/8 > 256): thread -> 16-byte chunk (grid.y tiles the channels), 4 independent row
    // streams per thread; every block writes its own partial row (no same-address atomics), folded afterwards
    const int ch = blockIdx.y * 256 + threadIdx.x;
    if (ch * 8 >= K) return;
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const long r0 = blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
    const unsigned short* px = x + ch * 8;
    long r = r0;
    for (; r + 4 <= r1; r += 4) {
        uint4 v0 = *(const uint4*)(px + (r + 0) * K), v1 = *(const uint4*)(px + (r + 1) * K);
        uint4 v2 = *(const uint4*)(px + (r + 2) * K), v3 = *(const uint4*)(px + (r + 3) * K);
        float f0[8], f1[8], f2[8], f3[8];
        unpack8(v0, f0); unpack8(v1, f1); unpack8(v2, f2); unpack8(v3, f3);
#pragma unroll
        for (int j = 0; j < 8; ++j) s[j] += (f0[j] + f1[j]) + (f2[j] + f3[j]);
    }
    for (; r < r1; ++r) {
        uint4 v = *(const uint4*)(px + r * K);
        float f[8]; unpack8(v, f);
#pragma unroll
        for (int j = 0; j < 8; ++j) s[j] += f[j];
    }
    float* dst = part + (long)blockIdx.x * K + ch * 8;
    *(float4*)(dst) = make_float4(s[0], s[1], s[2], s[3]);
    *(float4*)(dst + 4) = make_float4(s[4], s[5], s[6], s[7]);
}

int colsum_workspace_rows(long M, int K) {
    if (K / 8 <= 256) return 0;
    long b = (M + 31) / 32;
    if (b > 256) b = 256;
    if (b < 1) b = 1;
    return (int)b;
}

void colsum_bf16(const void* x, float* out, long M, int K, float* workspace, hipStream_t st) {
    const int cpr = K / 8;
    if (cpr <= 256) {
        int blocks = (int)min((long)512, (M + 63) / 64);
        if (blocks < 1) blocks = 1;
        long rpb = (M + blocks - 1) / blocks;
        hipLaunchKernelGGL(colsum_kernel, dim3(blocks), dim3(256), 0, st, (const unsigned short*)x, out, M, K, rpb);
    } else {
        const int by = (cpr + 255) / 256;
        const int blocks = colsum_workspace_rows(M, K);
        const long rpb = (M + blocks - 1) / blocks;
        hipLaunchKernelGGL(colsum_wide_kernel, dim3(blocks, by), dim3(256), 0, st, (const unsigned short*)x, workspace, M, K, rpb);
        reduce_partials(workspace, blocks, K, out, 1, st);   // out += column sums of the partial rows
    }
}

}  // namespace dpa
