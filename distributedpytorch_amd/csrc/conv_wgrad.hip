// Filter-gradient (wgrad) implicit GEMM for gfx950.
//
//   dw[k][kk] += sum_p dy[p][k] * im2col(x)[p][kk],   p = (n,ho,wo),  kk = (r,s,c)
//
// The reduction index is the PIXEL, which is the slow axis of both NHWC operands,
// so neither operand is K-contiguous.  Tiles are loaded as they lie in memory
// ([pixel][channel], 16 B per lane, fully coalesced) and the MFMA fragments are
// produced by gfx950's hardware-transposing LDS read ds_read_b64_tr_b16 (4 rows x
// 16 columns delivered column-major), so no software transpose pass exists.  The
// LDS image is XOR-swizzled at 64-byte granularity so each half-wave's transposed
// read covers a whole 256-byte bank row (conflict-free).
//
// MFMA: v_mfma_f32_32x32x16_bf16.  One accumulator register spans 2 rows x 32
// columns = two 128-byte segments, which is the full-rate shape for fp32 global
// atomics; the pixel axis is split over blockIdx.y and partial tiles are added into
// the fp32 gradient with global_atomic_add_f32 (gradients stay fp32 end to end).
//
// Block = 4 waves as 2x2 (128 out-channels x 128 filter columns) or 1x4
// (64 x 256, for 64-channel layers); 32 pixels per K step; one barrier per step.  Two staging
// variants: DMA (default) - both operands go global -> LDS with `buffer_load ... lds`, two tiles in
// flight over three LDS buffers under a counted vmcnt, the XOR swizzle applied on the source side;
// and the register ring it replaced (three register sets, ds_write_b128), kept selectable
// (DPA_WGRAD_DMA=0) as the measured baseline.
#include <stdlib.h>
#include "common.h"
#include "kernels.h"

namespace dpa {
namespace {

constexpr int BKP = 32;  // pixels per K step

typedef __attribute__((address_space(3))) short4v* lds_s4_ptr;

// 64-byte-granule XOR key of a tile row: the 4 rows one half-wave's transposed read
// touches must land in 4 different 64-byte windows of the 256-byte bank row.
template <int ROW_BYTES>
__device__ __forceinline__ int swz_key(int row) {
    return ROW_BYTES == 128 ? ((row >> 1) & 1) : (row & 3);
}

__device__ __forceinline__ short4v tr_read(const unsigned char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(p));
}

// LIN: 1x1 filter, stride 1, no padding (and every Linear layer): the x operand is a plain
// [pixel][channel] matrix like dy, so neither gather needs pixel arithmetic.
// BIAS: the workgroups of the first column tile also accumulate dbias[k] = sum_p dy[p][k] - one more MFMA per
// dy fragment against a B operand of ones, in the waves of column 0 only - instead of a separate pass over dy.
template <int WM, int WN, bool LIN, bool DMA, bool BIAS>
__global__ __launch_bounds__(256, DMA ? 3 : 2) void conv_wgrad_kernel(WgradArgs a, int tiles_n, int tiles, long P, int Kg, int steps_per_split) {
    constexpr int BM = WM * 64;           // out-channel rows of dw per block
    constexpr int BN = WN * 64;           // filter columns (r,s,c) per block
    constexpr int A_CPR = BM / 8;         // 16-B chunks per dy tile row
    constexpr int B_CPR = BN / 8;
    constexpr int A_LD = BKP * A_CPR / 256;  // loads per thread
    constexpr int B_LD = BKP * B_CPR / 256;
    constexpr int A_BYTES = BKP * BM * 2, B_BYTES = BKP * BN * 2;
    constexpr int STAGE = A_BYTES + B_BYTES;
    constexpr int NBUF = 3;
    __shared__ __attribute__((aligned(16))) unsigned char smem[NBUF * STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    // XCD-aware bijective remap (workgroups go round-robin over the 8 XCDs): each XCD walks a
    // contiguous range of (split, tile) pairs, tile fastest, so the output tiles that share one
    // pixel range - and therefore the same dy / x rows - run behind the same L2.
    int lid;
    {
        const int nwg = gridDim.x, bid = blockIdx.x;
        const int q = nwg >> 3, r8 = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        lid = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + idx;
    }
    const int tile = lid % tiles, split_id = lid / tiles;
    const int tile_n = tile % tiles_n, tile_m = tile / tiles_n;
    const int k0 = tile_m * BM;   // first out channel
    const int n0 = tile_n * BN;   // first filter column

    const long p_begin = (long)split_id * steps_per_split * BKP;
    long p_end = p_begin + (long)steps_per_split * BKP;
    if (p_end > P) p_end = P;
    if (p_begin >= p_end) return;
    const int nsteps = (int)((p_end - p_begin + BKP - 1) / BKP);

    // Buffer descriptors with 32-bit byte offsets; out-of-range offsets read zeros, so tile edges,
    // image padding and the end of this block's pixel range need no branch and no select.  The
    // dy descriptor ends at this block's last pixel: rows past p_end vanish by themselves.
    const unsigned dybytes = (unsigned)p_end * (unsigned)a.K * 2u;
    const unsigned xbytes = LIN ? (unsigned)p_end * (unsigned)a.C * 2u : (unsigned)a.N * a.H * a.W * a.C * 2u;
    const __amdgpu_buffer_rsrc_t dyrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, dybytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, xbytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;   // operands are < 2^31 bytes (host check): offsets >= 2^31 are always out of range

    // ---- A (dy) loads: thread -> (pixel row, channel chunk); the offset advances by a constant ----
    const int a_ch = tid % A_CPR, a_row0 = tid / A_CPR;
    constexpr int A_RSTEP = 256 / A_CPR;
    // DMA writes LDS in lane order, so the lane that owns slot `a_ch` of a row fetches the chunk the swizzle
    // puts there (the XOR key is the same for all of a thread's rows: they are A_RSTEP = 16 or 32 apart)
    const int a_src = DMA ? ((((a_ch >> 2) ^ swz_key<BM * 2>(a_row0)) << 2) | (a_ch & 3)) : a_ch;
    unsigned a_off[A_LD];
#pragma unroll
    for (int i = 0; i < A_LD; ++i) {
        a_off[i] = ((unsigned)(p_begin + a_row0 + i * A_RSTEP) * (unsigned)a.K + (unsigned)(k0 + a_src * 8)) * 2u;
        if (k0 + a_src * 8 >= a.K) a_off[i] |= OOB;
    }
    const unsigned a_step = (unsigned)BKP * (unsigned)a.K * 2u;

    // ---- B (x gather) loads: thread's filter column is fixed for the whole loop ----
    const int b_ch = tid % B_CPR, b_row0 = tid / B_CPR;
    constexpr int B_RSTEP = 256 / B_CPR;
    const int b_src = DMA ? ((((b_ch >> 2) ^ swz_key<BN * 2>(b_row0)) << 2) | (b_ch & 3)) : b_ch;
    const int col = n0 + b_src * 8;
    const bool b_cok = col < Kg;
    unsigned b_off[B_LD];                 // LIN: running byte offset
    int pn[B_LD], pho[B_LD], pwo[B_LD];   // general: pixel state (n, ho, wo) of each row
    int prem[B_LD];                       // general: pixels left before p_end
    int fr = 0, fs = 0, fc = 0;
    const unsigned b_step = (unsigned)BKP * (unsigned)a.C * 2u;
    const int q32 = BKP / a.Wo, r32 = BKP - q32 * a.Wo;
    if (LIN) {
#pragma unroll
        for (int i = 0; i < B_LD; ++i) {
            b_off[i] = ((unsigned)(p_begin + b_row0 + i * B_RSTEP) * (unsigned)a.C + (unsigned)col) * 2u;
            if (!b_cok) b_off[i] |= OOB;
        }
    } else {
        const int cc = b_cok ? col : 0;
        const int rs = cc / a.C;
        fc = cc - rs * a.C;
        fr = rs / a.S;
        fs = rs - fr * a.S;
        const unsigned HoWo = (unsigned)(a.Ho * a.Wo);
#pragma unroll
        for (int i = 0; i < B_LD; ++i) {
            const unsigned p = (unsigned)(p_begin + b_row0 + i * B_RSTEP);   // P < 2^31 (host check)
            const unsigned n = p / HoWo, rem = p - n * HoWo;
            pn[i] = (int)n; pho[i] = (int)(rem / (unsigned)a.Wo); pwo[i] = (int)(rem - (unsigned)pho[i] * (unsigned)a.Wo);
            prem[i] = b_cok ? (int)(p_end - (long)p) : 0;
        }
    }

    // Three register sets + three LDS buffers: the loads of tile t+3 are issued before the MFMAs
    // of tile t and written to LDS two iterations later, so each load has about three iterations
    // of MFMA work (times the resident blocks) to arrive.
    u32x4 ar0[A_LD], br0[B_LD], ar1[A_LD], br1[B_LD], ar2[A_LD], br2[B_LD];

#define DPA_WG_LOAD(AREG, BREG)                                                                  \
    {                                                                                            \
        _Pragma("unroll") for (int i = 0; i < A_LD; ++i) {                                       \
            AREG[i] = __builtin_amdgcn_raw_buffer_load_b128(dyrs, a_off[i], 0, 0);               \
            a_off[i] += a_step;                                                                  \
        }                                                                                        \
        _Pragma("unroll") for (int i = 0; i < B_LD; ++i) {                                       \
            if (LIN) {                                                                           \
                BREG[i] = __builtin_amdgcn_raw_buffer_load_b128(xrs, b_off[i], 0, 0);            \
                b_off[i] += b_step;                                                              \
            } else {                                                                             \
                const int hi = pho[i] * a.stride - a.pad + fr;                                   \
                const int wi = (a.asym ? pwo[i] * a.stride_w - a.pad_w : pwo[i] * a.stride - a.pad) + fs; \
                const bool ok = prem[i] > 0 && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W; \
                const unsigned off = ((unsigned)((pn[i] * a.H + hi) * a.W + wi) * (unsigned)a.C + (unsigned)fc) * 2u; \
                BREG[i] = __builtin_amdgcn_raw_buffer_load_b128(xrs, ok ? off : OOB, 0, 0);      \
                prem[i] -= BKP;                                                                  \
                pwo[i] += r32; pho[i] += q32;                                                    \
                if (pwo[i] >= a.Wo) { pwo[i] -= a.Wo; ++pho[i]; }                                \
                while (pho[i] >= a.Ho) { pho[i] -= a.Ho; ++pn[i]; }                              \
            }                                                                                    \
        }                                                                                        \
    }

    // The same gather as DPA_WG_LOAD, landing in LDS buffer BUF: per instruction a wave fills 1 KiB
    // (lane l -> +16 l), which is rows 4w..4w+3 (256-byte rows) of the i-th 4 KiB slab of the operand.
    const u32x4 dyrs4 = buffer_words(a.dy, dybytes), xrs4 = buffer_words(a.x, xbytes);
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
#define DPA_WG_DMA(BUF)                                                                          \
    {                                                                                            \
        unsigned char* as_ = smem + (BUF) * STAGE + wave_u * 1024;                               \
        unsigned char* bs_ = as_ + A_BYTES;                                                      \
        _Pragma("unroll") for (int i = 0; i < A_LD; ++i) {                                       \
            dma16<true>(dyrs, dyrs4, as_ + i * 4096, a_off[i]);                                  \
            a_off[i] += a_step;                                                                  \
        }                                                                                        \
        _Pragma("unroll") for (int i = 0; i < B_LD; ++i) {                                       \
            if (LIN) {                                                                           \
                dma16<true>(xrs, xrs4, bs_ + i * 4096, b_off[i]);                                \
                b_off[i] += b_step;                                                              \
            } else {                                                                             \
                const int hi = pho[i] * a.stride - a.pad + fr;                                   \
                const int wi = (a.asym ? pwo[i] * a.stride_w - a.pad_w : pwo[i] * a.stride - a.pad) + fs; \
                const bool ok = prem[i] > 0 && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W; \
                const unsigned off = ((unsigned)((pn[i] * a.H + hi) * a.W + wi) * (unsigned)a.C + (unsigned)fc) * 2u; \
                dma16<true>(xrs, xrs4, bs_ + i * 4096, ok ? off : OOB);                          \
                prem[i] -= BKP;                                                                  \
                pwo[i] += r32; pho[i] += q32;                                                    \
                if (pwo[i] >= a.Wo) { pwo[i] -= a.Wo; ++pho[i]; }                                \
                while (pho[i] >= a.Ho) { pho[i] -= a.Ho; ++pn[i]; }                              \
            }                                                                                    \
        }                                                                                        \
    }

    // LDS image: [pixel][channel] rows, 64-byte granules XOR-swizzled by the pixel row
    int a_dst[A_LD], b_dst[B_LD];
#pragma unroll
    for (int i = 0; i < A_LD; ++i) {
        const int row = a_row0 + i * A_RSTEP;
        const int ch = (((a_ch >> 2) ^ swz_key<BM * 2>(row)) << 2) | (a_ch & 3);
        a_dst[i] = row * (BM * 2) + ch * 16;
    }
#pragma unroll
    for (int i = 0; i < B_LD; ++i) {
        const int row = b_row0 + i * B_RSTEP;
        const int ch = (((b_ch >> 2) ^ swz_key<BN * 2>(row)) << 2) | (b_ch & 3);
        b_dst[i] = A_BYTES + row * (BN * 2) + ch * 16;
    }
#define DPA_WG_STORE(BUF, AREG, BREG)                                                            \
    {                                                                                            \
        unsigned char* st_ = smem + (BUF) * STAGE;                                               \
        _Pragma("unroll") for (int i = 0; i < A_LD; ++i) *(u32x4*)(st_ + a_dst[i]) = AREG[i];    \
        _Pragma("unroll") for (int i = 0; i < B_LD; ++i) *(u32x4*)(st_ + b_dst[i]) = BREG[i];    \
    }

    floatx16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    floatx16 accb[2];                              // BIAS: every column of the tile holds the row's sum
    bf16x8 ones;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) accb[i][r] = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (__bf16)1.0f;
    const bool do_bias = BIAS && a.dbias != nullptr && tile_n == 0 && wn == 0;   // wave-uniform

    // transposed-read lane roles
    const int gi = lane >> 4, li = lane & 15, tq = li >> 2, tp = li & 3;
    const int kofs = 8 * (gi >> 1) + tq;        // pixel row inside a 16-deep k slice (first read)
    const int cofs = (gi & 1) * 16 + 4 * tp;    // column offset inside a 32-wide tile

#define DPA_WG_COMPUTE(BUF)                                                                      \
    {                                                                                            \
        const unsigned char* as = smem + (BUF) * STAGE;                                          \
        const unsigned char* bs = as + A_BYTES;                                                  \
        _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) {                                       \
            bf16x8 af[2], bf[2];                                                                 \
            const int r0 = kk * 16 + kofs, r1 = r0 + 4;                                          \
            _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                      \
                const int c = wm * 64 + i * 32 + cofs;                                           \
                const int g = c >> 5, w = (c & 31) * 2;                                          \
                short4v lo = tr_read(as + r0 * (BM * 2) + ((g ^ swz_key<BM * 2>(r0)) << 6) + w); \
                short4v hi = tr_read(as + r1 * (BM * 2) + ((g ^ swz_key<BM * 2>(r1)) << 6) + w); \
                af[i] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7)); \
            }                                                                                    \
            _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                      \
                const int c = wn * 64 + j * 32 + cofs;                                           \
                const int g = c >> 5, w = (c & 31) * 2;                                          \
                short4v lo = tr_read(bs + r0 * (BN * 2) + ((g ^ swz_key<BN * 2>(r0)) << 6) + w); \
                short4v hi = tr_read(bs + r1 * (BN * 2) + ((g ^ swz_key<BN * 2>(r1)) << 6) + w); \
                bf[j] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7)); \
            }                                                                                    \
            _Pragma("unroll") for (int i = 0; i < 2; ++i)                                        \
                _Pragma("unroll") for (int j = 0; j < 2; ++j)                                    \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0); \
            if (BIAS && do_bias) {                                                               \
                _Pragma("unroll") for (int i = 0; i < 2; ++i)                                    \
                    accb[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], ones, accb[i], 0, 0, 0); \
            }                                                                                    \
        }                                                                                        \
    }

    // tiles past the block's range load zeros (never consumed), so the loop carries no conditions
    if (DMA) {
        constexpr int LPT = A_LD + B_LD;                       // DMA instructions per tile per lane
        DPA_WG_DMA(0)
        DPA_WG_DMA(1)
        __builtin_amdgcn_s_waitcnt(waitcnt_vm_only(LPT));      // tile 0 has landed, tile 1 may still fly
        __syncthreads();
        int cur = 0, nxt = 2;
        for (int t = 0; t < nsteps; ++t) {
            DPA_WG_DMA(nxt)                                    // tile t + 2 (its buffer was last read in step t - 1)
            __builtin_amdgcn_sched_barrier(0);
            DPA_WG_COMPUTE(cur)
            __builtin_amdgcn_s_waitcnt(waitcnt_vm_only(LPT));  // tile t + 1 has landed
            __syncthreads();
            cur = cur == 2 ? 0 : cur + 1;
            nxt = nxt == 2 ? 0 : nxt + 1;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
    DPA_WG_LOAD(ar0, br0)
    DPA_WG_LOAD(ar1, br1)
    DPA_WG_LOAD(ar2, br2)
    DPA_WG_STORE(0, ar0, br0)
    __syncthreads();
    int t = 0;
    for (; t + 2 < nsteps; t += 3) {
        DPA_WG_LOAD(ar0, br0)              // tile t + 3
        __builtin_amdgcn_sched_barrier(0);
        DPA_WG_COMPUTE(0)                  // tile t
        DPA_WG_STORE(1, ar1, br1)          // tile t + 1
        __syncthreads();
        DPA_WG_LOAD(ar1, br1)              // tile t + 4
        __builtin_amdgcn_sched_barrier(0);
        DPA_WG_COMPUTE(1)
        DPA_WG_STORE(2, ar2, br2)          // tile t + 2
        __syncthreads();
        DPA_WG_LOAD(ar2, br2)              // tile t + 5
        __builtin_amdgcn_sched_barrier(0);
        DPA_WG_COMPUTE(2)
        DPA_WG_STORE(0, ar0, br0)          // tile t + 3
        __syncthreads();
    }
    if (t < nsteps) {                      // one or two tiles left; buffer 0 holds tile t
        DPA_WG_COMPUTE(0)
        if (t + 1 < nsteps) {
            DPA_WG_STORE(1, ar1, br1)
            __syncthreads();
            DPA_WG_COMPUTE(1)
        }
    }
    }

    // ---- accumulate into fp32 dw: each register = 2 rows x 32 contiguous columns ----
    const int ccol = lane & 31, rhalf = (lane >> 5) * 4;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int cg = n0 + wn * 64 + j * 32 + ccol;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = k0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + rhalf;
                if (row < a.K && cg < Kg) atomicAdd(a.dw + (long)row * Kg + cg, acc[i][j][r]);
            }
        }
    if (BIAS && do_bias && ccol == 0) {            // lanes 0 and 32 hold column 0: 16 rows each
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = k0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + rhalf;
                if (row < a.K) atomicAdd(a.dbias + row, accb[i][r]);
            }
    }
}

#undef DPA_WG_LOAD
#undef DPA_WG_DMA
#undef DPA_WG_STORE
#undef DPA_WG_COMPUTE

// Workgroups a launch is split into.  512 (two per CU) for long reductions; 384 when the reduction is short (<= 32k
// pixels / tokens: GPT-2's 8192-token GEMMs, the 14 x 14 and 7 x 7 stages at batch 128): measured on one box GPT-2
// 675.5k -> 686.0k tok/s (288..416 all within 0.3 %), ResNet-152 4,848 -> 4,891 img/s, while ResNet-50 at batch 256
// prefers 512 everywhere (13,370 vs 13,320 img/s).  Fewer splits = fewer fp32 atomic passes over dw.
int wgrad_target_blocks(long P) {
    static int v = [] {
        const char* e = getenv("DPA_WGRAD_BLOCKS");
        return e ? atoi(e) : 0;
    }();
    return v > 0 ? v : (P <= 32768 ? 384 : 512);
}

template <int WM, int WN, bool LIN, bool DMA, bool BIAS = false>
void launch(const WgradArgs& a, hipStream_t st) {
    const long P = (long)a.N * a.Ho * a.Wo;
    const int Kg = a.R * a.S * a.C;
    const int tiles_m = (a.K + WM * 64 - 1) / (WM * 64);
    const int tiles_n = (Kg + WN * 64 - 1) / (WN * 64);
    const int tiles = tiles_m * tiles_n;
    const int steps = (int)((P + BKP - 1) / BKP);
    // ~2 workgroups per CU: every extra split adds a full fp32 copy of dw to the atomic traffic
    // (chip-wide float-atomic rate is ~1.3 TB/s), so do not over-split large filters.
    int split = wgrad_target_blocks(P) / tiles;
    if (split < 1) split = 1;
    if (deterministic()) split = 1;            // one writer per dw element: no float atomics race (runtime.hip)
    int max_split = steps / 8;
    if (max_split < 1) max_split = 1;
    if (split > max_split) split = max_split;
    const int sps = (steps + split - 1) / split;
    split = (steps + sps - 1) / sps;
    dim3 grid(tiles * split), block(256);
    hipLaunchKernelGGL((conv_wgrad_kernel<WM, WN, LIN, DMA, BIAS>), grid, block, 0, st, a, tiles_n, tiles, P, Kg, sps);
}

}  // namespace

namespace {
bool use_dma(const WgradArgs& a) {
    // DPA_WGRAD_DMA=0/1 forces a staging variant.  Measured over the ResNet-50 layers (scripts/bench_conv.py):
    // DMA -8..-13 % on every 3x3 and on the 1x1 layers of stages 3-4, +4..+11 % on the 1x1 layers of stages 1-2,
    // which stream > 200k pixels against tiny filters (HBM-bound; the ring's three tiles in flight matter there).
    static const int forced = [] { const char* e = getenv("DPA_WGRAD_DMA"); return e ? atoi(e) : -1; }();
    const bool lin = a.R == 1 && a.S == 1 && a.pad == 0 && a.stride == 1;
    return forced >= 0 ? forced != 0 : !(lin && (long)a.N * a.Ho * a.Wo >= 150000);
}
}  // namespace

// the bias-gradient variant exists for the Linear-shaped DMA kernels (every GPT-2 projection)
bool conv_wgrad_fuses_bias(const WgradArgs& a) {
    if (wgrad_wide_alone_pays(a)) return true;
    const bool lin = a.R == 1 && a.S == 1 && a.pad == 0 && a.stride == 1;
    return lin && use_dma(a);
}

void conv_wgrad(const WgradArgs& a, hipStream_t st) {
    if (wgrad_wide_alone_pays(a)) { wgrad_wide_group(&a, 1, st); return; }  // large Linear layers: wgrad_wide.hip (256 x 256 tiles)
    if (conv_wgrad_halo_ok(a)) { conv_wgrad_halo(a, st); return; }     // 3 x 3 on 56- / 28-wide maps: conv_wgrad_halo.hip
    const bool lin = a.R == 1 && a.S == 1 && a.pad == 0 && a.stride == 1;
    const bool dma = use_dma(a);
    const bool bias = a.dbias != nullptr && conv_wgrad_fuses_bias(a);
#define DPA_WG_DISPATCH(WM_, WN_)                                                                \
    if (dma) {                                                                                   \
        if (lin) { if (bias) launch<WM_, WN_, true, true, true>(a, st); else launch<WM_, WN_, true, true>(a, st); } \
        else launch<WM_, WN_, false, true>(a, st);                                               \
    } else { if (lin) launch<WM_, WN_, true, false>(a, st); else launch<WM_, WN_, false, false>(a, st); }
    if (a.K <= 64) { DPA_WG_DISPATCH(1, 4) }
    else           { DPA_WG_DISPATCH(2, 2) }
#undef DPA_WG_DISPATCH
}

}  // namespace dpa
