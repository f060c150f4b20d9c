// Causal multi-head self-attention (flash-style, no S x S matrix in memory) for gfx950,
// head dim DH = 64 or 128, any sequence length, qkv packed as [B, T, 3, H, DH] bf16 (the layout the fused QKV GEMM emits).
//
// Ragged T (T % 128 != 0): the grid covers ceil(T / 128) blocks; rows past T are CLAMPED to row T - 1 when loaded (valid
// memory, finite data) and never stored.  Keys past T need no extra mask - they are in every real query's future, so
// the causal mask of the diagonal tiles already removes them; queries past T are given lse = +big in the key-stationary
// kernel, which makes their probabilities exactly 0.
//
// MFMA: v_mfma_f32_32x32x16_bf16 throughout.  The score tile is computed TRANSPOSED
// (S^T = K . Q^T: keys in the accumulator registers, the query on the lane), so
//   * softmax statistics are per-lane scalars (one cross-half shuffle per row reduction),
//   * the accumulator registers are directly the B operand of the next product
//     (O^T += V^T . P^T, dQ^T += K^T . dS^T): no LDS round trip for P / dS,
//   * the V^T / K^T A-operands come from gfx950's transposing LDS read
//     (ds_read_b64_tr_b16) of the row-major [key][d] tile.
// Backward recomputes P from Q, K and the saved log-sum-exp; dK/dV are accumulated by a
// key-stationary kernel (no atomics: a workgroup owns its 128 keys), dQ by a
// query-stationary kernel that recomputes S^T (2 extra MFMA products, no atomics, no
// cross-workgroup reduction).
//
// Workgroup = 4 waves; each wave owns 32 queries (fwd, dQ) or 32 keys (dK/dV); K/V (or
// Q/dO) tiles of 64 rows are staged through LDS and shared by the 4 waves.
#include <math.h>
#include <stdlib.h>
#include <atomic>
#include "common.h"
#include "kernels.h"

namespace dpa {
namespace {

constexpr int TILE = 64;                 // rows per staged tile
constexpr float NEG_BIG = -1.0e30f;
template <int DH> constexpr int tile_bytes() { return TILE * DH * 2; }   // 8 / 16 KiB

typedef __attribute__((address_space(3))) short4v* lds_s4_ptr;

// byte offset of element (row, col) in a [64][DH] bf16 LDS image that serves row reads (ds_read_b128) and transposed
// reads (ds_read_b64_tr_b16) alike.  DH = 64 (128-byte rows, two rows per 256-byte bank row): the 16-byte chunk index
// XORed with (bit 1 of the row) << 2 | (row >> 2) & 3.  Bit 2 of the key moves rows 4m+2, 4m+3 to the other 64-byte
// window than rows 4m, 4m+1, so the 4 rows x 64 bytes of a transposed read cover the bank row once; the low bits make
// the 8 same-parity rows of a ds_read_b128 lane group ({0-3, 12-15, 20-27} / {4-11, 16-19, 28-31}) use 8 different
// chunks.  (A 64-byte-granule swizzle served the transposed reads only: PMC showed the row reads 4-way conflicted,
// 45-59 % of the three kernels' LDS cycles - profiles/pmc/attn_*.txt.)  DH = 128 (256-byte rows): the chunk index
// XORed with ((row & 3) << 2 | (row >> 2) & 3) - conflict-free for both kinds of read.
template <int DH>
__device__ __forceinline__ int img_off(int row, int col) {
    if (DH == 64) return row * 128 + ((((col >> 3) ^ ((((row >> 1) & 1) << 2) | ((row >> 2) & 3))) & 7) << 4) + (col & 7) * 2;
    return row * 256 + ((((col >> 3) ^ (((row & 3) << 2) | ((row >> 2) & 3))) & 15) << 4) + (col & 7) * 2;
}

// Global loads of a 64 x DH bf16 tile (row stride `rs` elements, first row `r0`, rows clamped to `rlast`) split in
// two, so the next tile's loads are in flight under the current tile's MFMAs: fetch_tile issues them into registers,
// stash_tile writes the registers to the LDS image
template <int DH>
__device__ __forceinline__ void fetch_tile(const unsigned short* __restrict__ g, long rs, int r0, int rlast, int tid,
                                           u32x4 (&r)[DH / 32]) {
    constexpr int CPR = DH / 8, RSTEP = 256 / CPR;
    const int c = tid % CPR;
#pragma unroll
    for (int i = 0; i < DH / 32; ++i) {
        int row = r0 + tid / CPR + i * RSTEP;
        row = row < rlast ? row : rlast;
        r[i] = *(const u32x4*)(g + (long)row * rs + c * 8);
    }
}
template <int DH>
__device__ __forceinline__ void stash_tile(unsigned char* lds, int tid, const u32x4 (&r)[DH / 32]) {
    constexpr int CPR = DH / 8, RSTEP = 256 / CPR;
    const int c = tid % CPR;
#pragma unroll
    for (int i = 0; i < DH / 32; ++i) *(u32x4*)(lds + img_off<DH>(tid / CPR + i * RSTEP, c * 8)) = r[i];
}

// MFMA A/B row fragment: 8 consecutive d of one row (k = 8h + j + 16kk)
template <int DH>
__device__ __forceinline__ bf16x8 row_frag(const unsigned char* lds, int row, int kk, int h) {
    return *(const bf16x8*)(lds + img_off<DH>(row, 16 * kk + 8 * h));
}

// transposed A fragment for  Y^T[d][..] += T^T[d x rows] . X[rows x ..]  where X is an
// accumulator tile used as B operand: element j <-> row rowbase + 8(j>>2) + (j&3)
// (rowbase already contains 16s + 4h), d = dbase + (lane & 31)
template <int DH>
__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* lds, int rowbase, int dbase, int lane) {
    const int gi = lane >> 4, li = lane & 15, tq = li >> 2, tp = li & 3;
    const int col = dbase + (gi & 1) * 16 + 4 * tp;
    const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(lds + img_off<DH>(rowbase + tq, col)));
    const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(lds + img_off<DH>(rowbase + 8 + tq, col)));
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

__device__ __forceinline__ bf16x8 pack_regs(const floatx16& x, int s) {   // registers 8s .. 8s+7 -> bf16x8
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = (__bf16)x[8 * s + j];
    return r;
}

// 2^x as ONE v_exp_f32: arguments here are <= 0 (score minus running max / log-sum-exp), so exp2f's denormal
// and overflow range handling (5 extra VALU instructions per element) buys nothing; very negative -> 0.
typedef float float2v __attribute__((ext_vector_type(2)));
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// The two halves of a wave hold the two halves of a query's row statistics: combine lane i with lane i ^ 32 through
// gfx950's v_permlane32_swap (a vector-ALU instruction) instead of __shfl_xor's ds_bpermute, which is an LDS round trip
// with a wait in the middle of every tile's dependency chain.  swap(x, x) returns (low half twice, high half twice).
// (The results are copied into scalars before the bit cast: __builtin_bit_cast applied to `r[1]` itself reads element 0 with
// this compiler - the sum came out as 2 * r[0].)
__device__ __forceinline__ void xhalf_swap(float x, float& lo, float& hi) {
    const unsigned u = __builtin_bit_cast(unsigned, x);
    const u32x2 r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    const unsigned a = r.x, b = r.y;
    lo = __builtin_bit_cast(float, a);
    hi = __builtin_bit_cast(float, b);
}
__device__ __forceinline__ float xhalf_max(float x) {
    float lo, hi;
    xhalf_swap(x, lo, hi);
    return fmaxf(lo, hi);
}
__device__ __forceinline__ float xhalf_sum(float x) {
    float lo, hi;
    xhalf_swap(x, lo, hi);
    return lo + hi;
}

// Attention dropout (GPT-2's attn_pdrop): element (b, head, q, k) of the probability matrix is kept or dropped by the
// counter hash of common.h at index ((b * H + head) * T + q) * T + k of the stream (seed, offset) - the same pure
// function in the forward kernel and in both backward kernels, so no mask is stored (elementwise.hip's dropout_mask
// with n = B * H * T * T reproduces the [B, H, T, T] mask for tests).  p = 0: every use is behind a wave-uniform branch.
struct AttnDrop { float p; uint64_t seed, offset; };
__device__ __forceinline__ float attn_keep(const AttnDrop& d, uint64_t row_base, int k, float keep_scale) {
    return dropout_keep(d.seed, d.offset, row_base + (uint64_t)k, d.p, keep_scale);
}

__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

__device__ __forceinline__ void zero16(floatx16& a) {
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = 0.f;
}

// store transposed accumulators (rows = d, col = this lane's token) as bf16 [token][DH]
template <int DH>
__device__ __forceinline__ void store_T(unsigned short* __restrict__ dst, const floatx16 (&acc)[DH / 32], float scale, int h) {
#pragma unroll
    for (int di = 0; di < DH / 32; ++di)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int d = 32 * di + 8 * g + 4 * h;
            uint2 pk;
            pk.x = pack_bf16x2(acc[di][4 * g + 0] * scale, acc[di][4 * g + 1] * scale);
            pk.y = pack_bf16x2(acc[di][4 * g + 2] * scale, acc[di][4 * g + 3] * scale);
            *(uint2*)(dst + d) = pk;
        }
}

// ------------------------------------------------------------------------------ forward
template <int DH, bool DROP, int OCC>
__global__ __launch_bounds__(256, OCC) void attn_fwd_kernel(const unsigned short* __restrict__ qkv, unsigned short* __restrict__ out,
                                                       float* __restrict__ lse, int T, int H, float c_log2, AttnDrop drop) {
    constexpr int KK = DH / 16, DI = DH / 32, TB = tile_bytes<DH>();
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * TB];
    unsigned char* Ks = smem;
    unsigned char* Vs = smem + TB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ql = lane & 31, h = lane >> 5;
    const int head = blockIdx.x, b = blockIdx.y;                    // block index slowest: see launch_fwd
    const int q0 = (gridDim.z - 1 - blockIdx.z) * 128, qw0 = q0 + 32 * wave, q = qw0 + ql;
    const int qc = q < T ? q : T - 1;                               // rows past T: clamped loads, no stores
    const long rs = 3L * H * DH;                                    // token stride of qkv
    const unsigned short* base = qkv + (long)b * T * rs + head * DH;  // (b, t=0, q, head)
    const unsigned short* kbase = base + (long)H * DH;
    const unsigned short* vbase = base + 2L * H * DH;

    bf16x8 qf[KK];
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) qf[kk] = *(const bf16x8*)(base + (long)qc * rs + 16 * kk + 8 * h);

    floatx16 O[DI];
#pragma unroll
    for (int di = 0; di < DI; ++di) zero16(O[di]);
    float m = NEG_BIG, l = 0.f;
    int ntiles = (q0 + 128) / TILE;
    if (ntiles > (T + TILE - 1) / TILE) ntiles = (T + TILE - 1) / TILE;
    u32x4 pk[DH / 32], pv[DH / 32];
    fetch_tile<DH>(kbase, rs, 0, T - 1, tid, pk);
    fetch_tile<DH>(vbase, rs, 0, T - 1, tid, pv);
    for (int j = 0; j < ntiles; ++j) {
        __syncthreads();
        stash_tile<DH>(Ks, tid, pk);
        stash_tile<DH>(Vs, tid, pv);
        __syncthreads();
        if (j + 1 < ntiles) {
            fetch_tile<DH>(kbase, rs, (j + 1) * TILE, T - 1, tid, pk);
            fetch_tile<DH>(vbase, rs, (j + 1) * TILE, T - 1, tid, pv);
        }
        if (j * TILE > qw0 + 31) continue;                       // wave-uniform: every key is in the future
        floatx16 S[2];
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
            zero16(S[t2]);
#pragma unroll
            for (int kk = 0; kk < KK; ++kk)
                S[t2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag<DH>(Ks, 32 * t2 + ql, kk, h), qf[kk], S[t2], 0, 0, 0);
        }
        const bool diag = (j * TILE + TILE - 1) > qw0;             // tile may contain keys > some query of the wave
        // VALU diet (the kernel is bound by the vector ALU at head dimension 64, not by the MFMAs): the running maximum is
        // taken over the RAW scores and scaled once per row (c_log2 > 0), the scale itself rides the exponent's fused
        // multiply-add - 32 multiplications per tile gone -, and the accumulator is rescaled only in tiles that raise some
        // row's maximum (after the first few tiles of a row almost none do: wave-uniform branch).
        float mx = NEG_BIG;
        if (diag) {                                              // wave-uniform: only these tiles pay for the mask
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float s = S[t2][r];
                    if ((j * TILE + 32 * t2 + acc_row(r, h)) > qc) s = NEG_BIG;
                    S[t2][r] = s;
                    mx = fmaxf(mx, s);
                }
        } else {
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, S[t2][r]);
        }
        mx = xhalf_max(mx);
        const float mn = fmaxf(m, mx * c_log2);                  // (masked scores: -1e30 * c stays hugely negative)
        const bool grew = mn > m;
        float rsum = 0.f;
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = fast_exp2(fmaf(S[t2][r], c_log2, -mn));
                S[t2][r] = p;
                rsum += p;
            }
        rsum = xhalf_sum(rsum);
        if (__any(grew)) {
            const float alpha = fast_exp2(m - mn);
            l *= alpha;
#pragma unroll
            for (int di = 0; di < DI; ++di)
#pragma unroll
                for (int r = 0; r < 16; ++r) O[di][r] *= alpha;
        }
        l += rsum;
        m = mn;
        if (DROP) {                                    // the normaliser l sees the undropped probabilities
            const float ks = 1.f / (1.f - drop.p);
            const uint64_t rb = ((uint64_t)(b * H + head) * T + qc) * T;
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                for (int r = 0; r < 16; ++r) S[t2][r] *= attn_keep(drop, rb, j * TILE + 32 * t2 + acc_row(r, h), ks);
        }
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bf16x8 pb = pack_regs(S[t2], s);
#pragma unroll
                for (int di = 0; di < DI; ++di)
                    O[di] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag<DH>(Vs, 32 * t2 + 16 * s + 4 * h, 32 * di, lane), pb,
                                                                    O[di], 0, 0, 0);
            }
    }
    if (q < T) {
        const float inv = 1.f / l;
        store_T<DH>(out + ((long)(b * T + q) * H + head) * DH, O, inv, h);
        if (h == 0) lse[((long)b * H + head) * T + q] = (m + log2f(l)) * 0.6931471805599453f;
    }
}

// ------------------------------------------------------------------------------ forward, software-pipelined (DH = 64, no dropout)
// OPT-IN, and slower as measured (see launch_fwd): kept as the record of the experiment the round-2 review asked for.
// The kernel above runs score MFMAs -> softmax -> P V MFMAs one after the other inside a wave and leaves the overlap to
// the three waves of a SIMD; removing a fifth of its vector-ALU work changed nothing (profiles/r3/attn_fwd_lm_ab.txt), and
// an ablation (profiles/r3/attn_fwd_ablation.txt) shows every part's time adding up: no softmax -9.7 us of 46.5, no score
// MFMAs -9.7, no P V -7.0, no staging -10.2, no exponentials -0.8.  Here a wave works on HALF tiles (32 keys) and keeps two score blocks: the MFMAs
// of half tile g + 1 are issued in front of the softmax of half tile g and run in the matrix pipe under its vector work;
// the P V MFMAs of g run under the softmax of g + 1.  K lives in a ring of three LDS tiles (half tile 2j + 2 needs tile
// j + 1 while tile j is still being read), V in two, so ONE barrier per 64 keys covers both the hand-over of the freshly
// written tiles and the reuse of the oldest ones (the kernel above needs two).  Half tiles also make the causal cut
// exact: a wave visits key blocks 0 .. qw0 / 32 and only the last one is masked.
__device__ __forceinline__ floatx16 score_half(const unsigned char* Kt, int row0, int ql, int h, const bf16x8 (&qf)[4]) {
    floatx16 S;
    zero16(S);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
        S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag<64>(Kt, row0 + ql, kk, h), qf[kk], S, 0, 0, 0);
    return S;
}

__global__ __launch_bounds__(256, 2) void attn_fwd_pipe_kernel(const unsigned short* __restrict__ qkv, unsigned short* __restrict__ out,
                                                               float* __restrict__ lse, int T, int H, float c_log2) {
    constexpr int DH = 64, KK = 4, DI = 2, TB = tile_bytes<64>();
    __shared__ __attribute__((aligned(16))) unsigned char smem[5 * TB];     // K ring of 3, V ring of 2: 40 KiB
    unsigned char* const Kr = smem;
    unsigned char* const Vr = smem + 3 * TB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ql = lane & 31, h = lane >> 5;
    const int head = blockIdx.x, b = blockIdx.y;
    const int q0 = (gridDim.z - 1 - blockIdx.z) * 128, qw0 = q0 + 32 * wave, q = qw0 + ql;
    const int qc = q < T ? q : T - 1;
    const long rs = 3L * H * DH;
    const unsigned short* base = qkv + (long)b * T * rs + head * DH;
    const unsigned short* kbase = base + (long)H * DH;
    const unsigned short* vbase = base + 2L * H * DH;

    bf16x8 qf[KK];
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) qf[kk] = *(const bf16x8*)(base + (long)qc * rs + 16 * kk + 8 * h);
    floatx16 O[DI];
#pragma unroll
    for (int di = 0; di < DI; ++di) zero16(O[di]);
    float m = NEG_BIG, l = 0.f;
    int ntiles = (q0 + 128) / TILE;
    if (ntiles > (T + TILE - 1) / TILE) ntiles = (T + TILE - 1) / TILE;
    int gmax = qw0 / 32;                                            // the wave's last (diagonal) half tile ...
    if (gmax > 2 * ntiles - 1) gmax = 2 * ntiles - 1;               // ... (ragged T: waves whose queries are all past T)

    u32x4 pk[2], pv[2];
    fetch_tile<DH>(kbase, rs, 0, T - 1, tid, pk);
    fetch_tile<DH>(vbase, rs, 0, T - 1, tid, pv);
    stash_tile<DH>(Kr, tid, pk);
    stash_tile<DH>(Vr, tid, pv);
    if (ntiles > 1) {
        fetch_tile<DH>(kbase, rs, TILE, T - 1, tid, pk);
        stash_tile<DH>(Kr + TB, tid, pk);
    }
    __syncthreads();
    if (ntiles > 2) fetch_tile<DH>(kbase, rs, 2 * TILE, T - 1, tid, pk);     // K of tile j + 2 and V of tile j + 1 ride in registers
    if (ntiles > 1) fetch_tile<DH>(vbase, rs, TILE, T - 1, tid, pv);
    floatx16 Sc = score_half(Kr, 0, ql, h, qf), Sn;

    // softmax of one half tile + its P V products; `diag`: the wave's own key block, masked above the diagonal
#define DPA_ATTN_HALF(G, VT, ROW0)                                                                                     \
    {                                                                                                                  \
        float mx = NEG_BIG;                                                                                            \
        if ((G) == gmax) {                                                                                             \
            const int lim = qc - 32 * (G) - 4 * h;                   /* row acc_row(r, 0) is visible iff <= lim */     \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                           \
                Sc[r] = acc_row(r, 0) <= lim ? Sc[r] : NEG_BIG;                                                        \
                mx = fmaxf(mx, Sc[r]);                                                                                 \
            }                                                                                                          \
        } else {                                                                                                       \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) mx = fmaxf(mx, Sc[r]);                                      \
        }                                                                                                              \
        mx = xhalf_max(mx);                                                                        \
        const float mn = fmaxf(m, mx * c_log2);                                                                        \
        const bool grew = mn > m;                                                                                      \
        float rsum = 0.f;                                                                                              \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                               \
            const float p = fast_exp2(fmaf(Sc[r], c_log2, -mn));                                                       \
            Sc[r] = p;                                                                                                 \
            rsum += p;                                                                                                 \
        }                                                                                                              \
        rsum = xhalf_sum(rsum);                                                                              \
        if (__any(grew)) {                                                                                             \
            const float alpha = fast_exp2(m - mn);                                                                     \
            l *= alpha;                                                                                                \
            _Pragma("unroll") for (int di = 0; di < DI; ++di)                                                          \
                _Pragma("unroll") for (int r = 0; r < 16; ++r) O[di][r] *= alpha;                                      \
        }                                                                                                              \
        l += rsum;                                                                                                     \
        m = mn;                                                                                                        \
        _Pragma("unroll") for (int s = 0; s < 2; ++s) {                                                                \
            const bf16x8 pb = pack_regs(Sc, s);                                                                        \
            _Pragma("unroll") for (int di = 0; di < DI; ++di)                                                          \
                O[di] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag<DH>(VT, (ROW0) + 16 * s + 4 * h, 32 * di, lane), pb,  \
                                                                O[di], 0, 0, 0);                                       \
        }                                                                                                              \
    }

    int kcur = 0, knext = 1, kfree = 2;                             // ring slots of K tiles j, j + 1, j + 2
    for (int j = 0; j < ntiles; ++j) {
        if (j > 0) __syncthreads();                                 // tiles written in step j - 1 are visible, tiles read in it are free
        if (j + 2 < ntiles) stash_tile<DH>(Kr + kfree * TB, tid, pk);
        if (j + 1 < ntiles) stash_tile<DH>(Vr + ((j + 1) & 1) * TB, tid, pv);
        if (j + 3 < ntiles) fetch_tile<DH>(kbase, rs, (j + 3) * TILE, T - 1, tid, pk);
        if (j + 2 < ntiles) fetch_tile<DH>(vbase, rs, (j + 2) * TILE, T - 1, tid, pv);
        const int g0 = 2 * j;
        if (g0 <= gmax) {                                           // wave-uniform
            const unsigned char* Kt = Kr + kcur * TB;
            const unsigned char* Vt = Vr + (j & 1) * TB;
            if (g0 + 1 <= gmax) Sn = score_half(Kt, 32, ql, h, qf);                 // half tile 2j + 1, under the softmax below
            DPA_ATTN_HALF(g0, Vt, 0)
            if (g0 + 1 <= gmax) {
                Sc = Sn;
                if (g0 + 2 <= gmax) Sn = score_half(Kr + knext * TB, 0, ql, h, qf);   // half tile 2j + 2 (tile j + 1)
                DPA_ATTN_HALF(g0 + 1, Vt, 32)
                Sc = Sn;
            }
        }
        const int t = kcur; kcur = knext; knext = kfree; kfree = t;
    }
#undef DPA_ATTN_HALF
    if (q < T) {
        const float inv = 1.f / l;
        store_T<DH>(out + ((long)(b * T + q) * H + head) * DH, O, inv, h);
        if (h == 0) lse[((long)b * H + head) * T + q] = (m + log2f(l)) * 0.6931471805599453f;
    }
}

// ------------------------------------------------------------------------------ dK / dV (key-stationary)
// (206 registers: two waves per SIMD.  Forced into 168 for three - __launch_bounds__(256, 3), 124 bytes of scratch inside the
// loop - the pair of backward kernels went from 168 to 221 us at 16 x 12 x 1024: profiles/r3/attn_bwd_kv_occ_ab.txt.)
template <int DH, bool DROP>
__global__ __launch_bounds__(256, DH == 64 ? 2 : 1) void attn_bwd_kv_kernel(const unsigned short* __restrict__ qkv, const unsigned short* __restrict__ d_o,
                                                          const float* __restrict__ lse, const float* __restrict__ delta,
                                                          unsigned short* __restrict__ dqkv, int T, int H, float scale, AttnDrop drop) {
    constexpr int KK = DH / 16, DI = DH / 32, TB = tile_bytes<DH>();
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * TB + 2 * TILE * 4];
    unsigned char* Qs = smem;
    unsigned char* Os = smem + TB;
    float* lse_s = (float*)(smem + 2 * TB);
    float* del_s = lse_s + TILE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kl = lane & 31, h = lane >> 5;
    const int head = blockIdx.x, b = blockIdx.y;
    const int k0 = blockIdx.z * 128, kw0 = k0 + 32 * wave, key = kw0 + kl;
    const int keyc = key < T ? key : T - 1;
    const long rs = 3L * H * DH, ors = (long)H * DH;
    const unsigned short* base = qkv + (long)b * T * rs + head * DH;
    const unsigned short* dobase = d_o + (long)b * T * ors + head * DH;
    const float* lse_b = lse + ((long)b * H + head) * T;
    const float* del_b = delta + ((long)b * H + head) * T;
    const float c_log2 = scale * 1.4426950408889634f;

    bf16x8 kf[KK], vf[KK];
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
        kf[kk] = *(const bf16x8*)(base + (long)keyc * rs + (long)H * DH + 16 * kk + 8 * h);
        vf[kk] = *(const bf16x8*)(base + (long)keyc * rs + 2L * H * DH + 16 * kk + 8 * h);
    }
    floatx16 dKt[DI], dVt[DI];
#pragma unroll
    for (int di = 0; di < DI; ++di) { zero16(dKt[di]); zero16(dVt[di]); }

    const int nq = (T + TILE - 1) / TILE;
    u32x4 pq[DH / 32], po[DH / 32];
    fetch_tile<DH>(base, rs, (k0 / TILE) * TILE, T - 1, tid, pq);
    fetch_tile<DH>(dobase, ors, (k0 / TILE) * TILE, T - 1, tid, po);
    for (int jq = k0 / TILE; jq < nq; ++jq) {
        __syncthreads();
        stash_tile<DH>(Qs, tid, pq);
        stash_tile<DH>(Os, tid, po);
        if (tid < TILE) {
            const int qi = jq * TILE + tid;                       // queries past T: lse = +big makes their probabilities 0
            lse_s[tid] = qi < T ? lse_b[qi] * 1.4426950408889634f : 1.0e30f;
            del_s[tid] = qi < T ? del_b[qi] * scale : 0.f;          // pre-scaled: dS = p * (dP * scale - delta * scale)
        }
        __syncthreads();
        if (jq + 1 < nq) {
            fetch_tile<DH>(base, rs, (jq + 1) * TILE, T - 1, tid, pq);
            fetch_tile<DH>(dobase, ors, (jq + 1) * TILE, T - 1, tid, po);
        }
        if (jq * TILE + TILE - 1 < kw0) continue;                 // wave-uniform: all queries are before the wave's keys
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
            const int qt0 = jq * TILE + 32 * t2;                  // first query of this 32-row half tile
            if (qt0 + 31 < kw0) continue;                         // wave-uniform: still entirely before the wave's keys
            floatx16 S, dP;
            zero16(S); zero16(dP);
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
                S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag<DH>(Qs, 32 * t2 + kl, kk, h), kf[kk], S, 0, 0, 0);
                dP = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag<DH>(Os, 32 * t2 + kl, kk, h), vf[kk], dP, 0, 0, 0);
            }
            // rows = queries (registers), column = this lane's key
            if (DROP) {              // dropped probabilities: dV sees p * keep, dS = p * (keep * dP - delta)
                const float ks = 1.f / (1.f - drop.p);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int qr = 32 * t2 + acc_row(r, h), qg = min(jq * TILE + qr, T - 1);
                    const float kp = attn_keep(drop, ((uint64_t)(b * H + head) * T + qg) * T, keyc, ks);
                    const float p = (jq * TILE + qr >= key) ? fast_exp2(S[r] * c_log2 - lse_s[qr]) : 0.f;
                    S[r] = p * kp;
                    dP[r] = p * (dP[r] * kp * scale - del_s[qr]);
                }
            } else {
                // two query rows at a time through the packed fp32 pipe (v_pk_fma_f32 / v_pk_mul_f32): the kernel is bound by
                // the vector ALU, and rows r, r + 1 of an accumulator are neighbours in the query index (acc_row)
                const float2v c2 = {c_log2, c_log2}, sc2 = {scale, scale};
                const float* ls = lse_s + 32 * t2 + 4 * h;
                const float* dl = del_s + 32 * t2 + 4 * h;
                if (qt0 < kw0 + 31) {                                      // diagonal half tile: causal mask (wave-uniform)
                    const int qk = key - (jq * TILE + 32 * t2 + 4 * h);    // row acc_row(r, 0) is visible iff >= qk
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const int a = acc_row(r, 0);
                        float2v e = float2v{S[r], S[r + 1]} * c2 - *(const float2v*)(ls + a);
                        e.x = a >= qk ? e.x : NEG_BIG;                     // selects, not branches
                        e.y = a + 1 >= qk ? e.y : NEG_BIG;
                        const float2v p = {fast_exp2(e.x), fast_exp2(e.y)};
                        const float2v ds = p * (float2v{dP[r], dP[r + 1]} * sc2 - *(const float2v*)(dl + a));
                        S[r] = p.x; S[r + 1] = p.y;
                        dP[r] = ds.x; dP[r + 1] = ds.y;
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const int a = acc_row(r, 0);
                        const float2v e = float2v{S[r], S[r + 1]} * c2 - *(const float2v*)(ls + a);
                        const float2v p = {fast_exp2(e.x), fast_exp2(e.y)};
                        const float2v ds = p * (float2v{dP[r], dP[r + 1]} * sc2 - *(const float2v*)(dl + a));
                        S[r] = p.x; S[r + 1] = p.y;
                        dP[r] = ds.x; dP[r + 1] = ds.y;
                    }
                }
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bf16x8 pb = pack_regs(S, s), dsb = pack_regs(dP, s);
                const int rowbase = 32 * t2 + 16 * s + 4 * h;
#pragma unroll
                for (int di = 0; di < DI; ++di) {
                    dVt[di] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag<DH>(Os, rowbase, 32 * di, lane), pb, dVt[di], 0, 0, 0);
                    dKt[di] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag<DH>(Qs, rowbase, 32 * di, lane), dsb, dKt[di], 0, 0, 0);
                }
            }
        }
    }
    if (key < T) {
        unsigned short* dst = dqkv + (long)(b * T + key) * rs + head * DH;
        store_T<DH>(dst + (long)H * DH, dKt, 1.f, h);
        store_T<DH>(dst + 2L * H * DH, dVt, 1.f, h);
    }
}

// ------------------------------------------------------------------------------ dQ (query-stationary)
template <int DH, bool DROP>
__global__ __launch_bounds__(256, DH == 64 ? 2 : 1) void attn_bwd_q_kernel(const unsigned short* __restrict__ qkv, const unsigned short* __restrict__ d_o,
                                                         const unsigned short* __restrict__ o_fwd,
                                                         const float* __restrict__ lse, float* __restrict__ delta,
                                                         unsigned short* __restrict__ dqkv, int T, int H, float scale, AttnDrop drop) {
    constexpr int KK = DH / 16, DI = DH / 32, TB = tile_bytes<DH>();
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * TB];
    unsigned char* Ks = smem;
    unsigned char* Vs = smem + TB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ql = lane & 31, h = lane >> 5;
    const int head = blockIdx.x, b = blockIdx.y;                    // block index slowest: see launch_fwd
    const int q0 = (gridDim.z - 1 - blockIdx.z) * 128, qw0 = q0 + 32 * wave, q = qw0 + ql;
    const int qc = q < T ? q : T - 1;
    const long rs = 3L * H * DH, ors = (long)H * DH;
    const unsigned short* base = qkv + (long)b * T * rs + head * DH;
    const unsigned short* kbase = base + (long)H * DH;
    const unsigned short* vbase = base + 2L * H * DH;
    const float c_log2 = scale * 1.4426950408889634f;

    // delta_q = rowsum(dO * O) of this lane's query, formed here from the dO fragments the kernel loads anyway plus the
    // matching O fragments (a lane holds half of the row: one cross-half shuffle), and WRITTEN for the key-stationary
    // kernel, which is launched after this one: no separate delta kernel (8.7 us + a launch, 12 x per GPT-2 step)
    bf16x8 qf[KK], dof[KK];
    float del_q = 0.f;
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
        qf[kk] = *(const bf16x8*)(base + (long)qc * rs + 16 * kk + 8 * h);
        dof[kk] = *(const bf16x8*)(d_o + ((long)(b * T + qc) * H + head) * DH + 16 * kk + 8 * h);
        const bf16x8 of = *(const bf16x8*)(o_fwd + ((long)(b * T + qc) * H + head) * DH + 16 * kk + 8 * h);
#pragma unroll
        for (int j = 0; j < 8; ++j) del_q += (float)dof[kk][j] * (float)of[j];
    }
    del_q = xhalf_sum(del_q);
    if (h == 0 && q < T) delta[((long)b * H + head) * T + q] = del_q;
    const float lse_q = lse[((long)b * H + head) * T + qc] * 1.4426950408889634f;
    floatx16 dQt[DI];
#pragma unroll
    for (int di = 0; di < DI; ++di) zero16(dQt[di]);

    int ntiles = (q0 + 128) / TILE;
    if (ntiles > (T + TILE - 1) / TILE) ntiles = (T + TILE - 1) / TILE;
    u32x4 pk[DH / 32], pv[DH / 32];
    fetch_tile<DH>(kbase, rs, 0, T - 1, tid, pk);
    fetch_tile<DH>(vbase, rs, 0, T - 1, tid, pv);
    for (int j = 0; j < ntiles; ++j) {
        __syncthreads();
        stash_tile<DH>(Ks, tid, pk);
        stash_tile<DH>(Vs, tid, pv);
        __syncthreads();
        if (j + 1 < ntiles) {
            fetch_tile<DH>(kbase, rs, (j + 1) * TILE, T - 1, tid, pk);
            fetch_tile<DH>(vbase, rs, (j + 1) * TILE, T - 1, tid, pv);
        }
        if (j * TILE > qw0 + 31) continue;
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
            const int kt0 = j * TILE + 32 * t2;                   // first key of this 32-row half tile
            if (kt0 > qw0 + 31) continue;                         // wave-uniform: entirely in the wave's future
            floatx16 St, dPt;
            zero16(St); zero16(dPt);
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
                St = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag<DH>(Ks, 32 * t2 + ql, kk, h), qf[kk], St, 0, 0, 0);
                dPt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag<DH>(Vs, 32 * t2 + ql, kk, h), dof[kk], dPt, 0, 0, 0);
            }
            if (DROP) {
                const float ks = 1.f / (1.f - drop.p);
                const uint64_t rb = ((uint64_t)(b * H + head) * T + qc) * T;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int kg = kt0 + acc_row(r, h);
                    const float p = (kg <= qc) ? fast_exp2(St[r] * c_log2 - lse_q) : 0.f;
                    dPt[r] = p * (dPt[r] * attn_keep(drop, rb, min(kg, T - 1), ks) - del_q) * scale;
                }
            } else {
                const float2v c2 = {c_log2, c_log2}, sc2 = {scale, scale}, l2 = {lse_q, lse_q}, d2 = {del_q * scale, del_q * scale};
                if (kt0 + 31 > qw0) {                               // diagonal half tile: causal mask (wave-uniform)
                    const int kq = qc - (kt0 + 4 * h);              // key row acc_row(r, 0) is visible iff <= kq
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {               // packed fp32 pairs, as in the key-stationary kernel
                        const int a = acc_row(r, 0);
                        float2v e = float2v{St[r], St[r + 1]} * c2 - l2;
                        e.x = a <= kq ? e.x : NEG_BIG;
                        e.y = a + 1 <= kq ? e.y : NEG_BIG;
                        const float2v p = {fast_exp2(e.x), fast_exp2(e.y)};
                        const float2v ds = p * (float2v{dPt[r], dPt[r + 1]} * sc2 - d2);
                        dPt[r] = ds.x; dPt[r + 1] = ds.y;
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const float2v e = float2v{St[r], St[r + 1]} * c2 - l2;
                        const float2v p = {fast_exp2(e.x), fast_exp2(e.y)};
                        const float2v ds = p * (float2v{dPt[r], dPt[r + 1]} * sc2 - d2);
                        dPt[r] = ds.x; dPt[r + 1] = ds.y;
                    }
                }
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bf16x8 dsb = pack_regs(dPt, s);
#pragma unroll
                for (int di = 0; di < DI; ++di)
                    dQt[di] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag<DH>(Ks, 32 * t2 + 16 * s + 4 * h, 32 * di, lane), dsb,
                                                                      dQt[di], 0, 0, 0);
            }
        }
    }
    if (q < T) store_T<DH>(dqkv + (long)(b * T + q) * rs + head * DH, dQt, 1.f, h);
}

// Grid = (head, batch, 128-row block), the block index SLOWEST.  Workgroups go to the 8 XCDs round-robin by linear id
// (x fastest), and under the causal mask a block's work grows with its index: with the block index on x and
// T = 1024 (8 blocks) every XCD received ONE block index - XCD 0 all the 18-tile blocks, XCD 7 all the 2-tile ones, the
// kernel as slow as a chip an eighth the size on the heaviest eighth.  With (head, batch) fastest every XCD sees every
// block index equally often, heaviest blocks are dispatched first, and - 8 | H * B permitting - the blocks of one
// (batch, head) land on one XCD, whose L2 then holds that head's K and V once.
std::atomic<int> g_fwd_pipe{[] { const char* e = getenv("DPA_ATTN_FWD_PIPE"); return e ? atoi(e) : 0; }()};

template <int DH>
void launch_fwd(const void* qkv, void* out, float* lse, int B, int T, int H, AttnDrop drop, hipStream_t st) {
    const float c_log2 = (1.0f / sqrtf((float)DH)) * 1.4426950408889634f;   // 1/sqrt(DH) * log2(e)
    // Without dropout the head-dimension-64 kernel also builds into 128 registers (four waves per SIMD instead of three).
    // Measured (profiles/r3/attn_occ_ab.txt): 48.6-49.3 us against 47.2-48.5 us at 16 x 12 x 1024, so the three-wave build
    // stays the default; DPA_ATTN_FWD_OCC=4 selects the other one.
    static const bool occ4 = [] { const char* e = getenv("DPA_ATTN_FWD_OCC"); return e && atoi(e) == 4; }();
    constexpr int OCC = DH == 64 ? 2 : 1;
    // The software-pipelined kernel is opt-in (DPA_ATTN_FWD_PIPE=1 or set_attention_fwd_pipe): measured 54-55 us against
    // 47-48 us at 16 x 12 x 1024 and 194 against 172 us at 4 x 12 x 4096 (profiles/r3/attn_fwd_pipe_ab.txt).
    const bool pipe = g_fwd_pipe.load(std::memory_order_relaxed) != 0;
    if (DH == 64 && pipe && !(drop.p > 0.f)) {
        hipLaunchKernelGGL(attn_fwd_pipe_kernel, dim3(H, B, (T + 127) / 128), dim3(256), 0, st, (const unsigned short*)qkv,
                           (unsigned short*)out, lse, T, H, c_log2);
        return;
    }
    if (drop.p > 0.f)
        hipLaunchKernelGGL((attn_fwd_kernel<DH, true, OCC>), dim3(H, B, (T + 127) / 128), dim3(256), 0, st, (const unsigned short*)qkv,
                           (unsigned short*)out, lse, T, H, c_log2, drop);
    else if (DH == 64 && occ4)
        hipLaunchKernelGGL((attn_fwd_kernel<DH, false, DH == 64 ? 4 : OCC>), dim3(H, B, (T + 127) / 128), dim3(256), 0, st, (const unsigned short*)qkv,
                           (unsigned short*)out, lse, T, H, c_log2, drop);
    else
        hipLaunchKernelGGL((attn_fwd_kernel<DH, false, OCC>), dim3(H, B, (T + 127) / 128), dim3(256), 0, st, (const unsigned short*)qkv,
                           (unsigned short*)out, lse, T, H, c_log2, drop);
}

template <int DH>
void launch_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, float* delta,
                int B, int T, int H, AttnDrop drop, hipStream_t st) {
    const float scale = 1.0f / sqrtf((float)DH);
    // query-stationary kernel first: it also produces delta = rowsum(dO * O) for the key-stationary kernel behind it
#define DPA_ATTN_BWD(DROP_)                                                                                                          \
    hipLaunchKernelGGL((attn_bwd_q_kernel<DH, DROP_>), dim3(H, B, (T + 127) / 128), dim3(256), 0, st, (const unsigned short*)qkv,    \
                       (const unsigned short*)dout, (const unsigned short*)out, lse, delta, (unsigned short*)dqkv, T, H, scale, drop); \
    hipLaunchKernelGGL((attn_bwd_kv_kernel<DH, DROP_>), dim3(H, B, (T + 127) / 128), dim3(256), 0, st, (const unsigned short*)qkv,   \
                       (const unsigned short*)dout, lse, delta, (unsigned short*)dqkv, T, H, scale, drop)
    if (drop.p > 0.f) { DPA_ATTN_BWD(true); } else { DPA_ATTN_BWD(false); }
#undef DPA_ATTN_BWD
}

}  // namespace

void set_attention_fwd_pipe(int on) { g_fwd_pipe.store(on ? 1 : 0, std::memory_order_relaxed); }

void attention_fwd(const void* qkv, void* out, float* lse, int B, int T, int H, int Dh, float drop_p, uint64_t seed,
                   uint64_t offset, hipStream_t st) {
    const AttnDrop drop{drop_p, seed, offset};
    if (Dh == 128) launch_fwd<128>(qkv, out, lse, B, T, H, drop, st);
    else launch_fwd<64>(qkv, out, lse, B, T, H, drop, st);
}

void attention_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, float* delta,
                   int B, int T, int H, int Dh, float drop_p, uint64_t seed, uint64_t offset, hipStream_t st) {
    const AttnDrop drop{drop_p, seed, offset};
    if (Dh == 128) launch_bwd<128>(qkv, out, dout, lse, dqkv, delta, B, T, H, drop, st);
    else launch_bwd<64>(qkv, out, dout, lse, dqkv, delta, B, T, H, drop, st);
}

}  // namespace dpa
