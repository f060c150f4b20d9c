// Causal multi-head self-attention (flash-style, no S x S matrix in memory) for gfx950,
// head dim 64, qkv packed as [B, T, 3, H, 64] bf16 (the layout the fused QKV GEMM emits).
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
#include "common.h"
#include "kernels.h"

namespace dpa {
namespace {

constexpr int D = 64;
constexpr int TILE = 64;                 // rows per staged tile
constexpr int TILE_BYTES = TILE * D * 2; // 8 KiB
constexpr float NEG_BIG = -1.0e30f;

typedef __attribute__((address_space(3))) short4v* lds_s4_ptr;

// byte offset of element (row, col) in a [64][64] bf16 LDS image; 64-byte granules are
// XOR-swizzled with bit 1 of the row so the 4 rows of a transposed read hit 4 distinct
// 64-byte windows of the 256-byte bank row
__device__ __forceinline__ int img_off(int row, int col) {
    return row * 128 + ((((col >> 5) ^ (row >> 1)) & 1) << 6) + (col & 31) * 2;
}

// cooperative load of a 64 x 64 bf16 tile (row stride `rs` elements) into an LDS image
__device__ __forceinline__ void load_tile(const unsigned short* __restrict__ g, long rs, unsigned char* lds, int tid) {
    const int c = tid & 7;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (tid >> 3) + i * 32;
        const u32x4 v = *(const u32x4*)(g + (long)row * rs + c * 8);
        *(u32x4*)(lds + img_off(row, c * 8)) = v;
    }
}

// the same load split in two, so the next tile's global loads are in flight under the current tile's
// MFMAs: fetch_tile issues them into registers, stash_tile writes the registers to the LDS image
__device__ __forceinline__ void fetch_tile(const unsigned short* __restrict__ g, long rs, int tid, u32x4 (&r)[2]) {
    const int c = tid & 7;
#pragma unroll
    for (int i = 0; i < 2; ++i) r[i] = *(const u32x4*)(g + (long)((tid >> 3) + i * 32) * rs + c * 8);
}
__device__ __forceinline__ void stash_tile(unsigned char* lds, int tid, const u32x4 (&r)[2]) {
    const int c = tid & 7;
#pragma unroll
    for (int i = 0; i < 2; ++i) *(u32x4*)(lds + img_off((tid >> 3) + i * 32, c * 8)) = r[i];
}

// MFMA A/B row fragment: 8 consecutive d of one row (k = 8h + j + 16kk)
__device__ __forceinline__ bf16x8 row_frag(const unsigned char* lds, int row, int kk, int h) {
    return *(const bf16x8*)(lds + img_off(row, 16 * kk + 8 * h));
}

// transposed A fragment for  Y^T[d][..] += T^T[d x rows] . X[rows x ..]  where X is an
// accumulator tile used as B operand: element j <-> row rowbase + 8(j>>2) + (j&3)
// (rowbase already contains 16s + 4h), d = dbase + (lane & 31)
__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* lds, int rowbase, int dbase, int lane) {
    const int gi = lane >> 4, li = lane & 15, tq = li >> 2, tp = li & 3;
    const int col = dbase + (gi & 1) * 16 + 4 * tp;
    const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(lds + img_off(rowbase + tq, col)));
    const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(lds + img_off(rowbase + 8 + tq, col)));
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
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

__device__ __forceinline__ void zero16(floatx16& a) {
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = 0.f;
}

// store a transposed accumulator pair (rows = d, col = this lane's token) as bf16 [token][64]
__device__ __forceinline__ void store_T(unsigned short* __restrict__ dst, const floatx16 (&acc)[2], float scale, int h) {
#pragma unroll
    for (int di = 0; di < 2; ++di)
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
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(const unsigned short* __restrict__ qkv, unsigned short* __restrict__ out,
                                                       float* __restrict__ lse, int T, int H, float c_log2) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * TILE_BYTES];
    unsigned char* Ks = smem;
    unsigned char* Vs = smem + TILE_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ql = lane & 31, h = lane >> 5;
    const int head = blockIdx.y, b = blockIdx.z;
    const int q0 = (gridDim.x - 1 - blockIdx.x) * 128, qw0 = q0 + 32 * wave, q = qw0 + ql;
    const long rs = 3L * H * D;                                     // token stride of qkv
    const unsigned short* base = qkv + (long)b * T * rs + head * D;   // (b, t=0, q, head)
    const unsigned short* kbase = base + (long)H * D;
    const unsigned short* vbase = base + 2L * H * D;

    bf16x8 qf[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) qf[kk] = *(const bf16x8*)(base + (long)q * rs + 16 * kk + 8 * h);

    floatx16 O[2];
    zero16(O[0]); zero16(O[1]);
    float m = NEG_BIG, l = 0.f;
    const int ntiles = (q0 + 128) / TILE;
    u32x4 pk[2], pv[2];
    fetch_tile(kbase, rs, tid, pk);
    fetch_tile(vbase, rs, tid, pv);
    for (int j = 0; j < ntiles; ++j) {
        __syncthreads();
        stash_tile(Ks, tid, pk);
        stash_tile(Vs, tid, pv);
        __syncthreads();
        if (j + 1 < ntiles) {
            fetch_tile(kbase + (long)(j + 1) * TILE * rs, rs, tid, pk);
            fetch_tile(vbase + (long)(j + 1) * TILE * rs, rs, tid, pv);
        }
        if (j * TILE > qw0 + 31) continue;                       // wave-uniform: every key is in the future
        floatx16 S[2];
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
            zero16(S[t2]);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
                S[t2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(Ks, 32 * t2 + ql, kk, h), qf[kk], S[t2], 0, 0, 0);
        }
        const bool diag = (j * TILE + TILE - 1) > qw0;             // tile may contain keys > some query of the wave
        float mx = NEG_BIG;
        if (diag) {                                              // wave-uniform: only these tiles pay for the mask
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float s = S[t2][r] * c_log2;
                    if ((j * TILE + 32 * t2 + acc_row(r, h)) > q) s = NEG_BIG;
                    S[t2][r] = s;
                    mx = fmaxf(mx, s);
                }
        } else {
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float s = S[t2][r] * c_log2;
                    S[t2][r] = s;
                    mx = fmaxf(mx, s);
                }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mn = fmaxf(m, mx);
        const float alpha = fast_exp2(m - mn);
        float rsum = 0.f;
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = fast_exp2(S[t2][r] - mn);
                S[t2][r] = p;
                rsum += p;
            }
        rsum += __shfl_xor(rsum, 32, 64);
        l = l * alpha + rsum;
        m = mn;
#pragma unroll
        for (int di = 0; di < 2; ++di)
#pragma unroll
            for (int r = 0; r < 16; ++r) O[di][r] *= alpha;
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bf16x8 pb = pack_regs(S[t2], s);
#pragma unroll
                for (int di = 0; di < 2; ++di)
                    O[di] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(Vs, 32 * t2 + 16 * s + 4 * h, 32 * di, lane), pb,
                                                                    O[di], 0, 0, 0);
            }
    }
    const float inv = 1.f / l;
    store_T(out + ((long)(b * T + q) * H + head) * D, O, inv, h);
    if (h == 0) lse[((long)b * H + head) * T + q] = (m + log2f(l)) * 0.6931471805599453f;
}

// ------------------------------------------------------------------------------ delta = rowsum(dO * O)
__global__ __launch_bounds__(256) void attn_delta_kernel(const unsigned short* __restrict__ o, const unsigned short* __restrict__ d_o,
                                                         float* __restrict__ delta, int T, int H, long rows) {
    // one 8-lane group per (b, t, head) row of 64 elements
    const long row = (long)blockIdx.x * 32 + (threadIdx.x >> 3);
    const int c = threadIdx.x & 7;
    float s = 0.f;
    if (row < rows) {
        uint4 a = *(const uint4*)(o + row * D + c * 8), g = *(const uint4*)(d_o + row * D + c * 8);
        float fa[8], fg[8]; unpack8(a, fa); unpack8(g, fg);
#pragma unroll
        for (int j = 0; j < 8; ++j) s += fa[j] * fg[j];
    }
    s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
    if (row < rows && c == 0) {
        const long bt = row / H; const int head = (int)(row - bt * H);
        const long bi = bt / T; const int t = (int)(bt - bi * T);
        delta[(bi * H + head) * T + t] = s;
    }
}

// ------------------------------------------------------------------------------ dK / dV (key-stationary)
__global__ __launch_bounds__(256, 2) void attn_bwd_kv_kernel(const unsigned short* __restrict__ qkv, const unsigned short* __restrict__ d_o,
                                                          const float* __restrict__ lse, const float* __restrict__ delta,
                                                          unsigned short* __restrict__ dqkv, int T, int H, float scale) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * TILE_BYTES + 2 * TILE * 4];
    unsigned char* Qs = smem;
    unsigned char* Os = smem + TILE_BYTES;
    float* lse_s = (float*)(smem + 2 * TILE_BYTES);
    float* del_s = lse_s + TILE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kl = lane & 31, h = lane >> 5;
    const int head = blockIdx.y, b = blockIdx.z;
    const int k0 = blockIdx.x * 128, kw0 = k0 + 32 * wave, key = kw0 + kl;
    const long rs = 3L * H * D, ors = (long)H * D;
    const unsigned short* base = qkv + (long)b * T * rs + head * D;
    const unsigned short* dobase = d_o + (long)b * T * ors + head * D;
    const float* lse_b = lse + ((long)b * H + head) * T;
    const float* del_b = delta + ((long)b * H + head) * T;
    const float c_log2 = scale * 1.4426950408889634f;

    bf16x8 kf[4], vf[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        kf[kk] = *(const bf16x8*)(base + (long)key * rs + (long)H * D + 16 * kk + 8 * h);
        vf[kk] = *(const bf16x8*)(base + (long)key * rs + 2L * H * D + 16 * kk + 8 * h);
    }
    floatx16 dKt[2], dVt[2];
    zero16(dKt[0]); zero16(dKt[1]); zero16(dVt[0]); zero16(dVt[1]);

    u32x4 pq[2], po[2];
    fetch_tile(base + (long)(k0 / TILE) * TILE * rs, rs, tid, pq);
    fetch_tile(dobase + (long)(k0 / TILE) * TILE * ors, ors, tid, po);
    for (int jq = k0 / TILE; jq < T / TILE; ++jq) {
        __syncthreads();
        stash_tile(Qs, tid, pq);
        stash_tile(Os, tid, po);
        if (tid < TILE) { lse_s[tid] = lse_b[jq * TILE + tid] * 1.4426950408889634f; del_s[tid] = del_b[jq * TILE + tid]; }
        __syncthreads();
        if (jq + 1 < T / TILE) {
            fetch_tile(base + (long)(jq + 1) * TILE * rs, rs, tid, pq);
            fetch_tile(dobase + (long)(jq + 1) * TILE * ors, ors, tid, po);
        }
        if (jq * TILE + TILE - 1 < kw0) continue;                 // wave-uniform: all queries are before the wave's keys
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
            const int qt0 = jq * TILE + 32 * t2;                  // first query of this 32-row half tile
            if (qt0 + 31 < kw0) continue;                         // wave-uniform: still entirely before the wave's keys
            floatx16 S, dP;
            zero16(S); zero16(dP);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(Qs, 32 * t2 + kl, kk, h), kf[kk], S, 0, 0, 0);
                dP = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(Os, 32 * t2 + kl, kk, h), vf[kk], dP, 0, 0, 0);
            }
            // rows = queries (registers), column = this lane's key
            if (qt0 < kw0 + 31) {                                 // diagonal half tile: causal mask
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int qr = 32 * t2 + acc_row(r, h);
                    const float p = (jq * TILE + qr >= key) ? fast_exp2(S[r] * c_log2 - lse_s[qr]) : 0.f;
                    S[r] = p;
                    dP[r] = p * (dP[r] - del_s[qr]) * scale;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int qr = 32 * t2 + acc_row(r, h);
                    const float p = fast_exp2(S[r] * c_log2 - lse_s[qr]);
                    S[r] = p;
                    dP[r] = p * (dP[r] - del_s[qr]) * scale;
                }
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bf16x8 pb = pack_regs(S, s), dsb = pack_regs(dP, s);
                const int rowbase = 32 * t2 + 16 * s + 4 * h;
#pragma unroll
                for (int di = 0; di < 2; ++di) {
                    dVt[di] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(Os, rowbase, 32 * di, lane), pb, dVt[di], 0, 0, 0);
                    dKt[di] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(Qs, rowbase, 32 * di, lane), dsb, dKt[di], 0, 0, 0);
                }
            }
        }
    }
    unsigned short* dst = dqkv + (long)(b * T + key) * rs + head * D;
    store_T(dst + (long)H * D, dKt, 1.f, h);
    store_T(dst + 2L * H * D, dVt, 1.f, h);
}

// ------------------------------------------------------------------------------ dQ (query-stationary)
__global__ __launch_bounds__(256, 2) void attn_bwd_q_kernel(const unsigned short* __restrict__ qkv, const unsigned short* __restrict__ d_o,
                                                         const float* __restrict__ lse, const float* __restrict__ delta,
                                                         unsigned short* __restrict__ dqkv, int T, int H, float scale) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * TILE_BYTES];
    unsigned char* Ks = smem;
    unsigned char* Vs = smem + TILE_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ql = lane & 31, h = lane >> 5;
    const int head = blockIdx.y, b = blockIdx.z;
    const int q0 = (gridDim.x - 1 - blockIdx.x) * 128, qw0 = q0 + 32 * wave, q = qw0 + ql;
    const long rs = 3L * H * D, ors = (long)H * D;
    const unsigned short* base = qkv + (long)b * T * rs + head * D;
    const unsigned short* kbase = base + (long)H * D;
    const unsigned short* vbase = base + 2L * H * D;
    const float c_log2 = scale * 1.4426950408889634f;

    bf16x8 qf[4], dof[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        qf[kk] = *(const bf16x8*)(base + (long)q * rs + 16 * kk + 8 * h);
        dof[kk] = *(const bf16x8*)(d_o + ((long)(b * T + q) * H + head) * D + 16 * kk + 8 * h);
    }
    const float lse_q = lse[((long)b * H + head) * T + q] * 1.4426950408889634f;
    const float del_q = delta[((long)b * H + head) * T + q];
    floatx16 dQt[2];
    zero16(dQt[0]); zero16(dQt[1]);

    const int ntiles = (q0 + 128) / TILE;
    u32x4 pk[2], pv[2];
    fetch_tile(kbase, rs, tid, pk);
    fetch_tile(vbase, rs, tid, pv);
    for (int j = 0; j < ntiles; ++j) {
        __syncthreads();
        stash_tile(Ks, tid, pk);
        stash_tile(Vs, tid, pv);
        __syncthreads();
        if (j + 1 < ntiles) {
            fetch_tile(kbase + (long)(j + 1) * TILE * rs, rs, tid, pk);
            fetch_tile(vbase + (long)(j + 1) * TILE * rs, rs, tid, pv);
        }
        if (j * TILE > qw0 + 31) continue;
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
            const int kt0 = j * TILE + 32 * t2;                   // first key of this 32-row half tile
            if (kt0 > qw0 + 31) continue;                         // wave-uniform: entirely in the wave's future
            floatx16 St, dPt;
            zero16(St); zero16(dPt);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                St = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(Ks, 32 * t2 + ql, kk, h), qf[kk], St, 0, 0, 0);
                dPt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag(Vs, 32 * t2 + ql, kk, h), dof[kk], dPt, 0, 0, 0);
            }
            if (kt0 + 31 > qw0) {                                 // diagonal half tile: causal mask
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int kg = kt0 + acc_row(r, h);
                    const float p = (kg <= q) ? fast_exp2(St[r] * c_log2 - lse_q) : 0.f;
                    dPt[r] = p * (dPt[r] - del_q) * scale;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float p = fast_exp2(St[r] * c_log2 - lse_q);
                    dPt[r] = p * (dPt[r] - del_q) * scale;
                }
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bf16x8 dsb = pack_regs(dPt, s);
#pragma unroll
                for (int di = 0; di < 2; ++di)
                    dQt[di] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(Ks, 32 * t2 + 16 * s + 4 * h, 32 * di, lane), dsb,
                                                                      dQt[di], 0, 0, 0);
            }
        }
    }
    store_T(dqkv + (long)(b * T + q) * rs + head * D, dQt, 1.f, h);
}

}  // namespace

void attention_fwd(const void* qkv, void* out, float* lse, int B, int T, int H, int Dh, hipStream_t st) {
    (void)Dh;
    const float c_log2 = 0.125f * 1.4426950408889634f;   // 1/sqrt(64) * log2(e)
    hipLaunchKernelGGL(attn_fwd_kernel, dim3(T / 128, H, B), dim3(256), 0, st, (const unsigned short*)qkv, (unsigned short*)out,
                       lse, T, H, c_log2);
}

void attention_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, float* delta,
                   int B, int T, int H, int Dh, hipStream_t st) {
    (void)Dh;
    const long rows = (long)B * T * H;
    hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)((rows + 31) / 32)), dim3(256), 0, st, (const unsigned short*)out,
                       (const unsigned short*)dout, delta, T, H, rows);
    hipLaunchKernelGGL(attn_bwd_kv_kernel, dim3(T / 128, H, B), dim3(256), 0, st, (const unsigned short*)qkv,
                       (const unsigned short*)dout, lse, delta, (unsigned short*)dqkv, T, H, 0.125f);
    hipLaunchKernelGGL(attn_bwd_q_kernel, dim3(T / 128, H, B), dim3(256), 0, st, (const unsigned short*)qkv,
                       (const unsigned short*)dout, lse, delta, (unsigned short*)dqkv, T, H, 0.125f);
}

}  // namespace dpa
