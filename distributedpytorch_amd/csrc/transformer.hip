// LayerNorm, embedding and partial-reduction kernels for the GPT-2 path (gfx950).
// Memory-bound: one 64-lane wave per row, 16 B (8 bf16) per lane, wave shuffles
// for the row reductions; parameter gradients are reduced per block in LDS and
// written as partial rows (no atomics), then folded by reduce_partials.
#include <cstdlib>
#include "common.h"
#include "kernels.h"

namespace dpa {
namespace {

// NCH = 16-byte chunks per lane (1, 2 or 4: C <= 512, 1024, 2048): the register arrays are sized by it, so the
// common C = 768 case runs at 4+ waves/SIMD instead of the 2 a fixed 4-chunk layout allows.
template <int NCH>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const unsigned short* __restrict__ x, const float* __restrict__ g,
                                                            const float* __restrict__ b, unsigned short* __restrict__ y,
                                                            float* __restrict__ mean, float* __restrict__ rstd, long rows,
                                                            int C, float eps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nch = C / 8;
    for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
        const unsigned short* xr = x + row * C;
        float v[NCH][8];
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int ch = lane + k * 64;
            if (ch < nch) {
                uint4 u = *(const uint4*)(xr + ch * 8);
                unpack8(u, v[k]);
#pragma unroll
                for (int j = 0; j < 8; ++j) { s += v[k][j]; q += v[k][j] * v[k][j]; }
            }
        }
        s = wave_reduce_sum(s); q = wave_reduce_sum(q);
        const float m = s / (float)C;
        float var = q / (float)C - m * m;
        var = var < 0.f ? 0.f : var;
        const float rs = rsqrtf(var + eps);
        if (lane == 0) { mean[row] = m; rstd[row] = rs; }
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int ch = lane + k * 64;
            if (ch < nch) {
                float o[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (v[k][j] - m) * rs * g[ch * 8 + j] + b[ch * 8 + j];
                *(uint4*)(y + row * C + ch * 8) = pack8(o);
            }
        }
    }
}

// One wave per row, LN_BWD_WAVES waves per workgroup, ONE partial (dgamma, dbeta) row per workgroup.  The row loop is a
// chain load -> two wave reductions -> store, i.e. one memory round trip per row per wave: with 4 waves on a CU and
// no prefetch it ran at 42 us for 8192 x 768 (HBM time: 10 us).  Eight waves per workgroup and the NEXT row's loads
// issued before the current row is reduced keep four times as many rows in flight without adding partial rows to fold.
template <bool NT>
__device__ __forceinline__ uint4 ln_ld16(const unsigned short* p) {
    if (NT) {
        const u32x4 v = __builtin_nontemporal_load((const u32x4*)p);
        return uint4{v[0], v[1], v[2], v[3]};
    }
    return *(const uint4*)p;
}

template <int NCH, int LN_BWD_WAVES, bool NT = false>
__global__ __launch_bounds__(LN_BWD_WAVES * 64) void layernorm_bwd_kernel(const unsigned short* __restrict__ dy, const unsigned short* __restrict__ x,
                                                            const float* __restrict__ g, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, unsigned short* __restrict__ dx,
                                                            float* __restrict__ dg_part, float* __restrict__ db_part,
                                                            long rows, int C, const unsigned short* __restrict__ addend) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nch = C / 8;
    float dg[NCH][8], db[NCH][8];                    // (gamma copied to registers up front: measured, no difference)
#pragma unroll
    for (int k = 0; k < NCH; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j) { dg[k][j] = 0.f; db[k][j] = 0.f; }
    const long rstep = (long)gridDim.x * LN_BWD_WAVES;
    long row = (long)blockIdx.x * LN_BWD_WAVES + wave;
    uint4 ndy[NCH], nx[NCH], nad[NCH];
    float nm = 0.f, nrs = 0.f;
#define DPA_LN_FETCH(R)                                                                         \
    if ((R) < rows) {                                                                           \
        nm = mean[R]; nrs = rstd[R];                                                            \
        _Pragma("unroll") for (int k = 0; k < NCH; ++k) {                                       \
            const int ch = lane + k * 64;                                                       \
            if (ch < nch) {                                                                     \
                ndy[k] = ln_ld16<NT>(dy + (R) * C + ch * 8);                                    \
                nx[k] = ln_ld16<NT>(x + (R) * C + ch * 8);                                      \
                if (addend != nullptr) nad[k] = ln_ld16<NT>(addend + (R) * C + ch * 8);         \
            }                                                                                   \
        }                                                                                       \
    }
    DPA_LN_FETCH(row)
    for (; row < rows; row += rstep) {
        const float m = nm, rs = nrs;
        uint4 cdy[NCH], cx[NCH], cad[NCH];
#pragma unroll
        for (int k = 0; k < NCH; ++k) { cdy[k] = ndy[k]; cx[k] = nx[k]; cad[k] = nad[k]; }
        const long nrow = row + rstep;
        DPA_LN_FETCH(nrow)                              // in flight under this row's arithmetic and reductions
        float xh[NCH][8], gy[NCH][8];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int ch = lane + k * 64;
            if (ch < nch) {
                float d[8], xv[8];
                unpack8(cdy[k], d);
                unpack8(cx[k], xv);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float h = (xv[j] - m) * rs;
                    xh[k][j] = h;
                    dg[k][j] += d[j] * h; db[k][j] += d[j];
                    const float t = d[j] * g[ch * 8 + j];
                    gy[k][j] = t;
                    s1 += t; s2 += t * h;
                }
            }
        }
        s1 = wave_reduce_sum(s1) / (float)C; s2 = wave_reduce_sum(s2) / (float)C;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int ch = lane + k * 64;
            if (ch < nch) {
                float o[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = rs * (gy[k][j] - s1 - xh[k][j] * s2);
                if (addend != nullptr) {           // gradient arriving over the residual connection around this LayerNorm
                    float ad[8];
                    unpack8(cad[k], ad);
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] += ad[j];
                }
                *(uint4*)(dx + row * C + ch * 8) = pack8(o);
            }
        }
    }
#undef DPA_LN_FETCH
    // fold the waves of the block through LDS, then one partial row per block
    __shared__ float red[LN_BWD_WAVES][64 * 8 + 8];
    for (int pass = 0; pass < 2 * NCH; ++pass) {
        const int k = pass >> 1;
        const bool is_g = (pass & 1) == 0;
        const int ch = lane + k * 64;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 8; ++j) red[wave][lane * 8 + j] = is_g ? dg[k][j] : db[k][j];
        __syncthreads();
        if (wave == 0 && ch < nch) {
            float* outp = (is_g ? dg_part : db_part) + (long)blockIdx.x * C + ch * 8;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float t = 0.f;
#pragma unroll
                for (int w = 0; w < LN_BWD_WAVES; ++w) t += red[w][lane * 8 + j];
                outp[j] = t;
            }
        }
    }
}

// out[c] (+)= sum_r part[r][c]: a block folds 16 columns, its 16 row groups stride the partial rows.
// blockIdx.y picks one of two stacked problems (part + y * rows * C -> out_a / out_b): LayerNorm's dgamma and
// dbeta fold in one launch.
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ part, int rows, int C,
                                                              float* __restrict__ out_a, float* __restrict__ out_b, int accumulate) {
    __shared__ float red[16][17];
    const int cl = threadIdx.x & 15, rg = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    part += (long)blockIdx.y * rows * C;
    float* __restrict__ out = blockIdx.y == 0 ? out_a : out_b;
    float s = 0.f;
    if (c < C) {
#pragma unroll 8                                       // independent loads issued together, not one round trip per row
        for (int r = rg; r < rows; r += 16) s += part[(long)r * C + c];
    }
    red[rg][cl] = s;
    __syncthreads();
    if (rg == 0 && c < C) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += red[i][cl];
        out[c] = accumulate ? out[c] + t : t;
    }
}

__global__ __launch_bounds__(256) void embedding_fwd_kernel(const int64_t* __restrict__ tok, const unsigned short* __restrict__ wte,
                                                            const unsigned short* __restrict__ wpe, unsigned short* __restrict__ y,
                                                            long BT, int T, int C, long V) {
    const int nch = C / 8;
    const long total = BT * nch;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int ch = (int)(i % nch);
        const long row = i / nch;
        const int t = (int)(row % T);
        long id = tok[row];
        id = id < 0 ? 0 : (id >= V ? V - 1 : id);
        float a[8], p[8];
        { uint4 u = *(const uint4*)(wte + id * C + ch * 8); unpack8(u, a); }
        { uint4 u = *(const uint4*)(wpe + (long)t * C + ch * 8); unpack8(u, p); }
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] += p[j];
        *(uint4*)(y + i * 8) = pack8(a);
    }
}

__global__ __launch_bounds__(256) void embedding_bwd_tok_kernel(const int64_t* __restrict__ tok, const unsigned short* __restrict__ dy,
                                                                float* __restrict__ dwte, long BT, int C, long V) {
    // one wave-instruction adds 64 consecutive floats (256 contiguous bytes) of one row
    const long total = BT * C;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long row = i / C;
        const int c = (int)(i - row * C);
        long id = tok[row];
        id = id < 0 ? 0 : (id >= V ? V - 1 : id);
        atomicAdd(dwte + id * C + c, bf16_to_float(dy[i]));
    }
}

// DPA_DETERMINISTIC route (runtime.hip): one workgroup per vocabulary row walks the token list in order - a single writer
// per gradient row, contributions added in position order, no atomics.  ~2 ms at 8192 tokens x 50k rows: a debugging mode.
__global__ __launch_bounds__(256) void embedding_bwd_tok_ordered_kernel(const int64_t* __restrict__ tok, const unsigned short* __restrict__ dy,
                                                                        float* __restrict__ dwte, long BT, int C, long V) {
    const long v = blockIdx.x;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};              // C <= 2048: channels threadIdx.x + 256 k
    bool any = false;
    for (long r = 0; r < BT; ++r) {
        long id = tok[r];
        id = id < 0 ? 0 : (id >= V ? V - 1 : id);
        if (id != v) continue;                             // block-uniform
        any = true;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int c = threadIdx.x + 256 * k;
            if (c < C) acc[k] += bf16_to_float(dy[r * C + c]);
        }
    }
    if (!any) return;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int c = threadIdx.x + 256 * k;
        if (c < C) atomicAdd(dwte + v * C + c, acc[k]);   // ONE add per element from this kernel; atomic only because the tied
                                                          // LM head's weight gradient may be adding its own single term on the side stream
    }
}

__global__ __launch_bounds__(256) void embedding_bwd_pos_kernel(const unsigned short* __restrict__ dy, float* __restrict__ dwpe,
                                                                long B, int T, int C) {
    const int nch = C / 8;
    const long total = (long)T * nch;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int ch = (int)(i % nch);
        const long t = i / nch;
        float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (long b = 0; b < B; ++b) {
            uint4 u = *(const uint4*)(dy + ((b * T + t) * C) + ch * 8);
            float f[8]; unpack8(u, f);
#pragma unroll
            for (int j = 0; j < 8; ++j) s[j] += f[j];
        }
        float* o = dwpe + t * C + ch * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] += s[j];
    }
}

inline int grid_for(long n) { long b = (n + 255) / 256; if (b > 4096) b = 4096; if (b < 1) b = 1; return (int)b; }

}  // namespace

// 16 waves (2 rows each, everything requested at once) needs the 128-register budget of a 1024-thread workgroup and
// spills: 53 us against 15 us stand-alone at 8192 x 768 (scripts/bench_ln.py)
constexpr int ln_bwd_waves() { return 8; }

int layernorm_partial_rows(long rows) {
    long b = (rows + ln_bwd_waves() - 1) / ln_bwd_waves();
    // one workgroup per CU.  Measured in the GPT-2 step (8192 x 768, fold on the side stream), tok/s: 128 blocks 580k,
    // 256: 624.6k, 384: 617k, 512: 620k, 1024: 616k, 2048: 599k - every extra block is another partial row to fold
    static const long cap = [] { const char* e = getenv("DPA_LN_ROWS"); return e ? atol(e) : 256L; }();
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (int)b;
}

void layernorm_fwd(const void* x, const float* g, const float* b, void* y, float* mean, float* rstd,
                   long rows, int C, float eps, hipStream_t st) {
    long blocks = (rows + 3) / 4; if (blocks > 4096) blocks = 4096; if (blocks < 1) blocks = 1;
#define DPA_LN_FWD(NCH_) hipLaunchKernelGGL(layernorm_fwd_kernel<NCH_>, dim3((unsigned)blocks), dim3(256), 0, st, \
        (const unsigned short*)x, g, b, (unsigned short*)y, mean, rstd, rows, C, eps)
    if (C <= 512) DPA_LN_FWD(1); else if (C <= 1024) DPA_LN_FWD(2); else DPA_LN_FWD(4);
#undef DPA_LN_FWD
}

void layernorm_bwd(const void* dy, const void* x, const float* g, const float* mean, const float* rstd,
                   void* dx, float* dg_partial, float* db_partial, int partial_rows, long rows, int C, const void* addend,
                   hipStream_t st) {
    static const bool nt = [] { const char* e = getenv("DPA_LN_NT"); return !(e && atoi(e) == 0); }();   // last use of x / dy / addend: GPT-2 +0.6 % together with the GELU input (757.1 / 759.5 vs 763.6 / 762.7k tok/s)
#define DPA_LN_BWD_H(NCH_, W_, NT_) hipLaunchKernelGGL((layernorm_bwd_kernel<NCH_, W_, NT_>), dim3(partial_rows), dim3(W_ * 64), 0, st, \
        (const unsigned short*)dy, (const unsigned short*)x, g, mean, rstd, (unsigned short*)dx, dg_partial, db_partial, rows, C, \
        (const unsigned short*)addend)
#define DPA_LN_BWD(NCH_) if (nt) DPA_LN_BWD_H(NCH_, 8, true); else DPA_LN_BWD_H(NCH_, 8, false)
    if (C <= 512) DPA_LN_BWD(1); else if (C <= 1024) DPA_LN_BWD(2); else DPA_LN_BWD(4);
#undef DPA_LN_BWD
#undef DPA_LN_BWD_H
}

void reduce_partials(const float* partial, int rows, int C, float* out, int accumulate, hipStream_t st) {
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((C + 15) / 16, 1), dim3(256), 0, st, partial, rows, C, out, out, accumulate);
}

void reduce_partials2(const float* partial, int rows, int C, float* out_a, float* out_b, int accumulate, hipStream_t st) {
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((C + 15) / 16, 2), dim3(256), 0, st, partial, rows, C, out_a, out_b, accumulate);
}

void embedding_fwd(const int64_t* tok, const void* wte, const void* wpe, void* y, long B, int T, int C, long V, hipStream_t st) {
    hipLaunchKernelGGL(embedding_fwd_kernel, dim3(grid_for(B * T * (C / 8))), dim3(256), 0, st, tok, (const unsigned short*)wte,
                       (const unsigned short*)wpe, (unsigned short*)y, B * T, T, C, V);
}

void embedding_bwd(const int64_t* tok, const void* dy, float* dwte, float* dwpe, long B, int T, int C, long V, hipStream_t st) {
    if (deterministic() && C <= 2048)
        hipLaunchKernelGGL(embedding_bwd_tok_ordered_kernel, dim3((unsigned)V), dim3(256), 0, st, tok, (const unsigned short*)dy,
                           dwte, B * T, C, V);
    else
        hipLaunchKernelGGL(embedding_bwd_tok_kernel, dim3(grid_for(B * T * C)), dim3(256), 0, st, tok, (const unsigned short*)dy,
                           dwte, B * T, C, V);
    hipLaunchKernelGGL(embedding_bwd_pos_kernel, dim3(grid_for((long)T * (C / 8))), dim3(256), 0, st,
                       (const unsigned short*)dy, dwpe, B, T, C);
}

}  // namespace dpa
