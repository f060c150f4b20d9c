// Streaming (memory-bound) kernels for gfx950: pooling, layout conversion,
// residual add / ReLU / GELU, dtype casts, fused SGD-momentum / AdamW over flat
// parameter buffers.  Everything moves 16 B per lane (8 bf16 or 4 fp32) with
// grid-stride loops capped at a few thousand blocks.
#include "common.h"
#include "kernels.h"

namespace dpa {
namespace {

inline int grid_for(long n_items) {
    long b = (n_items + 255) / 256;
    if (b > 4096) b = 4096;
    if (b < 1) b = 1;
    return (int)b;
}

#define GRID_STRIDE(i, n) for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < (n); i += (long)gridDim.x * 256)

// ------------------------------------------------------------------ max pool 3x3 / stride 2 / pad 1 (NHWC)
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const unsigned short* __restrict__ x, unsigned short* __restrict__ y,
                                                          uint8_t* __restrict__ idx, int N, int H, int W, int C, int Ho, int Wo) {
    const int cpr = C / 8;
    const long total = (long)N * Ho * Wo * cpr;
    GRID_STRIDE(i, total) {
        const int ch = (int)(i % cpr);
        long p = i / cpr;
        const int wo = (int)(p % Wo); p /= Wo;
        const int ho = (int)(p % Ho);
        const int n = (int)(p / Ho);
        float best[8]; int bi[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { best[j] = -INFINITY; bi[j] = 0; }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int hi = ho * 2 - 1 + r;
            if ((unsigned)hi >= (unsigned)H) continue;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int wi = wo * 2 - 1 + s;
                if ((unsigned)wi >= (unsigned)W) continue;
                uint4 v = *(const uint4*)(x + (((long)n * H + hi) * W + wi) * C + ch * 8);
                float f[8]; unpack8(v, f);
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (f[j] > best[j] || f[j] != f[j]) { best[j] = f[j]; bi[j] = r * 3 + s; }
            }
        }
        *(uint4*)(y + i * 8) = pack8(best);
        uint2 pk;
        pk.x = bi[0] | (bi[1] << 8) | (bi[2] << 16) | (bi[3] << 24);
        pk.y = bi[4] | (bi[5] << 8) | (bi[6] << 16) | (bi[7] << 24);
        *(uint2*)(idx + i * 8) = pk;
    }
}

__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const unsigned short* __restrict__ dy, const uint8_t* __restrict__ idx,
                                                          unsigned short* __restrict__ dx, int N, int H, int W, int C, int Ho, int Wo) {
    // one thread per (input pixel, 8 channels): gather from the <= 4 windows that contain it
    const int cpr = C / 8;
    const long total = (long)N * H * W * cpr;
    GRID_STRIDE(i, total) {
        const int ch = (int)(i % cpr);
        long p = i / cpr;
        const int wi = (int)(p % W); p /= W;
        const int hi = (int)(p % H);
        const int n = (int)(p / H);
        float g[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        // windows: ho with ho*2-1 <= hi <= ho*2+1  ->  ho in [(hi-1+1)/2 .. (hi+1)/2]
        const int ho_lo = hi / 2, ho_hi = (hi + 1) / 2;
        const int wo_lo = wi / 2, wo_hi = (wi + 1) / 2;
        for (int ho = ho_lo; ho <= ho_hi; ++ho) {
            if (ho >= Ho) continue;
            const int r = hi - (ho * 2 - 1);
            for (int wo = wo_lo; wo <= wo_hi; ++wo) {
                if (wo >= Wo) continue;
                const int s = wi - (wo * 2 - 1);
                const int tap = r * 3 + s;
                const long o = (((long)n * Ho + ho) * Wo + wo) * cpr + ch;
                uint2 ik = *(const uint2*)(idx + o * 8);
                uint4 dv = *(const uint4*)(dy + o * 8);
                float d[8]; unpack8(dv, d);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const unsigned w32 = j < 4 ? ik.x : ik.y;
                    const int t = (w32 >> ((j & 3) * 8)) & 0xff;
                    if (t == tap) g[j] += d[j];
                }
            }
        }
        *(uint4*)(dx + i * 8) = pack8(g);
    }
}

// ------------------------------------------------------------------ global average pool (NHWC -> [N, C])
__global__ __launch_bounds__(256) void avgpool_fwd_kernel(const unsigned short* __restrict__ x, unsigned short* __restrict__ y,
                                                          int N, int HW, int C) {
    const int cpr = C / 8;
    const long total = (long)N * cpr;
    GRID_STRIDE(i, total) {
        const int ch = (int)(i % cpr);
        const long n = i / cpr;
        float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int p = 0; p < HW; ++p) {
            uint4 v = *(const uint4*)(x + ((n * HW + p) * C) + ch * 8);
            float f[8]; unpack8(v, f);
#pragma unroll
            for (int j = 0; j < 8; ++j) s[j] += f[j];
        }
        const float inv = 1.f / (float)HW;
#pragma unroll
        for (int j = 0; j < 8; ++j) s[j] *= inv;
        *(uint4*)(y + i * 8) = pack8(s);
    }
}

__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const unsigned short* __restrict__ dy, unsigned short* __restrict__ dx,
                                                          int N, int HW, int C) {
    const int cpr = C / 8;
    const long total = (long)N * HW * cpr;
    const float inv = 1.f / (float)HW;
    GRID_STRIDE(i, total) {
        const int ch = (int)(i % cpr);
        const long n = i / ((long)cpr * HW);
        uint4 v = *(const uint4*)(dy + (n * cpr + ch) * 8);
        float f[8]; unpack8(v, f);
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] *= inv;
        *(uint4*)(dx + i * 8) = pack8(f);
    }
}

// ------------------------------------------------------------------ NCHW -> NHWC(bf16, channel padded)
template <typename T>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const T* __restrict__ x, unsigned short* __restrict__ y,
                                                           int N, int C, int HW, int Cpad) {
    // one thread per output pixel: reads C strided planes (coalesced across threads), writes Cpad bf16
    const long total = (long)N * HW;
    GRID_STRIDE(i, total) {
        const long n = i / HW;
        const long p = i - n * HW;
        for (int c0 = 0; c0 < Cpad; c0 += 8) {
            float f[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int c = c0 + j;
                float v = 0.f;
                if (c < C) {
                    if constexpr (sizeof(T) == 4) v = x[(n * C + c) * HW + p];
                    else v = bf16_to_float(x[(n * C + c) * HW + p]);
                }
                f[j] = v;
            }
            *(uint4*)(y + i * Cpad + c0) = pack8(f);
        }
    }
}

// ------------------------------------------------------------------ elementwise
template <int OP, bool NT = false>  // 0 add, 1 add+relu, 2 relu_bwd(a=dy, b=y), 3 gelu_fwd(a), 4 gelu_bwd(a=dy, b=x)
__global__ __launch_bounds__(256) void ew_kernel(const unsigned short* __restrict__ a, const unsigned short* __restrict__ b,
                                                 unsigned short* __restrict__ y, long nchunks) {
    GRID_STRIDE(i, nchunks) {
        float fa[8], fb[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (NT) { const u32x4 w = __builtin_nontemporal_load((const u32x4*)(a + i * 8)); uint4 v{w[0], w[1], w[2], w[3]}; unpack8(v, fa); }
        else { uint4 v = *(const uint4*)(a + i * 8); unpack8(v, fa); }
        if (OP != 3) { uint4 v = *(const uint4*)(b + i * 8); unpack8(v, fb); }
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (OP == 0) o[j] = fa[j] + fb[j];
            else if (OP == 1) o[j] = fmaxf(fa[j] + fb[j], 0.f);
            else if (OP == 2) o[j] = fb[j] > 0.f ? fa[j] : 0.f;
            else if (OP == 3) o[j] = gelu_tanh(fa[j]);
            else o[j] = fa[j] * gelu_tanh_grad(fb[j]);
        }
        *(uint4*)(y + i * 8) = pack8(o);
    }
}

template <int OP>
__global__ void ew_tail_kernel(const unsigned short* a, const unsigned short* b, unsigned short* y, long start, long n) {
    long i = start + (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float fa = bf16_to_float(a[i]), fb = (OP != 3) ? bf16_to_float(b[i]) : 0.f, o;
    if (OP == 0) o = fa + fb;
    else if (OP == 1) o = fmaxf(fa + fb, 0.f);
    else if (OP == 2) o = fb > 0.f ? fa : 0.f;
    else if (OP == 3) o = gelu_tanh(fa);
    else o = fa * gelu_tanh_grad(fb);
    y[i] = float_to_bf16(o);
}

template <int OP>
void launch_ew(const void* a, const void* b, void* y, long n, hipStream_t st) {
    const long nchunks = n / 8;
    static const bool nt = [] { const char* e = getenv("DPA_LN_NT"); return !(e && atoi(e) == 0); }();   // the GELU input is read again in backward only (see layernorm_bwd)
    if (nchunks > 0 && OP == 3 && nt)
        hipLaunchKernelGGL((ew_kernel<OP, true>), dim3(grid_for(nchunks)), dim3(256), 0, st, (const unsigned short*)a,
                           (const unsigned short*)b, (unsigned short*)y, nchunks);
    else if (nchunks > 0)
        hipLaunchKernelGGL((ew_kernel<OP>), dim3(grid_for(nchunks)), dim3(256), 0, st, (const unsigned short*)a,
                           (const unsigned short*)b, (unsigned short*)y, nchunks);
    if (n % 8)
        hipLaunchKernelGGL((ew_tail_kernel<OP>), dim3(1), dim3(64), 0, st, (const unsigned short*)a,
                           (const unsigned short*)b, (unsigned short*)y, nchunks * 8, n);
}

// ------------------------------------------------------------------ optimizers over flat buffers
typedef __attribute__((ext_vector_type(4))) float f32x4;
template <bool NT> __device__ __forceinline__ float4 ldf4(const float* p) {
    if (NT) { const f32x4 v = __builtin_nontemporal_load((const f32x4*)p); return make_float4(v[0], v[1], v[2], v[3]); }
    return *(const float4*)p;
}
template <bool NT> __device__ __forceinline__ void stf4(float* p, float a, float b, float c, float d) {
    if (NT) __builtin_nontemporal_store(f32x4{a, b, c, d}, (f32x4*)p);
    else *(float4*)p = make_float4(a, b, c, d);
}

// NT: the fp32 master weights, the momentum and the gradient buffer are streamed with non-temporal loads / stores (nothing
// reads them before the next optimizer step or the next backward); the bf16 shadow the next forward reads is stored normally
template <bool LP, bool NT>
__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                  unsigned short* __restrict__ plp, long n4, long n, float lr, float mu,
                                                  float wd_all, float gs_host, int zero_grad, int nesterov,
                                                  const long* __restrict__ seg_ends, const float* __restrict__ seg_wd, int nseg,
                                                  const float* __restrict__ gs_dev) {
    // gs_dev: optional device scalar multiplied into the gradient scale (the clip coefficient of clip_grad_norm:
    // clipping costs no pass of its own over the gradient buffer)
    const float gs = gs_dev != nullptr ? gs_host * gs_dev[0] : gs_host;
    GRID_STRIDE(i, n4) {
        // per-segment weight decay (conv / linear weights decay, norm affine + biases do not): binary
        // search the segment table; segments are 64-element aligned so a float4 never straddles one
        float wd = wd_all;
        if (nseg > 0) {
            const long e = i * 4;
            int lo = 0, hi = nseg - 1;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (seg_ends[mid] > e) hi = mid; else lo = mid + 1; }
            wd = seg_wd[lo];
        }
        float4 pv = ldf4<NT>(p + i * 4), gv = ldf4<NT>(g + i * 4);
        float4 mv = (mu != 0.f) ? ldf4<NT>(m + i * 4) : make_float4(0, 0, 0, 0);
        float pp[4] = {pv.x, pv.y, pv.z, pv.w}, gg[4] = {gv.x, gv.y, gv.z, gv.w}, mm[4] = {mv.x, mv.y, mv.z, mv.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float d = gg[j] * gs + wd * pp[j];
            if (mu != 0.f) {
                mm[j] = mu * mm[j] + d;
                d = nesterov ? d + mu * mm[j] : mm[j];
            }
            pp[j] -= lr * d;
        }
        stf4<NT>(p + i * 4, pp[0], pp[1], pp[2], pp[3]);
        if (mu != 0.f) stf4<NT>(m + i * 4, mm[0], mm[1], mm[2], mm[3]);
        if (zero_grad) stf4<NT>(g + i * 4, 0.f, 0.f, 0.f, 0.f);
        if (LP) {
            uint2 o; o.x = pack_bf16x2(pp[0], pp[1]); o.y = pack_bf16x2(pp[2], pp[3]);
            *(uint2*)(plp + i * 4) = o;
        }
    }
    // scalar tail (n % 4), handled by block 0
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const long i = n4 * 4 + threadIdx.x;
        const float wd = nseg > 0 ? seg_wd[nseg - 1] : wd_all;
        float d = g[i] * gs + wd * p[i];
        if (mu != 0.f) { float mm = mu * m[i] + d; m[i] = mm; d = nesterov ? d + mu * mm : mm; }
        const float pp = p[i] - lr * d;
        p[i] = pp;
        if (zero_grad) g[i] = 0.f;
        if (LP) plp[i] = float_to_bf16(pp);
    }
}

template <bool LP>
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, unsigned short* __restrict__ plp, long n,
                                                    float lr, float b1, float b2, float eps, float wd, float gs_host,
                                                    float bc1, float bc2, int zero_grad, const float* __restrict__ gs_dev) {
    const float gs = gs_dev != nullptr ? gs_host * gs_dev[0] : gs_host;
    GRID_STRIDE(i, n) {
        const float gg = g[i] * gs;
        const float mm = b1 * m[i] + (1.f - b1) * gg;
        const float vv = b2 * v[i] + (1.f - b2) * gg * gg;
        m[i] = mm; v[i] = vv;
        float pp = p[i];
        pp -= lr * wd * pp;
        pp -= lr * (mm / bc1) / (sqrtf(vv / bc2) + eps);
        p[i] = pp;
        if (zero_grad) g[i] = 0.f;
        if (LP) plp[i] = float_to_bf16(pp);
    }
}

__global__ __launch_bounds__(256) void cast_f2b_kernel(const float* __restrict__ x, unsigned short* __restrict__ y, long n) {
    GRID_STRIDE(i, n) y[i] = float_to_bf16(x[i]);
}
__global__ __launch_bounds__(256) void cast_b2f_kernel(const unsigned short* __restrict__ x, float* __restrict__ y, long n) {
    GRID_STRIDE(i, n) y[i] = bf16_to_float(x[i]);
}
__global__ __launch_bounds__(256) void scale_kernel(float* __restrict__ x, long n, float s) {
    GRID_STRIDE(i, n) x[i] *= s;
}
__global__ __launch_bounds__(256) void mul_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y, long n) {
    GRID_STRIDE(i, n) y[i] = a[i] * b[i];
}

// Philox-free counter hash RNG (splitmix64) for dropout masks: mask = (u >= p) / (1-p)
// `step` (optional): device-resident step counter, so a step captured in a hipGraph draws a fresh
// mask at every replay (a host-side counter would be frozen into the graph).
__global__ __launch_bounds__(256) void dropout_mask_kernel(float* __restrict__ mask, long n, float p, uint64_t seed, uint64_t offset,
                                                           const int64_t* __restrict__ step) {
    const float keep_scale = 1.f / (1.f - p);
    if (step != nullptr) offset += (uint64_t)step[0] * 2000003ull;
    GRID_STRIDE(i, n) mask[i] = dropout_keep(seed, offset, (uint64_t)i, p, keep_scale);
}

// ------------------------------------------------------------------ pair-packed stem (3-channel images)
// The 7x7 / stride 2 / pad 3 stem convolution on a 3-channel image padded to 8 channels spends 5/8 of its MFMA work and
// gather traffic on zeros.  Packed as PIXEL PAIRS - y[n][h][j][0:3] = pixel (h, 2j), y[n][h][j][3:6] = pixel (h, 2j+1),
// channels 6-7 zero - the same convolution is 7 x 4 taps over [N, H, W/2, 8] with stride (2, 1) and padding (3, 2):
// output column w covers input columns 2w-3 .. 2w+3 = pairs w-2 (odd half), w-1, w, w+1.  224 instead of 392 GEMM-K,
// half the input bytes.  The filter is re-packed to [K][7][4][8] (slot (sp, half) <- tap s = 2 sp + half - 1, none for
// s = -1) and its gradient unpacked back into the [K][7][7][8] layout the parameter has.
template <typename T>
__global__ __launch_bounds__(256) void nchw_to_pairs_kernel(const T* __restrict__ x, unsigned short* __restrict__ y, int N, int C,
                                                            int H, int W) {
    const int Wp = W / 2;
    const long total = (long)N * H * Wp, HW = (long)H * W;
    GRID_STRIDE(i, total) {
        const long n = i / ((long)H * Wp);
        const long r = i - n * (long)H * Wp;            // h * Wp + j
        const long h = r / Wp, j = r - h * Wp;
        float f[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int c = 0; c < C; ++c) {
            const T* p = x + (n * C + c) * HW + h * W + 2 * j;
            if constexpr (sizeof(T) == 4) { f[c] = p[0]; f[C + c] = p[1]; }
            else { f[c] = bf16_to_float(p[0]); f[C + c] = bf16_to_float(p[1]); }
        }
        *(uint4*)(y + i * 8) = pack8(f);
    }
}

// w [K][R][7][8] (3 real channels) -> wp [K][R][4][8]
__global__ __launch_bounds__(256) void stem_pack_filter_kernel(const unsigned short* __restrict__ w, unsigned short* __restrict__ wp,
                                                               int KR, int C) {
    const int total = KR * 4 * 8;
    GRID_STRIDE(i, total) {
        const int ch = (int)(i & 7), sp = (int)((i >> 3) & 3), kr = (int)(i >> 5);
        const int half = ch / C, c = ch - half * C, s = 2 * sp + half - 1;
        wp[i] = (half < 2 && s >= 0 && s < 7) ? w[(kr * 7 + s) * 8 + c] : (unsigned short)0;
    }
}

// dw [K][R][7][8] (fp32) += unpacked dwp [K][R][4][8]
__global__ __launch_bounds__(256) void stem_unpack_dw_kernel(const float* __restrict__ dwp, float* __restrict__ dw, int KR, int C) {
    const int total = KR * 7 * C;
    GRID_STRIDE(i, total) {
        const int c = (int)(i % C), s = (int)((i / C) % 7), kr = (int)(i / (7 * C));
        const int sp = (s + 1) >> 1, half = (s + 1) & 1;
        dw[(kr * 7 + s) * 8 + c] += dwp[(kr * 4 + sp) * 8 + half * C + c];
    }
}

// element dropout of a bf16 tensor with the mask drawn in line (no mask tensor), optionally added to a residual
__global__ __launch_bounds__(256) void dropout_bf16_kernel(const unsigned short* __restrict__ x, const unsigned short* __restrict__ addend,
                                                           unsigned short* __restrict__ y, long nchunks, float p, uint64_t seed,
                                                           uint64_t offset) {
    const float keep_scale = 1.f / (1.f - p);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nchunks; i += (long)gridDim.x * 256) {
        float f[8]; unpack8(*(const uint4*)(x + i * 8), f);
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] *= dropout_keep(seed, offset, (uint64_t)(i * 8 + j), p, keep_scale);
        if (addend != nullptr) {
            float a[8]; unpack8(*(const uint4*)(addend + i * 8), a);
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] += a[j];
        }
        *(uint4*)(y + i * 8) = pack8(f);
    }
}

}  // namespace

void maxpool3x3s2_fwd(const void* x, void* y, uint8_t* idx, int N, int H, int W, int C, int Ho, int Wo, hipStream_t st) {
    const long total = (long)N * Ho * Wo * (C / 8);
    hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, st, (const unsigned short*)x,
                       (unsigned short*)y, idx, N, H, W, C, Ho, Wo);
}
void maxpool3x3s2_bwd(const void* dy, const uint8_t* idx, void* dx, int N, int H, int W, int C, int Ho, int Wo, hipStream_t st) {
    const long total = (long)N * H * W * (C / 8);
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, st, (const unsigned short*)dy, idx,
                       (unsigned short*)dx, N, H, W, C, Ho, Wo);
}
void avgpool_fwd(const void* x, void* y, int N, int HW, int C, hipStream_t st) {
    hipLaunchKernelGGL(avgpool_fwd_kernel, dim3(grid_for((long)N * (C / 8))), dim3(256), 0, st, (const unsigned short*)x,
                       (unsigned short*)y, N, HW, C);
}
void avgpool_bwd(const void* dy, void* dx, int N, int HW, int C, hipStream_t st) {
    hipLaunchKernelGGL(avgpool_bwd_kernel, dim3(grid_for((long)N * HW * (C / 8))), dim3(256), 0, st, (const unsigned short*)dy,
                       (unsigned short*)dx, N, HW, C);
}
void nchw_to_nhwc_pad(const void* x, int x_is_bf16, void* y, int N, int C, int H, int W, int Cpad, hipStream_t st) {
    const long total = (long)N * H * W;
    if (x_is_bf16)
        hipLaunchKernelGGL((nchw_to_nhwc_kernel<unsigned short>), dim3(grid_for(total)), dim3(256), 0, st,
                           (const unsigned short*)x, (unsigned short*)y, N, C, H * W, Cpad);
    else
        hipLaunchKernelGGL((nchw_to_nhwc_kernel<float>), dim3(grid_for(total)), dim3(256), 0, st, (const float*)x,
                           (unsigned short*)y, N, C, H * W, Cpad);
}
void add_bf16(const void* a, const void* b, void* y, long n, hipStream_t st) { launch_ew<0>(a, b, y, n, st); }
void add_relu_fwd(const void* a, const void* b, void* y, long n, hipStream_t st) { launch_ew<1>(a, b, y, n, st); }
void relu_bwd(const void* dy, const void* yv, void* dx, long n, hipStream_t st) { launch_ew<2>(dy, yv, dx, n, st); }
void gelu_fwd(const void* x, void* y, long n, hipStream_t st) { launch_ew<3>(x, x, y, n, st); }
void gelu_bwd(const void* dy, const void* x, void* dx, long n, hipStream_t st) { launch_ew<4>(dy, x, dx, n, st); }

void sgd_momentum(float* p, float* g, float* m, void* p_lp, long n, float lr, float mu, float wd,
                  float grad_scale, int zero_grad, int nesterov, const long* seg_ends, const float* seg_wd, int nseg,
                  const float* grad_scale_dev, hipStream_t st) {
    const long n4 = n / 4;
    dim3 grid(grid_for(n4)), block(256);
    static const bool nt = [] { const char* e = getenv("DPA_SGD_NT"); return !(e && atoi(e) == 0); }();   // GPT-2 +0.5 % (740.4 / 742.5 vs 745.4 / 744.8k tok/s), ResNet-50 neutral; DPA_SGD_NT=0 turns it off
#define DPA_SGD(LP_, NT_, PLP_) hipLaunchKernelGGL((sgd_kernel<LP_, NT_>), grid, block, 0, st, p, g, m, (unsigned short*)(PLP_), n4, n, lr, mu, wd, \
                           grad_scale, zero_grad, nesterov, seg_ends, seg_wd, nseg, grad_scale_dev)
    if (p_lp) { if (nt) DPA_SGD(true, true, p_lp); else DPA_SGD(true, false, p_lp); }
    else      { if (nt) DPA_SGD(false, true, nullptr); else DPA_SGD(false, false, nullptr); }
#undef DPA_SGD
}
void adamw(float* p, float* g, float* m, float* v, void* p_lp, long n, float lr, float b1, float b2, float eps,
           float wd, float grad_scale, float bc1, float bc2, int zero_grad, const float* grad_scale_dev, hipStream_t st) {
    dim3 grid(grid_for(n)), block(256);
    if (p_lp)
        hipLaunchKernelGGL((adamw_kernel<true>), grid, block, 0, st, p, g, m, v, (unsigned short*)p_lp, n, lr, b1, b2, eps,
                           wd, grad_scale, bc1, bc2, zero_grad, grad_scale_dev);
    else
        hipLaunchKernelGGL((adamw_kernel<false>), grid, block, 0, st, p, g, m, v, (unsigned short*)nullptr, n, lr, b1, b2,
                           eps, wd, grad_scale, bc1, bc2, zero_grad, grad_scale_dev);
}
void cast_f32_to_bf16(const float* x, void* y, long n, hipStream_t st) {
    hipLaunchKernelGGL(cast_f2b_kernel, dim3(grid_for(n)), dim3(256), 0, st, x, (unsigned short*)y, n);
}
void cast_bf16_to_f32(const void* x, float* y, long n, hipStream_t st) {
    hipLaunchKernelGGL(cast_b2f_kernel, dim3(grid_for(n)), dim3(256), 0, st, (const unsigned short*)x, y, n);
}
void scale_f32(float* x, long n, float s, hipStream_t st) {
    hipLaunchKernelGGL(scale_kernel, dim3(grid_for(n)), dim3(256), 0, st, x, n, s);
}
void mul_f32(const float* a, const float* b, float* y, long n, hipStream_t st) {
    hipLaunchKernelGGL(mul_kernel, dim3(grid_for(n)), dim3(256), 0, st, a, b, y, n);
}
void dropout_mask(float* mask, long n, float p, uint64_t seed, uint64_t offset, const int64_t* step, hipStream_t st) {
    hipLaunchKernelGGL(dropout_mask_kernel, dim3(grid_for(n)), dim3(256), 0, st, mask, n, p, seed, offset, step);
}

// dst[i] += sum_k src[k * stride + i]   (N-way fp32 reduction of the direct all-reduce, 16 B per lane)
__global__ __launch_bounds__(256) void reduce_shards_kernel(float* __restrict__ dst, const float* __restrict__ src, int nsrc,
                                                            long stride, long n) {
    const long n4 = n / 4;
    const bool vec = (stride % 4 == 0) && ((uintptr_t)dst % 16 == 0) && ((uintptr_t)src % 16 == 0);
    if (vec) {
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
            float4 a = ((const float4*)dst)[i];
            for (int k = 0; k < nsrc; ++k) {
                const float4 b = *(const float4*)(src + (long)k * stride + i * 4);
                a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
            }
            ((float4*)dst)[i] = a;
        }
    }
    for (long i = (vec ? n4 * 4 : 0) + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        float a = dst[i];
        for (int k = 0; k < nsrc; ++k) a += src[(long)k * stride + i];
        dst[i] = a;
    }
}

void reduce_shards(float* dst, const float* src, int nsrc, long stride, long n, hipStream_t st) {
    hipLaunchKernelGGL(reduce_shards_kernel, dim3(grid_for((n + 3) / 4)), dim3(256), 0, st, dst, src, nsrc, stride, n);
}

// bf16 wire format of the direct all-reduce: dst[i] (fp32, this rank's own values) += sum_k bf16 src[k * stride + i], summed
// in fp32; the result is written back in fp32 (dst) and in bf16 (out: what goes back on the wire)
__global__ __launch_bounds__(256) void reduce_shards_bf16_kernel(float* __restrict__ dst, const unsigned short* __restrict__ src,
                                                                 int nsrc, long stride, long n, unsigned short* __restrict__ out) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        float a = dst[i];
        for (int k = 0; k < nsrc; ++k) a += bf16_to_float(src[(long)k * stride + i]);
        // the owner keeps the ROUNDED sum too: every peer receives bf16(a) and widens it, and data-parallel replicas
        // must hold bit-identical gradients (else their fp32 masters drift apart step by step)
        const unsigned short r = float_to_bf16(a);
        dst[i] = bf16_to_float(r);
        out[i] = r;
    }
}

void reduce_shards_bf16(float* dst, const void* src, int nsrc, long stride, long n, void* out, hipStream_t st) {
    hipLaunchKernelGGL(reduce_shards_bf16_kernel, dim3(grid_for(n)), dim3(256), 0, st, dst, (const unsigned short*)src, nsrc,
                       stride, n, (unsigned short*)out);
}

// dst[i] += (*g) * src[i]   (fp32; the upstream gradient stays on the device: no host read to test it against 1)
__global__ __launch_bounds__(256) void axpy_dev_kernel(float* __restrict__ dst, const float* __restrict__ src, long n,
                                                       const float* __restrict__ g) {
    const float a = *g;
    const long n4 = n / 4;
    const bool vec = ((uintptr_t)dst % 16 == 0) && ((uintptr_t)src % 16 == 0);
    if (vec) {
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
            float4 d = ((const float4*)dst)[i];
            const float4 v = ((const float4*)src)[i];
            d.x += a * v.x; d.y += a * v.y; d.z += a * v.z; d.w += a * v.w;
            ((float4*)dst)[i] = d;
        }
    }
    for (long i = (vec ? n4 * 4 : 0) + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
        dst[i] += a * src[i];
}

void axpy_dev(float* dst, const float* src, long n, const float* g, hipStream_t st) {
    hipLaunchKernelGGL(axpy_dev_kernel, dim3(grid_for((n + 3) / 4)), dim3(256), 0, st, dst, src, n, g);
}

__global__ void counter_add_kernel(int64_t* c, int64_t v) { if (threadIdx.x == 0 && blockIdx.x == 0) c[0] += v; }

void counter_add(int64_t* c, int64_t v, hipStream_t st) { hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(64), 0, st, c, v); }

void dropout_bf16(const void* x, const void* addend, void* y, long n, float p, uint64_t seed, uint64_t offset, hipStream_t st) {
    const long nchunks = n / 8;
    hipLaunchKernelGGL(dropout_bf16_kernel, dim3(grid_for(nchunks)), dim3(256), 0, st, (const unsigned short*)x,
                       (const unsigned short*)addend, (unsigned short*)y, nchunks, p, seed, offset);
}

void nchw_to_pairs(const void* x, int x_is_bf16, void* y, int N, int C, int H, int W, hipStream_t st) {
    const long total = (long)N * H * (W / 2);
    if (x_is_bf16)
        hipLaunchKernelGGL((nchw_to_pairs_kernel<unsigned short>), dim3(grid_for(total)), dim3(256), 0, st,
                           (const unsigned short*)x, (unsigned short*)y, N, C, H, W);
    else
        hipLaunchKernelGGL((nchw_to_pairs_kernel<float>), dim3(grid_for(total)), dim3(256), 0, st, (const float*)x,
                           (unsigned short*)y, N, C, H, W);
}
void stem_pack_filter(const void* w, void* wp, int KR, int C, hipStream_t st) {
    hipLaunchKernelGGL(stem_pack_filter_kernel, dim3(grid_for((long)KR * 32)), dim3(256), 0, st, (const unsigned short*)w,
                       (unsigned short*)wp, KR, C);
}
void stem_unpack_dw(const float* dwp, float* dw, int KR, int C, hipStream_t st) {
    hipLaunchKernelGGL(stem_unpack_dw_kernel, dim3(grid_for((long)KR * 7 * C)), dim3(256), 0, st, dwp, dw, KR, C);
}

}  // namespace dpa
