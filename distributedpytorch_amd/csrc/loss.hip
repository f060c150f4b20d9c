// Fused softmax cross-entropy for gfx950.
//
// Forward: one pass over each logits row with an online (max, sum-exp) pair kept
// per lane, 16-B vector loads, wave shuffles + one LDS hop to fold the block.
// Emits the row's log-sum-exp and adds (lse - logit[target]) * scale into a scalar.
// Backward: dlogits = (exp(logit - lse) - onehot) * upstream * scale, one read and
// one write, may run in place over the logits buffer (the 50k-class GPT-2 head
// would otherwise hold two [tokens, vocab] tensors).
#include <cstdlib>
#include "common.h"
#include "kernels.h"

namespace dpa {
namespace {

template <typename T> struct RowIO;
template <> struct RowIO<unsigned short> {
    static constexpr int VEC = 8;
    __device__ static void load(const unsigned short* p, float (&f)[8]) { uint4 v = *(const uint4*)p; unpack8(v, f); }
    __device__ static void store(unsigned short* p, const float (&f)[8]) { *(uint4*)p = pack8(f); }
    __device__ static float load1(const unsigned short* p) { return bf16_to_float(*p); }
    __device__ static void store1(unsigned short* p, float v) { *p = float_to_bf16(v); }
};
template <> struct RowIO<float> {
    static constexpr int VEC = 8;
    __device__ static void load(const float* p, float (&f)[8]) {
        float4 a = *(const float4*)p, b = *(const float4*)(p + 4);
        f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
    }
    __device__ static void store(float* p, const float (&f)[8]) {
        *(float4*)p = make_float4(f[0], f[1], f[2], f[3]);
        *(float4*)(p + 4) = make_float4(f[4], f[5], f[6], f[7]);
    }
    __device__ static float load1(const float* p) { return *p; }
    __device__ static void store1(float* p, float v) { *p = v; }
};

constexpr float NEG_BIG = -1.0e30f;

__device__ __forceinline__ void online_merge(float& m, float& s, float m2, float s2) {
    // m starts at the finite sentinel NEG_BIG (never -inf): exp(NEG_BIG - x) == 0 and
    // NEG_BIG - NEG_BIG == 0, so empty lanes merge without producing inf - inf = NaN.
    const float mn = fmaxf(m, m2);
    s = s * __expf(m - mn) + s2 * __expf(m2 - mn);
    m = mn;
}

// one block (256 threads) per row; rows with V*sizeof(T) not 16-B aligned use the scalar path
template <typename T>
__global__ __launch_bounds__(256) void ce_fwd_kernel(const T* __restrict__ logits, const int64_t* __restrict__ targets,
                                                     float* __restrict__ lse_out, float* __restrict__ loss_sum,
                                                     int V, long ld, float scale, int64_t ignore_index, int vec_ok) {
    const long row = blockIdx.x;
    const T* x = logits + row * ld;
    float m = NEG_BIG, s = 0.f;
    if (vec_ok) {
        const int nv = V / 8;
        for (int i = threadIdx.x; i < nv; i += 256) {
            float f[8]; RowIO<T>::load(x + i * 8, f);
            float lm = f[0];
#pragma unroll
            for (int j = 1; j < 8; ++j) lm = fmaxf(lm, f[j]);
            float ls = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) ls += __expf(f[j] - lm);
            online_merge(m, s, lm, ls);
        }
        for (int i = nv * 8 + threadIdx.x; i < V; i += 256) online_merge(m, s, RowIO<T>::load1(x + i), 1.f);
    } else {
        for (int i = threadIdx.x; i < V; i += 256) online_merge(m, s, RowIO<T>::load1(x + i), 1.f);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float m2 = __shfl_xor(m, off, 64), s2 = __shfl_xor(s, off, 64);
        online_merge(m, s, m2, s2);
    }
    __shared__ float sm[4], ss[4];
    if ((threadIdx.x & 63) == 0) { sm[threadIdx.x >> 6] = m; ss[threadIdx.x >> 6] = s; }
    __syncthreads();
    if (threadIdx.x == 0) {
        m = sm[0]; s = ss[0];
        for (int w = 1; w < 4; ++w) online_merge(m, s, sm[w], ss[w]);
        const float lse = m + __logf(s);
        lse_out[row] = lse;
        const int64_t t = targets[row];
        if (t != ignore_index && t >= 0 && t < V) atomicAdd(loss_sum, (lse - RowIO<T>::load1(x + t)) * scale);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void ce_bwd_kernel(const T* __restrict__ logits, const int64_t* __restrict__ targets,
                                                     const float* __restrict__ lse_in, const float* __restrict__ gscale_ptr,
                                                     T* __restrict__ dlogits, int V, long ld, float scale,
                                                     int64_t ignore_index, int vec_ok) {
    const long row = blockIdx.x;
    const T* x = logits + row * ld;
    T* d = dlogits + row * ld;
    const int64_t t = targets[row];
    const bool live = (t != ignore_index && t >= 0 && t < V);
    const float gs = live ? (*gscale_ptr) * scale : 0.f;
    const float lse = lse_in[row];
    if (vec_ok) {
        const int nv = V / 8;
        for (int i = threadIdx.x; i < nv; i += 256) {
            float f[8]; RowIO<T>::load(x + i * 8, f);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float p = __expf(f[j] - lse);
                if (i * 8 + j == t) p -= 1.f;
                f[j] = p * gs;
            }
            RowIO<T>::store(d + i * 8, f);
        }
        for (int i = nv * 8 + threadIdx.x; i < V; i += 256) {
            float p = __expf(RowIO<T>::load1(x + i) - lse);
            if (i == t) p -= 1.f;
            RowIO<T>::store1(d + i, p * gs);
        }
    } else {
        for (int i = threadIdx.x; i < V; i += 256) {
            float p = __expf(RowIO<T>::load1(x + i) - lse);
            if (i == t) p -= 1.f;
            RowIO<T>::store1(d + i, p * gs);
        }
    }
    // padded columns [V, ld) carry no class: their gradient is zero
    for (long i = V + threadIdx.x; i < ld; i += 256) RowIO<T>::store1(d + i, 0.f);
}

// Forward AND gradient in one pass (bf16, ld % 8 == 0, ld <= 8 * 1024 * NCH): one 1024-thread workgroup per row holds the
// whole row in registers as packed bf16 (NCH 16-byte chunks per thread: 28 VGPRs for GPT-2's 50,304 classes), so the
// logits are read from memory ONCE and their gradient (softmax - onehot) * scale is written over them - 2 bytes read +
// 2 written per logit instead of 2 + (2 + 2) for the forward / backward pair.  Three sweeps over the registers (max;
// sum of exp(x - max); exp(x - lse)): two exps per element rather than 8 * NCH fp32 values kept per thread.  At 7 chunks
// the kernel takes 98 VGPRs, i.e. ONE workgroup per CU; forcing 64 (two workgroups, one's sweeps under the other's
// loads and stores) spills the row to scratch and loses: 658k vs 675k tok/s on GPT-2 (unfused pair: 665k).
// The gradient assumes an upstream gradient of 1; scale_unless_one_kernel below corrects it when that is not so.
template <int NCH>
__global__ __launch_bounds__(1024) void ce_fused_kernel(unsigned short* __restrict__ logits, const int64_t* __restrict__ targets,
                                                           float* __restrict__ loss_sum, int V, long ld, float scale,
                                                           int64_t ignore_index) {
    const long row = blockIdx.x;
    unsigned short* x = logits + row * ld;
    const int tid = threadIdx.x, nv = (int)(ld / 8);          // every chunk of the padded row; columns >= V count as -inf
    const int64_t t = targets[row];
    const bool live = (t != ignore_index && t >= 0 && t < V);
    float xt = 0.f;
    if (tid == 0 && live) xt = bf16_to_float(x[t]);          // before anything is overwritten
    uint4 raw[NCH];
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int i = tid + k * 1024;
        raw[k] = make_uint4(0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u);      // -inf: exp() = 0, gradient 0
        if (i < nv && i * 8 < V) {
            raw[k] = *(const uint4*)(x + (long)i * 8);
            if (i * 8 + 8 > V) {                              // the chunk that straddles V (GPT-2: 50,257 classes in 50,304 columns)
                unsigned* u = (unsigned*)&raw[k];
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (i * 8 + j >= V) u[j >> 1] = (j & 1) ? ((u[j >> 1] & 0x0000ffffu) | 0xff800000u) : ((u[j >> 1] & 0xffff0000u) | 0x0000ff80u);
            }
        }
    }
    __shared__ float red[16];
    __shared__ float bc;
    float m = NEG_BIG;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        float f[8]; unpack8(raw[k], f);
#pragma unroll
        for (int j = 0; j < 8; ++j) m = fmaxf(m, f[j]);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if ((tid & 63) == 0) red[tid >> 6] = m;
    __syncthreads();
    if (tid == 0) {
        float v = red[0];
#pragma unroll
        for (int w = 1; w < 16; ++w) v = fmaxf(v, red[w]);
        bc = v;
    }
    __syncthreads();
    m = bc;
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        float f[8]; unpack8(raw[k], f);
#pragma unroll
        for (int j = 0; j < 8; ++j) s += __expf(f[j] - m);    // padding chunks are -inf: exp = 0
    }
    s = wave_reduce_sum(s);
    __syncthreads();                                          // red / bc are reused
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) v += red[w];
        const float lse = m + __logf(v);
        bc = lse;
        if (live) atomicAdd(loss_sum, (lse - xt) * scale);
    }
    __syncthreads();
    const float lse = bc;
    const float gs = live ? scale : 0.f;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int i = tid + k * 1024;
        if (i < nv) {
            float f[8]; unpack8(raw[k], f);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float p = __expf(f[j] - lse);
                if ((int64_t)(i * 8 + j) == t) p -= 1.f;
                f[j] = p * gs;
            }
            *(uint4*)(x + (long)i * 8) = pack8(f);
        }
    }
}

// d *= g[0] unless g[0] == 1 (then the launch reads four bytes per workgroup and returns): the correction behind
// ce_fused_kernel for a caller that back-propagates something other than 1 into the loss, without a host-side look
// at a device scalar
__global__ __launch_bounds__(256) void scale_unless_one_kernel(unsigned short* __restrict__ d, const float* __restrict__ g, long nchunks) {
    const float gs = g[0];
    if (gs == 1.f) return;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nchunks; i += (long)gridDim.x * 256) {
        float f[8]; unpack8(*(const uint4*)(d + i * 8), f);
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] *= gs;
        *(uint4*)(d + i * 8) = pack8(f);
    }
}

}  // namespace

void softmax_ce_fwd(const void* logits, int is_bf16, const int64_t* targets, float* lse, float* loss_sum,
                    long B, int V, long ld, float scale, int64_t ignore_index, hipStream_t st) {
    if (B == 0) return;
    const int esz = is_bf16 ? 2 : 4;
    const int vec_ok = ((ld * esz) % 16 == 0) && (((uintptr_t)logits) % 16 == 0);
    if (is_bf16)
        hipLaunchKernelGGL((ce_fwd_kernel<unsigned short>), dim3((unsigned)B), dim3(256), 0, st, (const unsigned short*)logits,
                           targets, lse, loss_sum, V, ld, scale, ignore_index, vec_ok);
    else
        hipLaunchKernelGGL((ce_fwd_kernel<float>), dim3((unsigned)B), dim3(256), 0, st, (const float*)logits, targets, lse,
                           loss_sum, V, ld, scale, ignore_index, vec_ok);
}

void softmax_ce_bwd(const void* logits, int is_bf16, const int64_t* targets, const float* lse, const float* gscale_ptr,
                    void* dlogits, long B, int V, long ld, float scale, int64_t ignore_index, hipStream_t st) {
    if (B == 0) return;
    const int esz = is_bf16 ? 2 : 4;
    const int vec_ok = ((ld * esz) % 16 == 0) && (((uintptr_t)logits) % 16 == 0) && (((uintptr_t)dlogits) % 16 == 0);
    if (is_bf16)
        hipLaunchKernelGGL((ce_bwd_kernel<unsigned short>), dim3((unsigned)B), dim3(256), 0, st, (const unsigned short*)logits,
                           targets, lse, gscale_ptr, (unsigned short*)dlogits, V, ld, scale, ignore_index, vec_ok);
    else
        hipLaunchKernelGGL((ce_bwd_kernel<float>), dim3((unsigned)B), dim3(256), 0, st, (const float*)logits, targets, lse,
                           gscale_ptr, (float*)dlogits, V, ld, scale, ignore_index, vec_ok);
}

bool softmax_ce_fused_supported(int is_bf16, int V, long ld, const void* logits) {
    return is_bf16 && V >= 1 && V <= ld && ld % 8 == 0 && ld <= 8 * 1024 * 8 && ((uintptr_t)logits) % 16 == 0 && !deterministic();
}

// logits [B, ld] bf16 are OVERWRITTEN by (softmax - onehot) * scale; loss_sum (zeroed by the caller) receives the loss
void softmax_ce_fused(void* logits, const int64_t* targets, float* loss_sum, long B, int V, long ld, float scale,
                      int64_t ignore_index, hipStream_t st) {
    if (B == 0) return;
    const int nch = (int)((ld / 8 + 1023) / 1024);
#define DPA_CE_FUSED(N_) hipLaunchKernelGGL((ce_fused_kernel<N_>), dim3((unsigned)B), dim3(1024), 0, st, (unsigned short*)logits, \
                                            targets, loss_sum, V, ld, scale, ignore_index)
    switch (nch) {
        case 1: DPA_CE_FUSED(1); break;
        case 2: DPA_CE_FUSED(2); break;
        case 3: DPA_CE_FUSED(3); break;
        case 4: DPA_CE_FUSED(4); break;
        case 5: DPA_CE_FUSED(5); break;
        case 6: DPA_CE_FUSED(6); break;
        case 7: DPA_CE_FUSED(7); break;
        default: DPA_CE_FUSED(8); break;
    }
#undef DPA_CE_FUSED
}

void scale_unless_one(void* d_bf16, const float* g, long n, hipStream_t st) {
    const long nchunks = n / 8;                                // callers pass whole 16-byte chunks (ld % 8 == 0)
    long blocks = (nchunks + 255) / 256; if (blocks > 4096) blocks = 4096; if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(scale_unless_one_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (unsigned short*)d_bf16, g, nchunks);
}

}  // namespace dpa
