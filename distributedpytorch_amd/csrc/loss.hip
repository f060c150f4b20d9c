// Fused softmax cross-entropy for gfx950.
//
// Forward: one pass over each logits row with an online (max, sum-exp) pair kept
// per lane, 16-B vector loads, wave shuffles + one LDS hop to fold the block.
// Emits the row's log-sum-exp and adds (lse - logit[target]) * scale into a scalar.
// Backward: dlogits = (exp(logit - lse) - onehot) * upstream * scale, one read and
// one write, may run in place over the logits buffer (the 50k-class GPT-2 head
// would otherwise hold two [tokens, vocab] tensors).
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

}  // namespace dpa
