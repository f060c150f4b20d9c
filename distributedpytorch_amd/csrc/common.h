// Shared device helpers for the gfx950 (MI355X, CDNA4) kernels.
// Wave = 64 lanes; MFMA bf16 tiles 16x16x32 / 32x32x16; LDS 160 KiB per CU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dpa {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short short4v;
typedef __attribute__((ext_vector_type(4))) float floatx4;
typedef __attribute__((ext_vector_type(16))) float floatx16;
// native 16-byte register vector: unlike HIP's uint4 struct it stays in VGPRs through selects
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

constexpr int kWave = 64;

__device__ __forceinline__ float bf16_to_float(unsigned short v) {
    return __builtin_bit_cast(float, ((unsigned int)v) << 16);
}

// Round-to-nearest-even via a plain cast: hipcc emits v_cvt_pk_bf16_f32, which
// keeps NaN a NaN (integer-trick rounding does not).
__device__ __forceinline__ unsigned short float_to_bf16(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(unsigned short, b);
}

__device__ __forceinline__ unsigned int pack_bf16x2(float lo, float hi) {
    return (unsigned int)float_to_bf16(lo) | ((unsigned int)float_to_bf16(hi) << 16);
}

// 8 bf16 packed in a uint4 <-> 8 floats
__device__ __forceinline__ void unpack8(const uint4& v, float (&f)[8]) {
    f[0] = bf16_to_float(v.x & 0xffff); f[1] = bf16_to_float(v.x >> 16);
    f[2] = bf16_to_float(v.y & 0xffff); f[3] = bf16_to_float(v.y >> 16);
    f[4] = bf16_to_float(v.z & 0xffff); f[5] = bf16_to_float(v.z >> 16);
    f[6] = bf16_to_float(v.w & 0xffff); f[7] = bf16_to_float(v.w >> 16);
}

__device__ __forceinline__ uint4 pack8(const float (&f)[8]) {
    uint4 v;
    v.x = pack_bf16x2(f[0], f[1]); v.y = pack_bf16x2(f[2], f[3]);
    v.z = pack_bf16x2(f[4], f[5]); v.w = pack_bf16x2(f[6], f[7]);
    return v;
}

__device__ __forceinline__ float wave_reduce_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__device__ __forceinline__ float wave_reduce_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

__device__ __forceinline__ float gelu_tanh(float x) {
    const float k0 = 0.7978845608028654f, k1 = 0.044715f;
    float u = k0 * (x + k1 * x * x * x);
    return 0.5f * x * (1.0f + tanhf(u));
}

__device__ __forceinline__ float gelu_tanh_grad(float x) {
    const float k0 = 0.7978845608028654f, k1 = 0.044715f;
    float x2 = x * x;
    float u = k0 * (x + k1 * x * x2);
    float t = tanhf(u);
    float du = k0 * (1.0f + 3.0f * k1 * x2);
    return 0.5f * (1.0f + t) + 0.5f * x * (1.0f - t * t) * du;
}

// 16 B per lane from a buffer descriptor straight into LDS (lane-linear destination from the wave-uniform `lds`).
// COUNTED = false: the compiler's builtin; it then treats every later LDS read as dependent on the DMA and
// drains vmcnt to 0 before each barrier.  COUNTED = true: the same instruction issued from inline assembly, which
// leaves the vmcnt bookkeeping to the kernel (counted waits, two tiles in flight).  Loads complete in issue
// order, so compiler-inserted waits for its own loads only ever become stricter by the ones it cannot see.
template <bool COUNTED>
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rs, u32x4 rs4, unsigned char* lds, unsigned voffset) {
    if (COUNTED) {
        typedef __attribute__((address_space(3))) unsigned char* lds_byte_ptr;
        const unsigned lds_addr = (unsigned)(unsigned long long)(lds_byte_ptr)lds;
        asm volatile("s_mov_b32 m0, %2\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds"
                     :: "v"(voffset), "s"(rs4), "s"(lds_addr) : "memory", "m0");
    } else {
        typedef __attribute__((address_space(3))) void* lds_void_ptr;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_ptr)lds, 16, voffset, 0, 0, 0);
    }
}

// Descriptor words of a raw buffer (what __builtin_amdgcn_make_buffer_rsrc(ptr, 0, bytes, 0x00020000) holds), for
// the inline-assembly loads.
__device__ __forceinline__ u32x4 buffer_words(const void* ptr, unsigned bytes) {
    const unsigned long long p = (unsigned long long)ptr;
    return u32x4{(unsigned)p, (unsigned)(p >> 32), bytes, 0x00020000u};
}

// s_waitcnt immediate that waits for vmcnt <= n only (gfx9 encoding: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[15:14])
constexpr int waitcnt_vm_only(int n) { return (n & 15) | ((n >> 4) << 14) | (7 << 4) | (15 << 8); }

// d/dx of the tanh-approximated GELU with tanh from one v_exp_f32 + one v_rcp_f32 (for GEMM epilogues, where the
// libm tanhf expansion would outweigh the tile's main loop); error << one bf16 ulp of the product it scales
__device__ __forceinline__ float gelu_tanh_grad_fast(float x) {
    const float k0 = 0.7978845608028654f, k1 = 0.044715f;
    const float x2 = x * x;
    const float u = k0 * (x + k1 * x * x2);
    const float e = __builtin_amdgcn_exp2f(fminf(u * 2.885390081777927f, 80.f));      // e^(2u), clamped (tanh -> 1)
    const float t = 1.f - 2.f * __builtin_amdgcn_rcpf(e + 1.f);
    return 0.5f * (1.f + t) + 0.5f * x * (1.f - t * t) * k0 * (1.f + 3.f * k1 * x2);
}

// tanh-approximated GELU with tanh from one v_exp_f32 + one v_rcp_f32 (GEMM epilogues)
__device__ __forceinline__ float gelu_tanh_fast(float x) {
    const float k0 = 0.7978845608028654f, k1 = 0.044715f;
    const float u = k0 * (x + k1 * x * x * x);
    const float e = __builtin_amdgcn_exp2f(fminf(u * 2.885390081777927f, 80.f));      // e^(2u), clamped (tanh -> 1)
    const float t = 1.f - 2.f * __builtin_amdgcn_rcpf(e + 1.f);
    return 0.5f * x * (1.f + t);
}

// Counter-hash RNG (splitmix64) for dropout: keep-or-drop decision of element i of the stream (seed, offset) - a pure
// function, so forward kernels draw their masks in line and nothing is stored.  Returns 0 or 1/(1-p).
__device__ __forceinline__ float dropout_keep(uint64_t seed, uint64_t offset, uint64_t i, float p, float keep_scale) {
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * (i + offset + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    const float u = (float)(z >> 40) * (1.0f / 16777216.0f);
    return u >= p ? keep_scale : 0.f;
}

inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

}  // namespace dpa
