// Stand-alone probe: what a 2-read + 1-write bf16 streaming kernel (the shape of bn_apply with a residual) can reach on
// this chip, by unroll depth / grid size / cache policy.  hipcc --offload-arch=gfx950 -O3 scripts/stream_probe.hip -o stream_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <bool NT>
__device__ __forceinline__ u32x4 ld(const u32x4* p) {
    if (NT) return __builtin_nontemporal_load(p);
    return *p;
}
template <bool NT>
__device__ __forceinline__ void st(u32x4* p, u32x4 v) {
    if (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}

__device__ __forceinline__ u32x4 fma8(u32x4 a, u32x4 b, float s) {
    u32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float a0 = __builtin_bit_cast(float, a[j] << 16), a1 = __builtin_bit_cast(float, a[j] & 0xffff0000u);
        const float b0 = __builtin_bit_cast(float, b[j] << 16), b1 = __builtin_bit_cast(float, b[j] & 0xffff0000u);
        const float r0 = fmaxf(a0 * s + b0, 0.f), r1 = fmaxf(a1 * s + b1, 0.f);
        o[j] = (__builtin_bit_cast(unsigned, r0) >> 16) | (__builtin_bit_cast(unsigned, r1) & 0xffff0000u);
    }
    return o;
}

// U chunks per thread per iteration, all loads issued before the first use; block-contiguous (thread t takes chunks
// base + u * 256 + t, so every wave instruction is 1 KiB contiguous)
template <int U, bool NTL, bool NTS, int READS>
__global__ __launch_bounds__(256) void stream_kernel(const u32x4* __restrict__ a, const u32x4* __restrict__ b, u32x4* __restrict__ y,
                                                     long nchunks, float s) {
    const long stride = (long)gridDim.x * 256 * U;
    for (long base = (long)blockIdx.x * 256 * U + threadIdx.x; base < nchunks; base += stride) {
        u32x4 va[U], vb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long i = base + u * 256;
            if (i < nchunks) {
                va[u] = ld<NTL>(a + i);
                if (READS == 2) vb[u] = ld<NTL>(b + i); else vb[u] = va[u];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long i = base + u * 256;
            if (i < nchunks) st<NTS>(y + i, fma8(va[u], vb[u], s));
        }
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int U, bool NTL, bool NTS, int READS>
float run(const u32x4* a, const u32x4* b, u32x4* y, long nchunks, int grid, int iters) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((stream_kernel<U, NTL, NTS, READS>), dim3(grid), dim3(256), 0, 0, a, b, y, nchunks, 1.01f);
    std::vector<float> ts;
    for (int i = 0; i < iters; ++i) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((stream_kernel<U, NTL, NTS, READS>), dim3(grid), dim3(256), 0, 0, a, b, y, nchunks, 1.01f);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2];
}

int main(int argc, char** argv) {
    const long mb = argc > 1 ? atol(argv[1]) : 822;
    const long bytes = mb * 1000 * 1000 / 4096 * 4096;
    const long nchunks = bytes / 16;
    u32x4 *a, *b, *y;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&y, bytes));
    CK(hipMemset(a, 0x3c, bytes)); CK(hipMemset(b, 0x3d, bytes)); CK(hipMemset(y, 0, bytes));
    CK(hipDeviceSynchronize());
    const int grids[] = {1024, 2048, 4096, 8192, 16384};
    printf("tensor %ld MB each; TB/s = (reads + 1 write) * bytes / time\n", mb);
#define ROW(U, NTL, NTS, READS)                                                                   \
    for (int g : grids) {                                                                         \
        const float ms = run<U, NTL, NTS, READS>(a, b, y, nchunks, g, 9);                         \
        printf("reads %d unroll %d ntl %d nts %d grid %5d: %8.1f us  %5.2f TB/s\n", READS, U, (int)NTL, (int)NTS, g, \
               ms * 1e3, (READS + 1) * (double)bytes / (ms * 1e-3) / 1e12);                       \
    }
    ROW(1, false, false, 2) ROW(2, false, false, 2) ROW(4, false, false, 2)
    ROW(1, true, false, 2) ROW(2, true, false, 2) ROW(4, true, false, 2)
    ROW(2, true, true, 2) ROW(4, true, true, 2)
    ROW(1, false, false, 1) ROW(4, false, false, 1) ROW(4, true, false, 1) ROW(4, true, true, 1)
    fflush(stdout);
    return 0;
}
