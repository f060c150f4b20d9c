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
// (64 x 256, for 64-channel layers); 32 pixels per K step; register-staged double
// buffering with one barrier per step.
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

template <int WM, int WN>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradArgs a, int tiles_n, long P, int Kg, int steps_per_split) {
    constexpr int BM = WM * 64;           // out-channel rows of dw per block
    constexpr int BN = WN * 64;           // filter columns (r,s,c) per block
    constexpr int A_CPR = BM / 8;         // 16-B chunks per dy tile row
    constexpr int B_CPR = BN / 8;
    constexpr int A_LD = BKP * A_CPR / 256;  // loads per thread
    constexpr int B_LD = BKP * B_CPR / 256;
    constexpr int A_BYTES = BKP * BM * 2, B_BYTES = BKP * BN * 2;
    constexpr int STAGE = A_BYTES + B_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int tile_n = blockIdx.x % tiles_n, tile_m = blockIdx.x / tiles_n;
    const int k0 = tile_m * BM;   // first out channel
    const int n0 = tile_n * BN;   // first filter column

    const long p_begin = (long)blockIdx.y * steps_per_split * BKP;
    long p_end = p_begin + (long)steps_per_split * BKP;
    if (p_end > P) p_end = P;
    if (p_begin >= p_end) return;
    const int nsteps = (int)((p_end - p_begin + BKP - 1) / BKP);

    const unsigned short* __restrict__ xg = (const unsigned short*)a.x;
    const unsigned short* __restrict__ dyg = (const unsigned short*)a.dy;

    // ---- A (dy) loads: thread -> (pixel row, channel chunk) ----
    const int a_ch = tid % A_CPR, a_row0 = tid / A_CPR;
    constexpr int A_RSTEP = 256 / A_CPR;
    const bool a_cok = (k0 + a_ch * 8) < a.K;
    // ---- B (x gather) loads: thread's filter column is fixed for the whole loop ----
    const int b_ch = tid % B_CPR, b_row0 = tid / B_CPR;
    constexpr int B_RSTEP = 256 / B_CPR;
    const int col = n0 + b_ch * 8;
    const bool b_cok = col < Kg;
    int fr, fs, fc;
    {
        int cc = b_cok ? col : 0;
        int rs = cc / a.C;
        fc = cc - rs * a.C;
        fr = rs / a.S;
        fs = rs - fr * a.S;
    }
    // pixel state for each of this thread's B rows: (n, ho, wo)
    int pn[B_LD], pho[B_LD], pwo[B_LD];
    const int HoWo = a.Ho * a.Wo;
#pragma unroll
    for (int i = 0; i < B_LD; ++i) {
        long p = p_begin + b_row0 + i * B_RSTEP;
        int n = (int)(p / HoWo);
        int rem = (int)(p - (long)n * HoWo);
        pn[i] = n; pho[i] = rem / a.Wo; pwo[i] = rem - pho[i] * a.Wo;
    }

    u32x4 areg[A_LD], breg[B_LD];
    bool aok[A_LD], bok[B_LD];
    const u32x4 zero4 = {0u, 0u, 0u, 0u};

    // Branch-free loads (see conv_igemm.hip): always load, select zero afterwards.
    auto load_tile = [&](int st) {
        const long pb = p_begin + (long)st * BKP;
#pragma unroll
        for (int i = 0; i < A_LD; ++i) {
            const long p = pb + a_row0 + i * A_RSTEP;
            const bool ok = a_cok && p < p_end;
            const long off = ok ? (p * a.K + k0 + a_ch * 8) : 0;
            areg[i] = *(const u32x4*)(dyg + off);
            aok[i] = ok;
        }
#pragma unroll
        for (int i = 0; i < B_LD; ++i) {
            const long p = pb + b_row0 + i * B_RSTEP;
            const int hi = pho[i] * a.stride - a.pad + fr;
            const int wi = pwo[i] * a.stride - a.pad + fs;
            const bool ok = b_cok && p < p_end && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
            const long off = ok ? ((((long)pn[i] * a.H + hi) * a.W + wi) * a.C + fc) : 0;
            breg[i] = *(const u32x4*)(xg + off);
            bok[i] = ok;
            // advance this row's pixel by BKP
            pwo[i] += BKP;
            while (pwo[i] >= a.Wo) { pwo[i] -= a.Wo; if (++pho[i] == a.Ho) { pho[i] = 0; ++pn[i]; } }
        }
    };

    // swizzle: 64-byte granule index ^= (pixel row & 3)
    auto store_tile = [&](int buf) {
        unsigned char* as = smem + buf * STAGE;
        unsigned char* bs = as + A_BYTES;
#pragma unroll
        for (int i = 0; i < A_LD; ++i) {
            const int row = a_row0 + i * A_RSTEP;
            const int ch = (((a_ch >> 2) ^ swz_key<BM * 2>(row)) << 2) | (a_ch & 3);
            *(u32x4*)(as + row * (BM * 2) + ch * 16) = aok[i] ? areg[i] : zero4;
        }
#pragma unroll
        for (int i = 0; i < B_LD; ++i) {
            const int row = b_row0 + i * B_RSTEP;
            const int ch = (((b_ch >> 2) ^ swz_key<BN * 2>(row)) << 2) | (b_ch & 3);
            *(u32x4*)(bs + row * (BN * 2) + ch * 16) = bok[i] ? breg[i] : zero4;
        }
    };

    floatx16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // transposed-read lane roles
    const int gi = lane >> 4, li = lane & 15, tq = li >> 2, tp = li & 3;
    const int kofs = 8 * (gi >> 1) + tq;        // pixel row inside a 16-deep k slice (first read)
    const int cofs = (gi & 1) * 16 + 4 * tp;    // column offset inside a 32-wide tile

    load_tile(0);
    store_tile(0);
    __syncthreads();
    int cur = 0;
    for (int st = 0; st < nsteps; ++st) {
        if (st + 1 < nsteps) load_tile(st + 1);
        const unsigned char* as = smem + cur * STAGE;
        const unsigned char* bs = as + A_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 af[2], bf[2];
            const int r0 = kk * 16 + kofs, r1 = r0 + 4;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int c = wm * 64 + i * 32 + cofs;  // element column in the dy tile
                const int g = c >> 5, w = (c & 31) * 2;
                short4v lo = tr_read(as + r0 * (BM * 2) + ((g ^ swz_key<BM * 2>(r0)) << 6) + w);
                short4v hi = tr_read(as + r1 * (BM * 2) + ((g ^ swz_key<BM * 2>(r1)) << 6) + w);
                af[i] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int c = wn * 64 + j * 32 + cofs;
                const int g = c >> 5, w = (c & 31) * 2;
                short4v lo = tr_read(bs + r0 * (BN * 2) + ((g ^ swz_key<BN * 2>(r0)) << 6) + w);
                short4v hi = tr_read(bs + r1 * (BN * 2) + ((g ^ swz_key<BN * 2>(r1)) << 6) + w);
                bf[j] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        if (st + 1 < nsteps) store_tile(cur ^ 1);
        __syncthreads();
        cur ^= 1;
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
}

template <int WM, int WN>
void launch(const WgradArgs& a, hipStream_t st) {
    const long P = (long)a.N * a.Ho * a.Wo;
    const int Kg = a.R * a.S * a.C;
    const int tiles_m = (a.K + WM * 64 - 1) / (WM * 64);
    const int tiles_n = (Kg + WN * 64 - 1) / (WN * 64);
    const int tiles = tiles_m * tiles_n;
    const int steps = (int)((P + BKP - 1) / BKP);
    // ~2 workgroups per CU: every extra split adds a full fp32 copy of dw to the atomic traffic
    // (chip-wide float-atomic rate is ~1.3 TB/s), so do not over-split large filters.
    int split = 512 / tiles;
    if (split < 1) split = 1;
    int max_split = steps / 8;
    if (max_split < 1) max_split = 1;
    if (split > max_split) split = max_split;
    const int sps = (steps + split - 1) / split;
    split = (steps + sps - 1) / sps;
    dim3 grid(tiles, split), block(256);
    hipLaunchKernelGGL((conv_wgrad_kernel<WM, WN>), grid, block, 0, st, a, tiles_n, P, Kg, sps);
}

}  // namespace

void conv_wgrad(const WgradArgs& a, hipStream_t st) {
    if (a.K <= 64) launch<1, 4>(a, st);
    else launch<2, 2>(a, st);
}

}  // namespace dpa
