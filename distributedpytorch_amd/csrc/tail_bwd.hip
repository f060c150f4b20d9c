// Backward of "1x1 conv -> BatchNorm (-> ReLU mask)" in ONE kernel for gfx950: the BatchNorm-backward apply pass,
// the conv's filter gradient, the conv's input gradient and the reduction pass of the BatchNorm in front of the
// conv, for the write-streaming layers (stage 1 of a ResNet: 64 -> 256 channels on 800k pixels at batch 256).
//
//   g[p][co]   = cA[co] * (dout[p][co] under its ReLU bit) + cB[co] * craw[p][co] + cC[co]       (never stored)
//   dw[co][cm] += sum_p g[p][co] * xin[p][cm]                                                     (fp32 atomics)
//   dx[p][cm]  = sum_co g[p][co] * w[co][cm]
//   stats     += (sum_p dx' , sum_p dx' * sx)  per cm,  dx' = dx under smask                       (optional)
//
// Why one kernel: op by op, g is written once (2 B/element) and read twice (wgrad, dgrad) on top of the apply pass's
// own 4.1 B/element - 10.1 B per element of the widest tensor of the block.  Here dout, craw and the mask are read
// once (4.1 B) and g only ever exists as an LDS tile.  All three consumers are HBM-bound on these layers (the MFMA
// work of a 64-row tile is ~0.4 us against ~4.7 us of HBM time per CU), so the kernel is organised around the
// load stream: persistent workgroups walk contiguous 64-row tiles; the NEXT tile's global loads are issued into
// registers right after the current tile has been transformed and written to LDS, and fly during the MFMA phase.
// The filter (32 KiB) stays in LDS for the whole kernel: global loads in the compute phase would queue BEHIND the
// prefetch (vmcnt is in-order) and serialise the two.
//
// LDS images: g and the filter as plain rows with the 16-byte chunk index XORed by ((row&3)<<2 | (row>>2)&3) - one
// image serves the transposed reads (ds_read_b64_tr_b16: wgrad sums over pixels) and the row reads (dgrad sums over
// channels) without bank conflicts; xin as in conv_wgrad.hip (64-byte granules XORed by the row); the dx tile goes
// through LDS once more so that it leaves in whole 128-byte rows and meets sx / smask chunk for chunk.
#include <stdio.h>
#include <stdlib.h>
#include "common.h"
#include "kernels.h"

namespace dpa {
namespace {

typedef __attribute__((address_space(3))) short4v* lds_s4_ptr;

__device__ __forceinline__ short4v tr_read(const unsigned char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_ptr)(p));
}

// XOR key of the 16-byte chunk index.  High two bits = row & 3: the four rows of a transposed read land in four
// different 64-byte windows.  Low two bits = a permutation G of (row >> 2) & 3: the 32x32x16 row read needs ANY
// permutation (16 rows, one chunk); the 16x16x32 row read mixes two neighbouring chunks in one 16-lane LDS group
// (rows {0-3, 12-15} with chunk c, rows {4-11} with chunk c + 1), which needs G(0), G(3), G(1)^1, G(2)^1 distinct:
// G = (0, 3, 2, 1) = -x & 3.  With the identity (the guide's image) that read was 2-way: SQ_LDS_BANK_CONFLICT was 43 %
// of the stage-2 kernel's LDS cycles (profiles/pmc/tail_s2).
__device__ __forceinline__ int xor_key(int row) { return ((row & 3) << 2) | ((0 - (row >> 2)) & 3); }

constexpr int TB_R = 64;        // pixel rows per tile
constexpr int TB_CM = 64;       // conv input channels
constexpr int TB_CO = 256;      // conv output channels (BatchNorm channels)
constexpr int D_ROWB = TB_CO * 2, X_ROWB = TB_CM * 2;
constexpr int D_BYTES = TB_R * D_ROWB;      // 32 KiB  g tile
constexpr int X_BYTES = TB_R * X_ROWB;      //  8 KiB  xin tile
constexpr int O_BYTES = TB_R * X_ROWB;      //  8 KiB  dx tile
constexpr int W_BYTES = TB_CM * D_ROWB;     // 32 KiB  filter [cm][co]
constexpr int K_BYTES = 3 * TB_CO * 4;      //  3 KiB  cA, cB, cC
constexpr int TB_LDS = D_BYTES + X_BYTES + O_BYTES + W_BYTES + K_BYTES;

// byte offset of 16-byte chunk `ch` of row `row` in an image with 512-byte rows
__device__ __forceinline__ int d_off(int row, int ch) {
    return row * D_ROWB + ((ch & ~15) << 4) + (((ch & 15) ^ xor_key(row)) << 4);
}

// AUX: cache-policy bits of the operand loads (2 = nt: every operand here is streamed once; DPA_TAIL_NT)
template <bool STATS, int AUX = 0>
__global__ __launch_bounds__(256, 1) void tail_bwd_kernel(TailBwdArgs a, int steps_per_wg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char tsm[];
    unsigned char* const dimg = tsm;
    unsigned char* const ximg = tsm + D_BYTES;
    unsigned char* const oimg = ximg + X_BYTES;
    unsigned char* const wimg = oimg + O_BYTES;
    float* const coef = (float*)(wimg + W_BYTES);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long P = a.P;
    const long p_begin = (long)blockIdx.x * steps_per_wg * TB_R;
    long p_end = p_begin + (long)steps_per_wg * TB_R;
    if (p_end > P) p_end = P;
    if (p_begin >= p_end) return;
    const int nsteps = (int)((p_end - p_begin + TB_R - 1) / TB_R);

    // descriptors end at this workgroup's last row: rows past it read zeros (and are forced to g = 0 below)
    const unsigned dbytes = (unsigned)p_end * (unsigned)D_ROWB, xbytes = (unsigned)p_end * (unsigned)X_ROWB;
    const __amdgpu_buffer_rsrc_t dors = __builtin_amdgcn_make_buffer_rsrc((void*)a.dout, 0, dbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t crrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.craw, 0, dbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t kors = __builtin_amdgcn_make_buffer_rsrc((void*)a.kout, 0, a.kout ? dbytes >> 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t xirs = __builtin_amdgcn_make_buffer_rsrc((void*)a.xin, 0, xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t sxrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.sx, 0, STATS ? xbytes : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t smrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.smask, 0, STATS ? xbytes >> 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t dxrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.dx, 0, xbytes, 0x00020000);

    // ---- filter and coefficients into LDS, once ----
    {
        const uint4* wsrc = (const uint4*)a.wt;                 // [cm][co]: 64 rows x 32 chunks
#pragma unroll
        for (int i = 0; i < TB_CM * (TB_CO / 8) / 256; ++i) {
            const int q = tid + 256 * i, row = q >> 5, ch = q & 31;
            *(uint4*)(wimg + d_off(row, ch)) = wsrc[q];
        }
        for (int i = tid; i < TB_CO; i += 256) {
            coef[i] = a.cA[i]; coef[TB_CO + i] = a.cB[i]; coef[2 * TB_CO + i] = a.cC[i];
        }
    }

    // ---- staging roles ----
    const int d_ch = tid & 31, d_row0 = tid >> 5;               // 8 chunks: rows d_row0 + 8 i
    const int x_ch = tid & 7, x_row0 = tid >> 3;                // 2 chunks: rows x_row0 + 32 i
    unsigned d_goff = ((unsigned)(p_begin + d_row0) * (unsigned)TB_CO + (unsigned)d_ch * 8u) * 2u;
    unsigned x_goff = ((unsigned)(p_begin + x_row0) * (unsigned)TB_CM + (unsigned)x_ch * 8u) * 2u;
    int d_lds[8], x_lds[2], o_lds[2];
#pragma unroll
    for (int i = 0; i < 8; ++i) d_lds[i] = d_off(d_row0 + 8 * i, d_ch);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = x_row0 + 32 * i;
        x_lds[i] = row * X_ROWB + (((((x_ch >> 2) ^ ((row >> 1) & 1)) << 2) | (x_ch & 3)) << 4);
        o_lds[i] = row * X_ROWB + ((x_ch ^ (row & 7)) << 4);
    }

    u32x4 dreg[8], creg[8], xreg[2], sreg[2];
    unsigned mreg[8], smr[2];
#define TB_LOAD_HALF(H)                                                                                     \
    {                                                                                                       \
        _Pragma("unroll") for (int i = 4 * (H); i < 4 * (H) + 4; ++i) {                                     \
            const unsigned o = d_goff + (unsigned)(i * 8 * D_ROWB);                                         \
            dreg[i] = __builtin_amdgcn_raw_buffer_load_b128(dors, o, 0, AUX);                                 \
            creg[i] = __builtin_amdgcn_raw_buffer_load_b128(crrs, o, 0, AUX);                                 \
            mreg[i] = a.kout ? (unsigned)__builtin_amdgcn_raw_buffer_load_b8(kors, o >> 4, 0, AUX) : 0xffu;   \
        }                                                                                                   \
        {                                                                                                   \
            const unsigned o = x_goff + (unsigned)((H) * 32 * X_ROWB);                                      \
            xreg[H] = __builtin_amdgcn_raw_buffer_load_b128(xirs, o, 0, AUX);                                 \
            if (STATS) {                                                                                    \
                sreg[H] = __builtin_amdgcn_raw_buffer_load_b128(sxrs, o, 0, AUX);                             \
                smr[H] = (unsigned)__builtin_amdgcn_raw_buffer_load_b8(smrs, o >> 4, 0, AUX);                 \
            }                                                                                               \
        }                                                                                                   \
        if (H) { d_goff += (unsigned)(TB_R * D_ROWB); x_goff += (unsigned)(TB_R * X_ROWB); }                \
    }

    // ---- wgrad roles: wave = 64 out channels x all 64 in channels; transposed reads as in conv_wgrad.hip ----
    floatx16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int gi = lane >> 4, li = lane & 15, tq = li >> 2, tp = li & 3;
    const int kofs = 8 * (gi >> 1) + tq;               // pixel row inside a 16-deep slice (first read; second = +4)
    const int cofs = (gi & 1) * 16 + 4 * tp;           // column inside a 32-wide block
    int a_tr[2][2], b_tr[2][2];                        // [block][first / second read] byte offsets at slice 0
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int row = kofs + 4 * h;
            const int col = wave * 64 + i * 32 + cofs;
            a_tr[i][h] = d_off(row, col >> 3) + 8 * ((col >> 2) & 1);
            const int xc = i * 32 + cofs;              // (i doubles as the in-channel block j here)
            b_tr[i][h] = row * X_ROWB + ((((xc >> 5) ^ ((row >> 1) & 1))) << 6) + (xc & 31) * 2;
        }

    // ---- dgrad roles: wave = 32 pixel rows x 32 in channels, all 256 out channels ----
    const int rb = wave >> 1, cb = wave & 1, l31 = lane & 31, hh = lane >> 5;
    const int g_row = rb * 32 + l31, w_row = cb * 32 + l31;
    const int g_base = g_row * D_ROWB, g_x = (hh ^ xor_key(g_row)) << 4;
    const int w_base = w_row * D_ROWB, w_x = (hh ^ xor_key(w_row)) << 4;
    int o_st[4];                                        // dx tile stores: 4 consecutive channels (8 bytes) each
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) o_st[g4] = g_row * X_ROWB + (((cb * 4 + g4) ^ (g_row & 7)) << 4) + 8 * hh;

    float ssum[8], qsum[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { ssum[j] = 0.f; qsum[j] = 0.f; }

    TB_LOAD_HALF(0)
    TB_LOAD_HALF(1)
    __syncthreads();                                    // filter + coefficients visible
    for (int t = 0; t < nsteps; ++t) {
        const long row_base = p_begin + (long)t * TB_R;
        u32x4 sxe[2]; unsigned sme[2];
        // ---- BatchNorm-backward apply on the way into LDS ----
        {
            float kA[8], kB[8], kC[8];
            {
                const float4* c4 = (const float4*)(coef + d_ch * 8);
                const float4 a0 = c4[0], a1 = c4[1], b0 = c4[TB_CO / 4], b1 = c4[TB_CO / 4 + 1], c0 = c4[TB_CO / 2], c1 = c4[TB_CO / 2 + 1];
                kA[0] = a0.x; kA[1] = a0.y; kA[2] = a0.z; kA[3] = a0.w; kA[4] = a1.x; kA[5] = a1.y; kA[6] = a1.z; kA[7] = a1.w;
                kB[0] = b0.x; kB[1] = b0.y; kB[2] = b0.z; kB[3] = b0.w; kB[4] = b1.x; kB[5] = b1.y; kB[6] = b1.z; kB[7] = b1.w;
                kC[0] = c0.x; kC[1] = c0.y; kC[2] = c0.z; kC[3] = c0.w; kC[4] = c1.x; kC[5] = c1.y; kC[6] = c1.z; kC[7] = c1.w;
            }
            // the tile arrives in two halves (rows 0..31, 32..63), each re-issued for tile t + 1 as soon as its registers
            // are free: the load stream never drains while this workgroup transforms and multiplies
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int i = 4 * h; i < 4 * h + 4; ++i) {
                    float dv[8], cv[8], o[8];
                    unpack8(uint4{dreg[i].x, dreg[i].y, dreg[i].z, dreg[i].w}, dv);
                    unpack8(uint4{creg[i].x, creg[i].y, creg[i].z, creg[i].w}, cv);
                    const unsigned m = mreg[i];
                    const bool live = row_base + d_row0 + 8 * i < p_end;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float gd = ((m >> j) & 1u) ? dv[j] : 0.f;
                        const float v = __builtin_fmaf(kA[j], gd, __builtin_fmaf(kB[j], cv[j], kC[j]));   // as bn_bwd_apply_kernel
                        o[j] = live ? v : 0.f;
                    }
                    const uint4 pk = pack8(o);
                    *(u32x4*)(dimg + d_lds[i]) = u32x4{pk.x, pk.y, pk.z, pk.w};
                }
                *(u32x4*)(ximg + x_lds[h]) = xreg[h];
                if (STATS) { sxe[h] = sreg[h]; sme[h] = smr[h]; }
                __builtin_amdgcn_sched_barrier(0);
                if (h == 0) TB_LOAD_HALF(0) else TB_LOAD_HALF(1)      // past p_end: zeros, never used
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();

        // ---- wgrad: dw[co][cm] += g^T xin over the tile's 64 pixels ----
#pragma unroll
        for (int kk = 0; kk < TB_R / 16; ++kk) {
            bf16x8 af[2], bf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                short4v lo = tr_read(dimg + a_tr[i][0] + kk * 16 * D_ROWB);
                short4v hi = tr_read(dimg + a_tr[i][1] + kk * 16 * D_ROWB);
                af[i] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
                short4v xl = tr_read(ximg + b_tr[i][0] + kk * 16 * X_ROWB);
                short4v xh = tr_read(ximg + b_tr[i][1] + kk * 16 * X_ROWB);
                bf[i] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(xl, xh, 0, 1, 2, 3, 4, 5, 6, 7));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        // ---- dgrad, transposed: dxT[cm][p] = sum_co wt[cm][co] * g[p][co] ----
        floatx16 dacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) dacc[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < TB_CO / 16; ++ks) {
            const int c0 = ((2 * ks) & 15) << 4, half = (ks >> 3) << 8;
            const bf16x8 wf = *(const bf16x8*)(wimg + w_base + half + (c0 ^ w_x));
            const bf16x8 gf = *(const bf16x8*)(dimg + g_base + half + (c0 ^ g_x));
            dacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, gf, dacc, 0, 0, 0);
        }
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            uint2 v;
            v.x = pack_bf16x2(dacc[4 * g4 + 0], dacc[4 * g4 + 1]);
            v.y = pack_bf16x2(dacc[4 * g4 + 2], dacc[4 * g4 + 3]);
            *(uint2*)(oimg + o_st[g4]) = v;
        }
        __syncthreads();                                // dx tile complete; g / xin tiles free for the next transform
        // ---- dx out in whole rows; reduction pass of the BatchNorm behind it ----
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const u32x4 v = *(const u32x4*)(oimg + o_lds[i]);
            const long row = row_base + x_row0 + 32 * i;
            const unsigned off = row < p_end ? (unsigned)((row * TB_CM + x_ch * 8) * 2) : 0xFFFFFFF0u;
            __builtin_amdgcn_raw_buffer_store_b128(v, dxrs, off, 0, 0);
            if (STATS) {
                float g[8], xf[8];
                unpack8(uint4{v.x, v.y, v.z, v.w}, g);
                unpack8(uint4{sxe[i].x, sxe[i].y, sxe[i].z, sxe[i].w}, xf);
                const unsigned m = sme[i];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float gm = ((m >> j) & 1u) ? g[j] : 0.f;       // rows past p_end: mask and sx read as zero
                    ssum[j] += gm; qsum[j] += gm * xf[j];
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- dw: each accumulator register = 2 rows x 32 contiguous fp32 columns (full-rate atomics) ----
    {
        const int ccol = lane & 31, rhalf = (lane >> 5) * 4;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = wave * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + rhalf;
                    atomicAdd(a.dw + (long)row * TB_CM + j * 32 + ccol, acc[i][j][r]);
                }
    }
    if (STATS) {
        __syncthreads();
        float* red = (float*)dimg;                      // [256 threads][16]
#pragma unroll
        for (int j = 0; j < 8; ++j) { red[tid * 17 + j] = ssum[j]; red[tid * 17 + 8 + j] = qsum[j]; }
        __syncthreads();
        if (tid < 2 * TB_CM) {
            const int which = tid / TB_CM, c = tid % TB_CM, ch = c >> 3, j = c & 7;
            float s = 0.f;
            for (int r = 0; r < 32; ++r) s += red[(r * 8 + ch) * 17 + which * 8 + j];
            atomicAdd(a.stats + ((long)(blockIdx.x % a.stats_rows) * 2 + which) * TB_CM + c, s);
        }
    }
}
#undef TB_LOAD_HALF


// ------------------------------------------------------------------------------------------------------------------
// The same kernel for 128 -> 512 channels (stage 2 of a ResNet).  What changes: the filter is 128 KiB, so with it
// resident only a 16-pixel tile fits beside it (g 16 KiB + xin 4 + dx 4 + coefficients 6 = 158 KiB, one workgroup per
// CU); the filter gradient is 512 x 128 fp32 = 256 KiB of accumulators, i.e. 8 waves x 128 registers; the input
// gradient of a 16-pixel tile is 8 column blocks of v_mfma_f32_16x16x32_bf16, one per wave.  ONE tile (46 KB per CU)
// is prefetched: 204 us against 251 us for the three kernels, 2.8 TB/s - latency-bound, not bandwidth-bound.  A second
// register set (two tiles in flight) does not fit beside the accumulators: it spilled 53 VGPRs and ran at 301 us.
constexpr int T2_R = 16, T2_CM = 128, T2_CO = 512, T2_NT = 512;
constexpr int D2_ROWB = T2_CO * 2, X2_ROWB = T2_CM * 2;
constexpr int D2_BYTES = T2_R * D2_ROWB;     // 16 KiB
constexpr int X2_BYTES = T2_R * X2_ROWB;     //  4 KiB
constexpr int O2_BYTES = T2_R * X2_ROWB;     //  4 KiB
constexpr int W2_BYTES = T2_CM * D2_ROWB;    // 128 KiB
constexpr int K2_BYTES = 3 * T2_CO * 4;      //  6 KiB
constexpr int T2_LDS = D2_BYTES + X2_BYTES + O2_BYTES + W2_BYTES + K2_BYTES;
static_assert(T2_LDS <= 160 * 1024, "stage-2 tail kernel must fit one CU's LDS");

__device__ __forceinline__ int d2_off(int row, int ch) {
    return row * D2_ROWB + ((ch & ~15) << 4) + (((ch & 15) ^ xor_key(row)) << 4);
}

template <bool STATS, int AUX = 0>
__global__ __launch_bounds__(T2_NT) void tail_bwd_kernel_128(TailBwdArgs a, int steps_per_wg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char tsm[];
    unsigned char* const dimg = tsm;
    unsigned char* const ximg = tsm + D2_BYTES;
    unsigned char* const oimg = ximg + X2_BYTES;
    unsigned char* const wimg = oimg + O2_BYTES;
    float* const coef = (float*)(wimg + W2_BYTES);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long P = a.P;
    const long p_begin = (long)blockIdx.x * steps_per_wg * T2_R;
    long p_end = p_begin + (long)steps_per_wg * T2_R;
    if (p_end > P) p_end = P;
    if (p_begin >= p_end) return;
    const int nsteps = (int)((p_end - p_begin + T2_R - 1) / T2_R);

    const unsigned dbytes = (unsigned)p_end * (unsigned)D2_ROWB, xbytes = (unsigned)p_end * (unsigned)X2_ROWB;
    const __amdgpu_buffer_rsrc_t dors = __builtin_amdgcn_make_buffer_rsrc((void*)a.dout, 0, dbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t crrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.craw, 0, dbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t kors = __builtin_amdgcn_make_buffer_rsrc((void*)a.kout, 0, a.kout ? dbytes >> 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t xirs = __builtin_amdgcn_make_buffer_rsrc((void*)a.xin, 0, xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t sxrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.sx, 0, STATS ? xbytes : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t smrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.smask, 0, STATS ? xbytes >> 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t dxrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.dx, 0, xbytes, 0x00020000);

    {
        const uint4* wsrc = (const uint4*)a.wt;                 // [cm][co]: 128 rows x 64 chunks
#pragma unroll
        for (int i = 0; i < T2_CM * (T2_CO / 8) / T2_NT; ++i) {
            const int q = tid + T2_NT * i, row = q >> 6, ch = q & 63;
            *(uint4*)(wimg + d2_off(row, ch)) = wsrc[q];
        }
        for (int i = tid; i < T2_CO; i += T2_NT) {
            coef[i] = a.cA[i]; coef[T2_CO + i] = a.cB[i]; coef[2 * T2_CO + i] = a.cC[i];
        }
    }

    // ---- staging roles ----
    const int d_ch = tid & 63, d_row0 = tid >> 6;               // 2 chunks: rows d_row0 + 8 i
    const bool xrole = tid < 256;                               // wave-uniform: waves 0..3 also stage xin / finish dx
    const int x_ch = tid & 15, x_row = (tid >> 4) & 15;
    unsigned d_goff = ((unsigned)(p_begin + d_row0) * (unsigned)T2_CO + (unsigned)d_ch * 8u) * 2u;
    unsigned x_goff = ((unsigned)(p_begin + x_row) * (unsigned)T2_CM + (unsigned)x_ch * 8u) * 2u;
    const int d_lds0 = d2_off(d_row0, d_ch), d_lds1 = d2_off(d_row0 + 8, d_ch);
    const int x_lds = x_row * X2_ROWB + (((((x_ch >> 2) ^ (x_row & 3)) << 2) | (x_ch & 3)) << 4);
    const int o_lds = x_row * X2_ROWB + ((x_ch ^ (x_row & 7)) << 4);

    u32x4 dreg[2], creg[2], xreg, sreg;
    unsigned mreg[2], smr = 0;
#define T2_LOAD()                                                                                           \
    {                                                                                                       \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                     \
            const unsigned o = d_goff + (unsigned)(i * 8 * D2_ROWB);                                        \
            dreg[i] = __builtin_amdgcn_raw_buffer_load_b128(dors, o, 0, AUX);                                 \
            creg[i] = __builtin_amdgcn_raw_buffer_load_b128(crrs, o, 0, AUX);                                 \
            mreg[i] = a.kout ? (unsigned)__builtin_amdgcn_raw_buffer_load_b8(kors, o >> 4, 0, AUX) : 0xffu;   \
        }                                                                                                   \
        if (xrole) {                                                                                        \
            xreg = __builtin_amdgcn_raw_buffer_load_b128(xirs, x_goff, 0, AUX);                               \
            if (STATS) {                                                                                    \
                sreg = __builtin_amdgcn_raw_buffer_load_b128(sxrs, x_goff, 0, AUX);                           \
                smr = (unsigned)__builtin_amdgcn_raw_buffer_load_b8(smrs, x_goff >> 4, 0, AUX);               \
            }                                                                                               \
        }                                                                                                   \
        d_goff += (unsigned)(T2_R * D2_ROWB); x_goff += (unsigned)(T2_R * X2_ROWB);                         \
    }

    // ---- wgrad roles: wave = 64 out channels x all 128 in channels; one 16-pixel slice per tile ----
    floatx16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int gi = lane >> 4, li = lane & 15, tq = li >> 2, tp = li & 3;
    const int kofs = 8 * (gi >> 1) + tq;
    const int cofs = (gi & 1) * 16 + 4 * tp;
    int a_tr[2][2], b_tr[4][2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int row = kofs + 4 * h;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int col = wave * 64 + i * 32 + cofs;
            a_tr[i][h] = d2_off(row, col >> 3) + 8 * ((col >> 2) & 1);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int xc = j * 32 + cofs;
            b_tr[j][h] = row * X2_ROWB + (((xc >> 5) ^ (row & 3)) << 6) + (xc & 31) * 2;
        }
    }

    // ---- dgrad roles: wave = 16 in channels x the tile's 16 pixels, all 512 out channels (16x16x32 MFMAs) ----
    const int l15 = lane & 15, q4 = lane >> 4;
    const int w_row = wave * 16 + l15;
    const int w_base = w_row * D2_ROWB, w_x = (q4 ^ xor_key(w_row)) << 4;
    const int g_base = l15 * D2_ROWB, g_x = (q4 ^ xor_key(l15)) << 4;
    const int o_st = l15 * X2_ROWB + (((2 * wave + (q4 >> 1)) ^ (l15 & 7)) << 4) + 8 * (q4 & 1);

    float ssum[8], qsum[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { ssum[j] = 0.f; qsum[j] = 0.f; }

    T2_LOAD()
    __syncthreads();                                    // filter + coefficients visible
    for (int t = 0; t < nsteps; ++t) {
        const long row_base = p_begin + (long)t * T2_R;
        u32x4 sxe; unsigned sme = 0;
        {
            float kA[8], kB[8], kC[8];
            {
                const float4* c4 = (const float4*)(coef + d_ch * 8);
                const float4 a0 = c4[0], a1 = c4[1], b0 = c4[T2_CO / 4], b1 = c4[T2_CO / 4 + 1], c0 = c4[T2_CO / 2], c1 = c4[T2_CO / 2 + 1];
                kA[0] = a0.x; kA[1] = a0.y; kA[2] = a0.z; kA[3] = a0.w; kA[4] = a1.x; kA[5] = a1.y; kA[6] = a1.z; kA[7] = a1.w;
                kB[0] = b0.x; kB[1] = b0.y; kB[2] = b0.z; kB[3] = b0.w; kB[4] = b1.x; kB[5] = b1.y; kB[6] = b1.z; kB[7] = b1.w;
                kC[0] = c0.x; kC[1] = c0.y; kC[2] = c0.z; kC[3] = c0.w; kC[4] = c1.x; kC[5] = c1.y; kC[6] = c1.z; kC[7] = c1.w;
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                float dv[8], cv[8], o[8];
                unpack8(uint4{dreg[i].x, dreg[i].y, dreg[i].z, dreg[i].w}, dv);
                unpack8(uint4{creg[i].x, creg[i].y, creg[i].z, creg[i].w}, cv);
                const unsigned m = mreg[i];
                const bool live = row_base + d_row0 + 8 * i < p_end;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float gd = ((m >> j) & 1u) ? dv[j] : 0.f;
                    const float v = __builtin_fmaf(kA[j], gd, __builtin_fmaf(kB[j], cv[j], kC[j]));   // as bn_bwd_apply_kernel
                    o[j] = live ? v : 0.f;
                }
                const uint4 pk = pack8(o);
                *(u32x4*)(dimg + (i == 0 ? d_lds0 : d_lds1)) = u32x4{pk.x, pk.y, pk.z, pk.w};
            }
            if (xrole) {
                *(u32x4*)(ximg + x_lds) = xreg;
                if (STATS) { sxe = sreg; sme = smr; }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        T2_LOAD()                                       // tile t + 1 flies during the MFMA phase (past p_end: zeros)
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();

        // ---- wgrad: dw[co][cm] += g^T xin over the tile's 16 pixels ----
        {
            bf16x8 af[2], bf[4];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                short4v lo = tr_read(dimg + a_tr[i][0]);
                short4v hi = tr_read(dimg + a_tr[i][1]);
                af[i] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                short4v xl = tr_read(ximg + b_tr[j][0]);
                short4v xh = tr_read(ximg + b_tr[j][1]);
                bf[j] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(xl, xh, 0, 1, 2, 3, 4, 5, 6, 7));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        // ---- dgrad, transposed: dxT[cm][p] = sum_co wt[cm][co] * g[p][co] ----
        floatx4 dacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < T2_CO / 32; ++ks) {
            const int c0 = ((4 * ks) & 15) << 4, half = (ks >> 2) << 8;
            const bf16x8 wf = *(const bf16x8*)(wimg + w_base + half + (c0 ^ w_x));
            const bf16x8 gf = *(const bf16x8*)(dimg + g_base + half + (c0 ^ g_x));
            dacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, gf, dacc, 0, 0, 0);
        }
        {
            uint2 v;
            v.x = pack_bf16x2(dacc[0], dacc[1]);
            v.y = pack_bf16x2(dacc[2], dacc[3]);
            *(uint2*)(oimg + o_st) = v;
        }
        __syncthreads();                                // dx tile complete; g / xin tiles free for the next transform
        if (xrole) {
            const u32x4 v = *(const u32x4*)(oimg + o_lds);
            const long row = row_base + x_row;
            const unsigned off = row < p_end ? (unsigned)((row * T2_CM + x_ch * 8) * 2) : 0xFFFFFFF0u;
            __builtin_amdgcn_raw_buffer_store_b128(v, dxrs, off, 0, 0);
            if (STATS) {
                float g[8], xf[8];
                unpack8(uint4{v.x, v.y, v.z, v.w}, g);
                unpack8(uint4{sxe.x, sxe.y, sxe.z, sxe.w}, xf);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float gm = ((sme >> j) & 1u) ? g[j] : 0.f;     // rows past p_end: mask and sx read as zero
                    ssum[j] += gm; qsum[j] += gm * xf[j];
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    {
        const int ccol = lane & 31, rhalf = (lane >> 5) * 4;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = wave * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + rhalf;
                    atomicAdd(a.dw + (long)row * T2_CM + j * 32 + ccol, acc[i][j][r]);
                }
    }
    if (STATS) {
        __syncthreads();
        float* red = (float*)dimg;                      // [256 threads][17] = 17 KiB over the g + xin tiles
        if (xrole) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { red[tid * 17 + j] = ssum[j]; red[tid * 17 + 8 + j] = qsum[j]; }
        }
        __syncthreads();
        if (tid < 2 * T2_CM) {
            const int which = tid / T2_CM, c = tid % T2_CM, ch = c >> 3, j = c & 7;
            float s = 0.f;
            for (int r = 0; r < 16; ++r) s += red[(r * 16 + ch) * 17 + which * 8 + j];
            atomicAdd(a.stats + ((long)(blockIdx.x % a.stats_rows) * 2 + which) * T2_CM + c, s);
        }
    }
}
#undef T2_LOAD

}  // namespace

bool tail_bwd_supported(const TailBwdArgs& a) {
    static const bool off = [] { const char* e = getenv("DPA_NO_TAIL_BWD"); return e && atoi(e) != 0; }();
    static const bool off128 = [] { const char* e = getenv("DPA_NO_TAIL_BWD_128"); return e && atoi(e) != 0; }();
    const bool shape = (a.CM == TB_CM && a.CO == TB_CO) || (!off128 && a.CM == T2_CM && a.CO == T2_CO);
    return !off && !deterministic() && shape && a.P > 0 && a.P * (long)a.CO * 2 < (1ll << 31) - (1ll << 20);
}

template <class K>
static void allow_lds(K kernel, int bytes) {
    const hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) {
        fprintf(stderr, "tail_bwd: cannot allow %d bytes of LDS: %s\n", bytes, hipGetErrorString(e));
        abort();
    }
}

void tail_bwd(const TailBwdArgs& a, hipStream_t st) {
    static bool prepared = false;
    if (!prepared) {
        allow_lds(tail_bwd_kernel<true>, TB_LDS); allow_lds(tail_bwd_kernel<false>, TB_LDS);
        allow_lds(tail_bwd_kernel_128<true>, T2_LDS); allow_lds(tail_bwd_kernel_128<false>, T2_LDS);
        allow_lds(tail_bwd_kernel<true, 2>, TB_LDS); allow_lds(tail_bwd_kernel<false, 2>, TB_LDS);
        allow_lds(tail_bwd_kernel_128<true, 2>, T2_LDS); allow_lds(tail_bwd_kernel_128<false, 2>, T2_LDS);
        prepared = true;
    }
    // one workgroup per CU (83 / 158 KiB of LDS each): 254 us at 256 workgroups, 264 us at 512 on the stage-1 shape
    static const int target = [] { const char* e = getenv("DPA_TAIL_BWD_BLOCKS"); return e ? atoi(e) : 256; }();
    const bool wide = a.CM == T2_CM;
    const int R = wide ? T2_R : TB_R;
    const long steps = (a.P + R - 1) / R;
    long grid = steps < target ? steps : target;
    const int spw = (int)((steps + grid - 1) / grid);
    grid = (steps + spw - 1) / spw;
    static const bool nt = [] { const char* e = getenv("DPA_TAIL_NT"); return !(e && atoi(e) == 0); }();   // +0.3 % (14,782 / 14,831 vs 14,848 / 14,869 img/s at batch 512)
#define DPA_TAIL(KERNEL_, THREADS_, LDS_)                                                                        \
    if (a.stats != nullptr) {                                                                                    \
        if (nt) hipLaunchKernelGGL((KERNEL_<true, 2>), dim3((unsigned)grid), dim3(THREADS_), LDS_, st, a, spw);   \
        else    hipLaunchKernelGGL((KERNEL_<true, 0>), dim3((unsigned)grid), dim3(THREADS_), LDS_, st, a, spw);   \
    } else {                                                                                                     \
        if (nt) hipLaunchKernelGGL((KERNEL_<false, 2>), dim3((unsigned)grid), dim3(THREADS_), LDS_, st, a, spw);  \
        else    hipLaunchKernelGGL((KERNEL_<false, 0>), dim3((unsigned)grid), dim3(THREADS_), LDS_, st, a, spw);  \
    }
    if (wide) { DPA_TAIL(tail_bwd_kernel_128, T2_NT, T2_LDS) }
    else      { DPA_TAIL(tail_bwd_kernel, 256, TB_LDS) }
#undef DPA_TAIL
}

}  // namespace dpa
