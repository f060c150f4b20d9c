// Wide-tile bf16 GEMM for gfx950: 256 x 256 output tile, K step 64, 8 waves (512 threads), one
// workgroup per CU with 128 KiB of LDS, operands by LDS-DMA (buffer_load ... lds) kept in flight
// ACROSS barriers behind counted s_waitcnt vmcnt(N).
//
//     y[m][n] = sum_k x[m][k] * w[n][k]          (both operands K-contiguous, fp32 accumulation)
//
// This is the R = S = 1, stride 1 case of ConvArgs (kernels.h): every Linear layer (forward and
// input gradient), every 1x1 convolution.  conv_igemm() routes a launch here when the grid fills
// the chip with 256-wide tiles (wide_gemm_eligible below); the 128 x 128 kernel of conv_igemm.hip
// keeps the rest.  Why a second main loop: the 128^2 / BK=32 loop stages 64 FLOP per byte through
// L2 -> LDS and pays a barrier every 32-deep K step; it tops out near 0.7 PFLOP/s (docs/DESIGN.md
// section 6).  This loop stages 128 FLOP per byte and never drains the load queue:
//
//   * wave (wr, wc), wr = wave >> 2, wc = wave & 3, owns 128 output channels x 64 pixels as
//     8 x 4 MFMA tiles (v_mfma_f32_16x16x32_bf16; A operand = filter rows, B operand = pixel rows,
//     so a lane ends up with 4 consecutive channels of one pixel);
//   * a K step is FOUR phases of 16 MFMAs (one quadrant of the wave tile each); a phase is a LOAD
//     segment (the quadrant's fragment reads, two 1-KiB DMA pieces of a LATER K tile, a counted
//     vmcnt) and an MFMA segment, each closed by a raw s_barrier;
//   * waves 4-7 run ONE barrier behind waves 0-3, so on every SIMD one wave is in its MFMA segment
//     while its partner is in its load segment (the two waves of a SIMD share the matrix pipe, so
//     two MFMA segments side by side would gain nothing; matrix beside memory does);
//   * each K tile is cut into four 16-KiB regions (filter rows of quadrants 0-1 / 2-3, pixel rows
//     of quadrants 0,3 / 1,2), each read from LDS in exactly ONE phase (the fragments then live in
//     registers), so a region of the buffer can be re-filled long before the whole tile is done:
//     every DMA is issued 5-6 phases ahead of its first read and at least 2 phases after the last
//     read of what it overwrites; `s_waitcnt vmcnt(8)` at the end of every load segment retires
//     exactly the region the NEXT phase reads (8 = 2 pieces x the 4 regions issued since);
//   * LDS rows are 128 B (64 bf16); the 16-byte chunk index is XOR-ed with (row & 7): the
//     ds_read_b128 fragment reads are bank-conflict free.  The DMA's LDS image is lane-linear, so
//     the swizzle is applied to the SOURCE address (lane -> chunk (lane & 7) ^ (lane >> 3));
//   * XCD-aware, grouped tile order: a workgroup's id is first mapped so that the workgroups of one
//     XCD (id mod 8) own a contiguous range of tiles, then tiles are walked 8 pixel tiles x
//     4.. channel tiles at a time: the 32 workgroups running on an XCD share 8 + 4 operand panels
//     in its 4 MiB L2 instead of 1 + 32.
//   * epilogue: each wave stages its own 64 x 128 tile in a private LDS region (no workgroup
//     barrier) and writes 256 contiguous bytes per pixel row, 16 B per lane; bias / activation /
//     residual addend (+ bit mask) / GELU backward / activated twin are applied there.
#include <stdlib.h>
#include "common.h"
#include "kernels.h"

namespace dpa {

namespace {

constexpr int WBK = 64;                    // K step (bf16 elements)
constexpr int ROWB = WBK * 2;              // bytes per LDS tile row
constexpr int TILE_B = 256 * ROWB;         // one operand tile: 32 KiB
constexpr int BUF_B = 2 * TILE_B;          // one K tile: filter rows, then pixel rows
constexpr int EPI_ROWB = 128 * 2 + 16;     // staged output row of a wave: 128 channels + 16 B pad
constexpr int EPI_WAVE_B = 64 * EPI_ROWB;  // 17,408 B per wave
constexpr int WIDE_LDS = (2 * BUF_B > 8 * EPI_WAVE_B) ? 2 * BUF_B : 8 * EPI_WAVE_B;   // 139,264 B
constexpr unsigned OOB = 0x80000000u;      // byte offset past every buffer (operands are < 2^31 bytes)

typedef __attribute__((address_space(3))) unsigned char* lds_byte_ptr;

// one 1-KiB LDS-DMA piece: 16 B per lane from buffer offset `voff` to LDS address `lds_addr` + 16 * lane
__device__ __forceinline__ void dma_piece(u32x4 rs4, unsigned lds_addr, unsigned voff) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds"
                 :: "v"(voff), "s"(rs4), "s"(lds_addr) : "memory", "m0");
}

__global__ __launch_bounds__(512, 2) void gemm_wide_kernel(ConvArgs a, int tiles_m, int tiles_n, int M, int Kd) {
    extern __shared__ __attribute__((aligned(16))) unsigned char wsm[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int Nn = a.K;                                  // output channels = GEMM columns

    // ---- tile of this workgroup: XCD-contiguous id, then groups of 8 pixel tiles ----
    int lid;
    {
        const int nwg = gridDim.x, bid = blockIdx.x;
        const int q = nwg >> 3, r8 = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        lid = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + idx;
    }
    constexpr int GM = 8;
    const int width = GM * tiles_n, grp = lid / width, first = grp * GM;
    const int gsz = min(tiles_m - first, GM), inner = lid - grp * width;
    const int tm = first + inner % gsz, tn = inner / gsz;
    const int m0 = tm * 256, c0 = tn * 256;

    const unsigned xbytes = (unsigned)M * (unsigned)Kd * 2u, wbytes = (unsigned)Nn * (unsigned)Kd * 2u;
    const u32x4 xrs4 = buffer_words(a.x, xbytes), wrs4 = buffer_words(a.w, wbytes);
    const unsigned lds0 = (unsigned)(unsigned long long)(lds_byte_ptr)wsm;

    // ---- DMA plan of this wave: per region two pieces of 8 rows x 128 B ----
    // filter region X (0: channel tiles 0-3 of both wave rows, 1: tiles 4-7): 8-row group q = 2 * wave + u
    //   covers rows (q >> 3) * 128 + X * 64 + (q & 7) * 8 ..;  pixel region Y (0: pixel tiles 0-1 of every
    //   wave column, 1: tiles 2-3): rows (q >> 2) * 64 + Y * 32 + (q & 3) * 8 ..
    const int lrow = lane >> 3;
    const unsigned src_chunk = (unsigned)((lane & 7) ^ lrow) * 16u;       // source-side swizzle (see header)
    unsigned woff[2][2], xoff[2][2], wlds[2][2], xlds[2][2];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int q = wave * 2 + u;
            const int wrow = (q >> 3) * 128 + r * 64 + (q & 7) * 8;
            const int xrow = (q >> 2) * 64 + r * 32 + (q & 3) * 8;
            wlds[r][u] = lds0 + (unsigned)wrow * ROWB;
            xlds[r][u] = lds0 + TILE_B + (unsigned)xrow * ROWB;
            const int ch = c0 + wrow + lrow, px = m0 + xrow + lrow;
            woff[r][u] = ch < Nn ? (unsigned)ch * (unsigned)Kd * 2u + src_chunk : OOB;
            xoff[r][u] = px < M ? (unsigned)px * (unsigned)Kd * 2u + src_chunk : OOB;
        }
    const int nk = Kd / WBK;

#define WG_DMA(RS4, OFF, LDS, REG, KT)                                                            \
    {                                                                                             \
        const int kt_ = (KT);                                                                     \
        const bool ok_ = kt_ < nk;                                                                \
        const unsigned kb_ = (unsigned)kt_ * (unsigned)ROWB, bo_ = (unsigned)(kt_ & 1) * BUF_B;   \
        _Pragma("unroll") for (int u = 0; u < 2; ++u)                                             \
            dma_piece(RS4, LDS[REG][u] + bo_, ok_ ? OFF[REG][u] + kb_ : OOB);                     \
    }
#define WG_DMA_W(REG, KT) WG_DMA(wrs4, woff, wlds, REG, KT)
#define WG_DMA_X(REG, KT) WG_DMA(xrs4, xoff, xlds, REG, KT)

    // ---- fragment addresses: lane reads row frow of a 16-row tile, chunk (ks * 4 + kq) ^ (frow & 7) ----
    const int frow = lane & 15, kq = lane >> 4;
    const unsigned fo0 = (unsigned)frow * ROWB + (unsigned)((kq ^ (frow & 7)) << 4);
    const unsigned fo1 = (unsigned)frow * ROWB + (unsigned)(((4 + kq) ^ (frow & 7)) << 4);
    const unsigned char* const wfb = wsm + (wr * 128) * ROWB;
    const unsigned char* const xfb = wsm + TILE_B + (wc * 64) * ROWB;

    floatx4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
    bf16x8 wf[4][2], xa[2][2], xb[2][2];

#define WG_READ_W(I0)                                                                             \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                               \
        wf[i][0] = *(const bf16x8*)(wp_ + ((I0) + i) * (16 * ROWB) + fo0);                        \
        wf[i][1] = *(const bf16x8*)(wp_ + ((I0) + i) * (16 * ROWB) + fo1);                        \
    }
#define WG_READ_X(XF, J0)                                                                         \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                               \
        XF[j][0] = *(const bf16x8*)(xp_ + ((J0) + j) * (16 * ROWB) + fo0);                        \
        XF[j][1] = *(const bf16x8*)(xp_ + ((J0) + j) * (16 * ROWB) + fo1);                        \
    }
#define WG_MMA(I0, J0, XF)                                                                        \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                              \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                             \
            _Pragma("unroll") for (int j = 0; j < 2; ++j)                                         \
                acc[(I0) + i][(J0) + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i][ks], XF[j][ks], acc[(I0) + i][(J0) + j], 0, 0, 0);
    // end of a load segment: retire the region the next phase reads, meet, wait for this phase's fragments
#define WG_TO_MMA()                                                                               \
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                                              \
    __builtin_amdgcn_sched_barrier(0);                                                            \
    __builtin_amdgcn_s_barrier();                                                                 \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                            \
    __builtin_amdgcn_sched_barrier(0);                                                            \
    __builtin_amdgcn_s_setprio(1);
#define WG_TO_LOAD()                                                                              \
    __builtin_amdgcn_s_setprio(0);                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                            \
    __builtin_amdgcn_s_barrier();                                                                 \
    __builtin_amdgcn_sched_barrier(0);

    // ---- prologue: all of K tile 0, and the two regions of tile 1 that phases 2-3 of a tile issue ----
    WG_DMA_W(0, 0) WG_DMA_X(0, 0) WG_DMA_X(1, 0) WG_DMA_W(1, 0)
    WG_DMA_W(0, 1) WG_DMA_X(0, 1)
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");     // filter region 0 and pixel region 0 of tile 0 have landed
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();           // waves 4-7 run one barrier behind

    for (int t = 0; t < nk; ++t) {
        const unsigned char* const wp_ = wfb + (t & 1) * BUF_B;
        const unsigned char* const xp_ = xfb + (t & 1) * BUF_B;
        // phase 0: quadrant (channel tiles 0-3, pixel tiles 0-1)
        WG_READ_W(0) WG_READ_X(xa, 0)
        WG_DMA_X(1, t + 1)
        WG_TO_MMA()
        WG_MMA(0, 0, xa)
        WG_TO_LOAD()
        // phase 1: quadrant (0-3, 2-3)
        WG_READ_X(xb, 2)
        WG_DMA_W(1, t + 1)
        WG_TO_MMA()
        WG_MMA(0, 2, xb)
        WG_TO_LOAD()
        // phase 2: quadrant (4-7, 2-3)
        WG_READ_W(4)
        WG_DMA_W(0, t + 2)
        WG_TO_MMA()
        WG_MMA(4, 2, xb)
        WG_TO_LOAD()
        // phase 3: quadrant (4-7, 0-1); its operands are in registers already
        WG_DMA_X(0, t + 2)
        WG_TO_MMA()
        WG_MMA(4, 0, xa)
        WG_TO_LOAD()
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the tail pieces (zeros) must land before LDS is reused
    if (wr == 0) __builtin_amdgcn_s_barrier();           // waves 0-3 catch up: barrier counts are equal again
    __syncthreads();

#undef WG_DMA
#undef WG_DMA_W
#undef WG_DMA_X
#undef WG_READ_W
#undef WG_READ_X
#undef WG_MMA
#undef WG_TO_MMA
#undef WG_TO_LOAD

    // ---- epilogue: bias / activation in fp32, stage the wave's 64 x 128 bf16 tile in its own LDS region ----
    unsigned char* const cs = wsm + wave * EPI_WAVE_B;
    const int q4 = kq * 4;
    if (a.bias == nullptr && a.act == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                uint2 pk;
                pk.x = pack_bf16x2(acc[i][j][0], acc[i][j][1]);
                pk.y = pack_bf16x2(acc[i][j][2], acc[i][j][3]);
                *(uint2*)(cs + (j * 16 + frow) * EPI_ROWB + (i * 16 + q4) * 2) = pk;
            }
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int ch = c0 + wr * 128 + i * 16 + q4;
            float b[4] = {0.f, 0.f, 0.f, 0.f};
            if (a.bias != nullptr) {
#pragma unroll
                for (int r = 0; r < 4; ++r) b[r] = a.bias[min(ch + r, Nn - 1)];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float tv = acc[i][j][r] + b[r];
                    if (a.act == 1) tv = fmaxf(tv, 0.f);
                    else if (a.act == 2) tv = gelu_tanh(tv);
                    v[r] = tv;
                }
                uint2 pk;
                pk.x = pack_bf16x2(v[0], v[1]);
                pk.y = pack_bf16x2(v[2], v[3]);
                *(uint2*)(cs + (j * 16 + frow) * EPI_ROWB + (i * 16 + q4) * 2) = pk;
            }
        }
    }
    // the region is private to the wave: LDS executes a wave's accesses in order, so no barrier - only
    // keep the compiler from moving the reads above the writes
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();

    // ---- write-out: lane = (row % 4, 16-byte chunk); 16 passes of 4 pixel rows x 256 B ----
    const int och = lane & 15, orow = lane >> 4;
    const int co = c0 + wr * 128 + och * 8;
    const bool cok = co < Nn;                               // Nn % 8 == 0 (host check)
    unsigned short* const yg = (unsigned short*)a.y;
    const bool gelu_bwd = a.mode == 1 && a.epi == 5;
    const void* pre_src = gelu_bwd ? a.ep_src : a.addend;
    const unsigned ybytes = (unsigned)M * (unsigned)Nn * 2u;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        unsigned eoff[8];
        u32x4 pre[8];
        unsigned pmk[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int px = m0 + wc * 64 + (half * 8 + s) * 4 + orow;
            eoff[s] = (px < M && cok) ? (unsigned)px * (unsigned)Nn + (unsigned)co : ~0u;
        }
        if (pre_src != nullptr) {
            const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc((void*)pre_src, 0, ybytes, 0x00020000);
#pragma unroll
            for (int s = 0; s < 8; ++s)
                pre[s] = __builtin_amdgcn_raw_buffer_load_b128(srs, eoff[s] != ~0u ? eoff[s] * 2u : 0xFFFFFFF0u, 0, 0);
            if (!gelu_bwd && a.addend_mask != nullptr) {
                const __amdgpu_buffer_rsrc_t mrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.addend_mask, 0, ybytes >> 4, 0x00020000);
#pragma unroll
                for (int s = 0; s < 8; ++s)
                    pmk[s] = __builtin_amdgcn_raw_buffer_load_b8(mrs, eoff[s] != ~0u ? eoff[s] >> 3 : 0xFFFFFFF0u, 0, 0);
            } else {
#pragma unroll
                for (int s = 0; s < 8; ++s) pmk[s] = 0xFFu;
            }
        }
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            if (eoff[s] == ~0u) continue;
            const int row = (half * 8 + s) * 4 + orow;
            uint4 v = *(const uint4*)(cs + row * EPI_ROWB + och * 16);
            if (gelu_bwd) {
                float f[8], g[8];
                unpack8(v, f); unpack8(__builtin_bit_cast(uint4, pre[s]), g);
#pragma unroll
                for (int j = 0; j < 8; ++j) f[j] *= gelu_tanh_grad_fast(g[j]);
                v = pack8(f);
            } else if (pre_src != nullptr) {
                float f[8], g[8];
                unpack8(v, f); unpack8(__builtin_bit_cast(uint4, pre[s]), g);
                const unsigned mk = pmk[s];
#pragma unroll
                for (int j = 0; j < 8; ++j) f[j] += ((mk >> j) & 1u) ? g[j] : 0.f;
                v = pack8(f);
            }
            *(uint4*)(yg + (size_t)eoff[s]) = v;
            if (a.mode == 0 && a.y_act != nullptr) {
                float f[8];
                unpack8(v, f);
#pragma unroll
                for (int j = 0; j < 8; ++j) f[j] = gelu_tanh(f[j]);
                *(uint4*)((unsigned short*)a.y_act + (size_t)eoff[s]) = pack8(f);
            }
        }
    }
}

int g_wide_min_tiles = -1;      // -1: not yet read from the environment

}  // namespace

// Smallest number of 256 x 256 tiles for which conv_igemm() prefers this kernel (DPA_WIDE_MIN_TILES; 0 = never).
void set_wide_gemm_min_tiles(int n) { g_wide_min_tiles = n < 0 ? 0 : n; }
int wide_gemm_min_tiles() {
    if (g_wide_min_tiles < 0) {
        const char* e = getenv("DPA_WIDE_MIN_TILES");
        g_wide_min_tiles = e ? atoi(e) : 300;
        if (g_wide_min_tiles < 0) g_wide_min_tiles = 0;
    }
    return g_wide_min_tiles;
}

// A launch the wide kernel can take: a plain GEMM (1x1, stride 1, no padding) with K a multiple of 64, one of the
// epilogues implemented above, operands addressable by 31-bit byte offsets.
bool wide_gemm_supported(const ConvArgs& a) {
    if (!(a.R == 1 && a.S == 1 && a.pad == 0 && a.stride == 1)) return false;
    if (a.mode != 0 && a.mode != 1) return false;
    if (a.C % WBK != 0 || a.K % 8 != 0 || a.y == nullptr) return false;
    if (a.stats != nullptr) return false;
    if (!(a.epi == 0 || (a.mode == 1 && a.epi == 5))) return false;
    const long M = (long)a.N * a.Ho * a.Wo;
    if (M * a.C * 2 >= (1l << 31) || (long)a.K * a.C * 2 >= (1l << 31) || M * a.K >= (1l << 31) - 16) return false;
    return true;
}

bool wide_gemm_eligible(const ConvArgs& a) {
    const int min_tiles = wide_gemm_min_tiles();
    if (min_tiles <= 0 || !wide_gemm_supported(a)) return false;
    const long M = (long)a.N * a.Ho * a.Wo;
    const long tiles = ((M + 255) / 256) * ((a.K + 255) / 256);
    return tiles >= min_tiles;
}

void gemm_wide(const ConvArgs& a, hipStream_t st) {
    if (!wide_gemm_supported(a)) {
        fprintf(stderr, "gemm_wide: unsupported launch (host dispatch error)\n");
        abort();
    }
    static bool allowed = false;
    if (!allowed) {
        const hipError_t err = hipFuncSetAttribute((const void*)gemm_wide_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, WIDE_LDS);
        if (err != hipSuccess) {
            fprintf(stderr, "gemm_wide: cannot allow %d bytes of LDS: %s\n", WIDE_LDS, hipGetErrorString(err));
            abort();
        }
        allowed = true;
    }
    const long M = (long)a.N * a.Ho * a.Wo;
    const int tiles_m = (int)((M + 255) / 256), tiles_n = (a.K + 255) / 256;
    hipLaunchKernelGGL(gemm_wide_kernel, dim3(tiles_m * tiles_n), dim3(512), WIDE_LDS, st, a, tiles_m, tiles_n, (int)M, a.C);
}

}  // namespace dpa
