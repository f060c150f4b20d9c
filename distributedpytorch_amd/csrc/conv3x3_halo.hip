// Halo-tiled 3 x 3 / stride 1 / pad 1 convolution for 64 -> 64 channels (ResNet stage 1), forward and input gradient.
//
//   MODE 0:  y[n][h][w][k] = sum_{r,s,c} x[n][h+r-1][w+s-1][c] * w[k][r][s][c]
//   MODE 1:  the same sum with the taps mirrored (r, s) -> (2-r, 2-s) over the K/C-swapped filter twin: the input gradient
//
// conv_igemm.hip runs these layers as an implicit GEMM on 256 x 64 tiles: every input pixel is gathered into LDS nine
// times (once per tap, with per-lane bounds arithmetic for each), the 73 KB filter is re-staged by every one of 6,272
// workgroups, and a barrier sits between every two 32-deep K-steps: 557 / 587 TFLOP/s (212 / 202 us at batch 512) on
// layers whose own HBM traffic (read x, write y) is worth ~80 us.  Here:
//
//   * the workgroup is PERSISTENT (one per CU, 4 waves = one per SIMD) and walks tiles of 4 whole image rows;
//   * the FILTER LIVES IN REGISTERS: all 9 x 64 x 64 weights as MFMA A-fragments, 288 registers per lane - at one wave per
//     SIMD a wave owns the whole 512-entry register file, and MFMA operands may sit in the accumulation half.  It is
//     loaded once per workgroup and never touches LDS;
//   * each x element is staged ONCE per tile: the 6 rows around the tile arrive by LDS-DMA into rows of RS = 64 slots
//     (slot c of a row = pixel c - 1; slot 0 and slots 57.. stay zero: written once, never touched again), rows outside
//     the image are zero-filled by the buffer descriptor.  An output "slot" is the flat index into that image, so a tap
//     is nothing but an LDS address offset - no per-tap bounds arithmetic - and the reduction over the slot index also
//     computes 8 junk columns per row, which are dropped at the write-out (224 of 256 MFMA columns carry data);
//   * wave w owns output row w of the tile (64 slots) x all 64 output channels: 2 x 2 accumulators of 32 x 32, and per
//     16-deep K-step two B fragments (ds_read_b128; 16-byte chunks XOR-keyed by bits 1..3 of the slot: conflict-free) for
//     four MFMAs.  With a row stride of 64 slots a tap ROW leaves the key alone, so every read address is one of 12
//     per-lane registers (3 tap columns x 4 K-steps) plus an immediate;
//   * two patch buffers: the DMA of tile t + 1 flies under the 144 MFMAs of tile t; ONE barrier per tile;
//   * epilogue through a wave-private LDS image (no barrier): 16-byte stores over whole NHWC rows, and the BatchNorm
//     statistics of the layer behind it - forward: (sum, sum of squares) of the bf16 output; input gradient ("epi 4"):
//     (sum g', sum g' * c) under the ReLU bits of the BatchNorm in front, whose input row arrives by LDS-DMA with the
//     patch - kept in registers for ALL the tiles a workgroup walks: one atomic per channel and statistic per WORKGROUP
//     (256 x 128) instead of per 256-pixel tile (6,272 x 128).
#include <stdio.h>
#include <stdlib.h>
#include "common.h"
#include "kernels.h"

namespace dpa {
namespace {

constexpr int HW_ = 56;                       // image width this instantiation serves
constexpr int HTR = 4;                        // output rows per tile = waves per workgroup
constexpr int HRS = 64;                       // slots per stored row
constexpr int HLEAD = 8;                      // slack in front: tap (0, 0) of out slot 0 reads slot -1
constexpr int HPATCH_SLOTS = HLEAD + (HTR + 2) * HRS;          // junk-slot reads of the last row run 1 slot past: allocated LDS
constexpr int HPATCH_BYTES = HPATCH_SLOTS * 128;               // 50,176
constexpr int HSTAGE_BYTES = 4 * 8192;                         // epilogue image: 64 slots x 128 bytes per wave
constexpr int HCIMG_BYTES = 4 * 7168;                          // epi 4: one image row (56 pixels) of the BatchNorm input per wave
constexpr int HLDS_BYTES = 2 * HPATCH_BYTES + HSTAGE_BYTES + HCIMG_BYTES;   // 161,792
constexpr int HPIECES = (HTR + 2) * (HW_ / 8);                 // 8-pixel DMA pieces per tile
constexpr int HLPT = (HPIECES + 3) / 4;                        // DMA instructions per wave per tile
static_assert(HLDS_BYTES <= 160 * 1024, "LDS plan");

template <int MODE>
__global__ __launch_bounds__(256, 1) void conv3x3_c64_kernel(ConvArgs a, int tiles_total) {
    constexpr int W = HW_, RS = HRS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const stage = smem + 2 * HPATCH_BYTES;
    unsigned char* const cimg = stage + HSTAGE_BYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // XCD-aware split of the tile range: workgroups that share an XCD (id mod 8) walk neighbouring ranges, so the two
    // halo rows a tile shares with its neighbour are found in that XCD's L2
    int lid;
    {
        const int nwg = gridDim.x, bid = blockIdx.x;
        const int q = nwg >> 3, r8 = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        lid = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + idx;
    }
    const int per = (tiles_total + (int)gridDim.x - 1) / (int)gridDim.x;
    const int t_begin = lid * per;
    int t_end = t_begin + per;
    if (t_end > tiles_total) t_end = tiles_total;

    // zero both patch buffers once: pad slots, lead slots and slack are never written with data
    for (int i = tid * 16; i < 2 * HPATCH_BYTES; i += 256 * 16) *(u32x4*)(smem + i) = u32x4{0u, 0u, 0u, 0u};
    __syncthreads();
    if (t_begin >= t_end) return;

    // ---- the filter, all of it, as A fragments: lane (row = lane & 31, half = lane >> 5) holds 8 consecutive reduction
    //      channels of one output channel for every (tap, 16-deep K-step, output-channel half) ----
    bf16x8 wf[9][4][2];
    {
        const unsigned short* wg = (const unsigned short*)a.w;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int tt = MODE == 1 ? 8 - t : t;            // input gradient: mirrored taps of the swapped twin
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int chh = 0; chh < 2; ++chh) {
                    const int row = chh * 32 + (lane & 31);
                    wf[t][ks][chh] = *(const bf16x8*)(wg + ((row * 9 + tt) * 64 + ks * 16 + 8 * (lane >> 5)));
                }
        }
    }

    const unsigned xbytes = (unsigned)a.N * a.H * W * 64u * 2u;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, xbytes, 0x00020000);
    const u32x4 xrs4 = buffer_words(a.x, xbytes);
    constexpr unsigned OOB = 0x80000000u;
    const int tiles_per_img = a.H / HTR;
    const bool epi4 = MODE == 1 && a.epi == 4;
    const void* csrc_ = epi4 ? a.ep_src : a.x;               // BatchNorm input behind the input gradient (epi 4)
    const __amdgpu_buffer_rsrc_t crs = __builtin_amdgcn_make_buffer_rsrc((void*)csrc_, 0, xbytes, 0x00020000);
    const u32x4 crs4 = buffer_words(csrc_, xbytes);

    // ---- DMA lane roles: lane -> (pixel of the 8-pixel piece, 16-byte chunk position); the chunk it FETCHES is the one
    //      the slot's XOR key puts at its position (key = bits 1..3 of the absolute slot index) ----
    const int dpx = lane >> 3, dch = lane & 7;
    auto issue = [&](int tile, int buf) {
        const int n = tile / tiles_per_img, h0 = (tile - n * tiles_per_img) * HTR;
        unsigned char* base = smem + buf * HPATCH_BYTES;
#pragma unroll
        for (int i = 0; i < HLPT; ++i) {
            const int piece = wave + 4 * i;                              // wave-uniform
            if (piece < HPIECES) {
                const int j = piece / (W / 8), grp = piece - j * (W / 8);
                const int h = h0 - 1 + j;
                const int s0 = HLEAD + j * RS + 1 + grp * 8;             // first slot of the piece
                const int key = ((s0 + dpx) >> 1) & 7;
                unsigned off = ((unsigned)((n * a.H + h) * W + grp * 8 + dpx) * 64u + (unsigned)((dch ^ key) * 8)) * 2u;
                if ((unsigned)h >= (unsigned)a.H) off = OOB;             // rows above / below the image: zeros
                dma16<true>(xrs, xrs4, base + s0 * 128, off);
            } else {
                dma16<true>(xrs, xrs4, base, OOB);                       // keeps every wave's vmcnt arithmetic the same (zeros into lead slots)
            }
        }
    };

    // ---- B-fragment read addresses: lane (col = lane & 31, half = lane >> 5) reads 8 consecutive channels (K-step ks,
    //      half) of slot (wave + r) * 64 + 32 * pxh + col + s - 1.  Tap row r and pixel half pxh move the slot by multiples
    //      of 16, which the key ignores: 3 x 4 registers (tap column s, K-step), everything else an immediate ----
    unsigned baddr[3][4];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const int S = HLEAD + wave * RS + (lane & 31) + s - 1;
        const unsigned b0 = (unsigned)(S * 128 + ((((lane >> 5) ^ (S >> 1)) & 7) << 4));
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) baddr[s][ks] = b0 ^ (unsigned)(ks << 5);
    }

    // ---- write-out roles: lane -> chunk (lane & 7) of this wave's slots (lane >> 3) + 8 i = output row `wave`, column
    //      slot - 1: element offset relative to the tile's first pixel = eoff0 + 512 i; slot 0 (i = 0, lane < 8) and
    //      slots 57.. (i = 7, lane >= 8) are the junk columns ----
    const int och = lane & 7, osl = lane >> 3;
    const unsigned eoff0 = (unsigned)((wave * W + osl - 1) * 64 + och * 8);      // (wraps for osl = 0: never used there)
    const unsigned rd_base = (unsigned)(osl * 128 + ((och ^ osl) << 4));
    unsigned char* const my_stage = stage + wave * 8192;
    unsigned char* const my_cimg = cimg + wave * 7168;

    float s8[8], q8[8];                     // this lane's statistics over every tile of the workgroup
#pragma unroll
    for (int j = 0; j < 8; ++j) { s8[j] = 0.f; q8[j] = 0.f; }
    const bool want_stats = MODE == 0 ? a.stats != nullptr : a.epi == 4;
    unsigned short* const yg = (unsigned short*)a.y;

    issue(t_begin, 0);
    int bufoff = 0;                          // byte offset of the patch buffer holding tile t
    for (int t = t_begin; t < t_end; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's pieces of tile t have landed (and its stores left)
        __syncthreads();                                          // ... everyone's; and everyone has left the buffer tile t + 1 takes
        if (t + 1 < t_end) issue(t + 1, bufoff == 0 ? 1 : 0);
        const int n = t / tiles_per_img, h0 = (t - n * tiles_per_img) * HTR;
        const unsigned pix0 = (unsigned)((n * a.H + h0) * W) * 64u;      // element offset of the tile's first pixel
        unsigned kx[8];
        if (epi4) {
            // the BatchNorm input behind THIS tile's row (LDS-DMA, lands under the MFMAs) and the ReLU bits of this lane's chunks
            const unsigned cb = (pix0 + (unsigned)(wave * W) * 64u) * 2u + (unsigned)lane * 16u;
#pragma unroll
            for (int i = 0; i < 7; ++i) dma16<true>(crs, crs4, my_cimg + i * 1024, cb + (unsigned)i * 1024u);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const bool ok = !(i == 0 && osl == 0) && !(i == 7 && osl >= 1);
                kx[i] = ok ? (unsigned)a.ep_mask_in[(size_t)(pix0 + eoff0 + 512u * i) >> 3] : 0u;
            }
        }
        __builtin_amdgcn_sched_barrier(0);

        floatx16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        const unsigned char* pb = smem + bufoff;
        // 36 steps (tap, K-step) of 2 fragment reads + 4 MFMAs (128 MFMA cycles).  One wave per SIMD has no neighbour to hide
        // an LDS round trip behind, so the reads run TWO steps ahead of their MFMAs through three named register pairs
        // (hipcc left alone re-uses one register quad and waits lgkmcnt(0) in front of every MFMA pair); the scheduling
        // barriers keep each step's reads in front of the MFMAs they fly under.
        bf16x8 bq[3][2];
#define DPA_HALO_READ(STEP, SET)                                                                                  \
        {                                                                                                         \
            const unsigned char* p_ = pb + baddr[((STEP) / 4) % 3][(STEP) % 4] + ((STEP) / 12) * (RS * 128);      \
            bq[SET][0] = *(const bf16x8*)(p_);                                                                    \
            bq[SET][1] = *(const bf16x8*)(p_ + 32 * 128);                                                         \
        }
        DPA_HALO_READ(0, 0)
        DPA_HALO_READ(1, 1)
#pragma unroll
        for (int st_ = 0; st_ < 36; ++st_) {
            if (st_ + 2 < 36) DPA_HALO_READ(st_ + 2, (st_ + 2) % 3)
            __builtin_amdgcn_sched_barrier(0);
            const int tp = st_ / 4, ks = st_ % 4;
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[tp][ks][0], bq[st_ % 3][0], acc[0][0], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[tp][ks][1], bq[st_ % 3][0], acc[1][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[tp][ks][0], bq[st_ % 3][1], acc[0][1], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[tp][ks][1], bq[st_ % 3][1], acc[1][1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
#undef DPA_HALO_READ

        // ---- epilogue: accumulators -> bf16 -> this wave's LDS image ([slot][64 channels], 16-byte chunks XOR-keyed by
        //      slot & 7), then 16 bytes per lane over whole NHWC rows.  C layout of the 32 x 32 MFMA: column = lane & 31 (the
        //      slot), rows 8 (reg >> 2) + 4 (lane >> 5) + (reg & 3) (the channel): a register quad is 4 consecutive channels ----
        if (epi4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's own image of the BatchNorm input (wave-private: no barrier)
#pragma unroll
        for (int chh = 0; chh < 2; ++chh)
#pragma unroll
            for (int pxh = 0; pxh < 2; ++pxh) {
                const int sl = 32 * pxh + (lane & 31);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    uint2 pk;
                    pk.x = pack_bf16x2(acc[chh][pxh][4 * g + 0], acc[chh][pxh][4 * g + 1]);
                    pk.y = pack_bf16x2(acc[chh][pxh][4 * g + 2], acc[chh][pxh][4 * g + 3]);
                    *(uint2*)(my_stage + sl * 128 + ((((4 * chh + g) ^ sl) & 7) << 4) + 8 * (lane >> 5)) = pk;
                }
            }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint4 v = *(const uint4*)(my_stage + rd_base + i * 1024);
            const bool ok = !(i == 0 && osl == 0) && !(i == 7 && osl >= 1);
            if (ok) {
                const unsigned e = pix0 + eoff0 + 512u * i;
                *(uint4*)(yg + (size_t)e) = v;
                if (want_stats) {
                    float f[8];
                    unpack8(v, f);
                    if (MODE == 0) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) { s8[j] += f[j]; q8[j] += f[j] * f[j]; }
                    } else {
                        float c[8];
                        unpack8(*(const uint4*)(my_cimg + ((osl - 1 + 8 * i) * 128 + och * 16)), c);
                        const unsigned mk = kx[i];
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const float gm = ((mk >> j) & 1u) ? f[j] : 0.f;
                            s8[j] += gm; q8[j] += gm * c[j];
                        }
                    }
                }
            }
        }
        bufoff = bufoff == 0 ? HPATCH_BYTES : 0;
    }

    // ---- statistics: 32 partials (4 waves x 8 lane rows) per channel and statistic, folded through LDS; one atomic per
    //      channel and statistic per workgroup, into replica (workgroup mod rows) of the zero-initialised [rows][2][64] buffer ----
    if (want_stats) {
        __syncthreads();
        float* red = (float*)smem;                                   // [2][32][64]
        const int prow = wave * 8 + osl;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            red[(0 * 32 + prow) * 64 + och * 8 + j] = s8[j];
            red[(1 * 32 + prow) * 64 + och * 8 + j] = q8[j];
        }
        __syncthreads();
        if (tid < 128) {
            const int st_ = tid >> 6, ch = tid & 63;
            float tsum = 0.f;
#pragma unroll 8
            for (int r = 0; r < 32; ++r) tsum += red[(st_ * 32 + r) * 64 + ch];
            float* dst = MODE == 0 ? a.stats : a.ep_stats;
            const int rows = MODE == 0 ? a.stats_rows : a.ep_stats_rows;
            atomicAdd(dst + ((long)(lid % rows) * 2 + st_) * 64 + ch, tsum);
        }
    }
}

int g_halo_conv = -1;       // -1: not yet read from the environment (DPA_CONV_HALO, default 1)
int halo_conv_mode() {
    if (g_halo_conv < 0) { const char* e = getenv("DPA_CONV_HALO"); g_halo_conv = e ? (atoi(e) != 0) : 1; }
    return g_halo_conv;
}

template <int MODE>
void launch_c64(const ConvArgs& a, hipStream_t st) {
    const int tiles_total = a.N * (a.H / HTR);
    auto kernel = conv3x3_c64_kernel<MODE>;
    static bool attr_set = false;
    if (!attr_set) {
        const hipError_t err = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, HLDS_BYTES);
        if (err != hipSuccess) {
            fprintf(stderr, "conv3x3_halo: cannot allow %d bytes of LDS: %s\n", HLDS_BYTES, hipGetErrorString(err));
            abort();
        }
        attr_set = true;
    }
    const int grid = tiles_total < 256 ? tiles_total : 256;     // one 4-wave workgroup per CU: a wave owns a SIMD's register file
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), HLDS_BYTES, st, a, tiles_total);
}

}  // namespace

// 3 x 3, stride 1, pad 1, 64 -> 64 channels on 56-wide maps (H a multiple of 4), forward with optional BatchNorm statistics
// or input gradient with optional epi 4; everything else stays on conv_igemm.hip.  DPA_CONV_HALO=0 switches it off.
bool conv3x3_halo_ok(const ConvArgs& a) {
    if (halo_conv_mode() == 0 || deterministic()) return false;
    if (a.R != 3 || a.S != 3 || a.stride != 1 || a.pad != 1 || a.asym) return false;
    if (a.C != 64 || a.K != 64 || a.W != HW_ || a.Wo != HW_ || a.H != a.Ho || a.H % HTR != 0) return false;
    if (a.bias != nullptr || a.act != 0 || a.addend != nullptr || a.y_act != nullptr || a.y == nullptr) return false;
    if ((long)a.N * a.H * a.W * 64 >= (1L << 30)) return false;
    if (a.mode == 0) return a.epi == 0 && (a.stats == nullptr || a.stats_rows >= 1);
    return (a.epi == 0 && a.stats == nullptr) || (a.epi == 4 && a.ep_stats != nullptr && a.ep_stats_rows >= 1);
}

void set_conv3x3_halo(int on) { g_halo_conv = on ? 1 : 0; }
int conv3x3_halo_enabled() { return halo_conv_mode(); }

void conv3x3_halo(const ConvArgs& a, hipStream_t st) {
    if (a.mode == 0) launch_c64<0>(a, st);
    else             launch_c64<1>(a, st);
}

}  // namespace dpa
