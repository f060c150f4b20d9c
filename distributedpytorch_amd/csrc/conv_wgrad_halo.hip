// Halo-tiled weight gradient of the 3 x 3 / stride 1 / pad 1 convolutions (gfx950).
//
//   dw[k][r][s][c] += sum_{n,h,w} dy[n][h][w][k] * x[n][h+r-1][w+s-1][c]
//
// conv_wgrad.hip treats this as a GEMM with 9 C filter columns and gathers every x pixel nine times (once per tap) and
// every dy pixel once per 128 / 256 columns: 21-32 MACs per staged byte, and per 8 MFMAs 16 transposing LDS reads and
// ~63 VALU instructions of gather arithmetic - three pipes as busy as the matrix core (~300 TFLOP/s).
// Here a workgroup owns one 64-channel slab of dy and one 64-channel slab of x and walks image-row groups: TR rows of
// dy and the TR + 2 rows of x around them are staged ONCE (LDS-DMA) and all nine taps are formed from them, a tap being
// nothing but an LDS address offset: 144 MACs per staged byte.  The rows are stored RS = W + 2 slots wide with zero
// slots on both sides (written once, never touched again), so the reduction runs over the flat slot index, image
// borders need no bounds arithmetic, and x rows outside the image are zero-filled by the buffer descriptor.
//
//   block = 8 waves: (k-step parity) x (32 of the 64 out channels) x (32 of the 64 in channels); a wave keeps nine
//   32 x 32 accumulators (one per tap, 144 AGPRs) and per 16-slot k-step issues 2 + 18 transposing reads for 9 MFMAs;
//   every read address is a per-lane constant (the XOR key depends on bit 1 of the slot, which a k-step of 16 and a
//   second read 4 slots on leave alone) plus an immediate.
//   LDS: three stage buffers, the DMAs of tiles t + 1 and t + 2 in flight under the MFMAs of tile t, one barrier per tile.
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

template <int W, int TR>
struct HaloCfg {
    static constexpr int RS = W + 2;                             // slots per stored row (even for every W here)
    static constexpr int PPR = (W + 7) / 8;                      // 8-pixel DMA pieces per row
    static constexpr int KSTEPS = ((TR * RS + 15) / 16 + 1) / 2 * 2;   // 16-slot k-steps per tile, even (two wave groups)
    static constexpr int DY_SLOTS = KSTEPS * 16 + 8;             // tail slots stay zero; + 8: slot "-1" of the x buffer behind it
    static constexpr int X_SLOTS = (KSTEPS * 16 + 2 * RS + 2 + 7) / 8 * 8;   // furthest slot a tap read can touch
    static constexpr int DY_BYTES = DY_SLOTS * 128, X_BYTES = X_SLOTS * 128;
    static constexpr int DUMMY = DY_BYTES + X_BYTES;             // 1 KiB landing zone of the padding DMAs
    static constexpr int STAGE = DUMMY + 1024;
    static constexpr int PIECES = (2 * TR + 2) * PPR;
    static constexpr int LPT = (PIECES + 7) / 8;                 // DMA instructions per wave per tile
    static constexpr int FOLD_BYTES = 4 * 144 * 64 * 4;          // epilogue: four waves' accumulators
    static constexpr int LDS_BYTES = 3 * STAGE > FOLD_BYTES ? 3 * STAGE : FOLD_BYTES;
};

template <int W, int TR>
__global__ __launch_bounds__(512) void wgrad3x3_halo_kernel(WgradArgs a, int slabs_c, int wg_per_slab, int tiles_total) {
    using Cfg = HaloCfg<W, TR>;
    constexpr int RS = Cfg::RS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kpar = wave >> 2, kb = (wave >> 1) & 1, cb = wave & 1;

    const int slab = blockIdx.x / wg_per_slab, part = blockIdx.x - slab * wg_per_slab;
    const int k0 = (slab / slabs_c) * 64, c0 = (slab % slabs_c) * 64;
    // this workgroup's run of tiles (tile = TR rows of one image)
    const int per = (tiles_total + wg_per_slab - 1) / wg_per_slab;
    const int t_begin = part * per;
    int t_end = t_begin + per;
    if (t_end > tiles_total) t_end = tiles_total;

    // zero the stage buffers once: pad slots, slack slots and the landing zone are never written with data
    for (int i = tid * 16; i < 3 * Cfg::STAGE; i += 512 * 16) *(u32x4*)(smem + i) = u32x4{0u, 0u, 0u, 0u};
    __syncthreads();
    if (t_begin >= t_end) return;

    const unsigned dybytes = (unsigned)a.N * a.H * W * a.K * 2u, xbytes = (unsigned)a.N * a.H * W * a.C * 2u;
    const __amdgpu_buffer_rsrc_t dyrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, dybytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, xbytes, 0x00020000);
    const u32x4 dyrs4 = buffer_words(a.dy, dybytes), xrs4 = buffer_words(a.x, xbytes);
    constexpr unsigned OOB = 0x80000000u;

    // ---- DMA lane roles: lane -> (pixel of the 8-pixel piece, 16-byte chunk); the chunk it FETCHES is the one the
    //      swizzle puts at its LDS position.  key = bit 1 of the slot; slot = row * RS + 1 + 8 * piece + pixel, so it is
    //      bit 1 of (1 + pixel) flipped by the row's parity when RS / 2 is odd (RS = 58, 30): two lane constants ----
    const int dpx = lane >> 3, dch = lane & 7;
    const int dkey = ((1 + dpx) >> 1) & 1;
    // (the flipped key moves the chunk by 4 = 64 bytes; every other term of the offset is a multiple of 128: XOR 64)
    const int dsrc0 = (((dch >> 2) ^ dkey) << 2) | (dch & 3);
    const unsigned lane_dy0 = ((unsigned)dpx * (unsigned)a.K + (unsigned)(k0 + dsrc0 * 8)) * 2u;
    const unsigned lane_x0 = ((unsigned)dpx * (unsigned)a.C + (unsigned)(c0 + dsrc0 * 8)) * 2u;
    const int tiles_per_img = a.H / TR;

    // pieces of a tile: rows 0 .. TR-1 = dy rows h0 .., rows TR .. 2 TR + 1 = x rows h0 - 1 ..; PPR pieces per row
    auto issue = [&](int tile, int buf) {
        const int n = tile / tiles_per_img, h0 = (tile - n * tiles_per_img) * TR;
        unsigned char* base = smem + buf * Cfg::STAGE;
#pragma unroll
        for (int i = 0; i < Cfg::LPT; ++i) {
            const int piece = wave + 8 * i;                       // wave-uniform
            const int j = piece / Cfg::PPR, grp = piece - j * Cfg::PPR;
            if (piece < Cfg::PIECES) {
                const bool is_dy = j < TR;
                const int h = is_dy ? h0 + j : h0 - 1 + (j - TR);
                const bool row_ok = is_dy || ((unsigned)h < (unsigned)a.H);
                const unsigned pix = (unsigned)((n * a.H + h) * W + grp * 8);
                const int lrow = is_dy ? j : j - TR;
                unsigned char* dst = base + (is_dy ? 0 : Cfg::DY_BYTES) + (lrow * RS + 1 + grp * 8) * 128;
                const unsigned fl = (unsigned)((lrow * (RS / 2)) & 1) << 6;    // wave-uniform
                unsigned off = is_dy ? pix * (unsigned)a.K * 2u + (lane_dy0 ^ fl) : pix * (unsigned)a.C * 2u + (lane_x0 ^ fl);
                if (!row_ok) off = OOB;
                if (W % 8 == 0 || grp * 8 + dpx < W) {            // the last piece of a 28-wide row: 4 pixels only
                    if (is_dy) dma16<true>(dyrs, dyrs4, dst, off);
                    else       dma16<true>(xrs, xrs4, dst, off);
                }
            } else {
                dma16<true>(xrs, xrs4, base + Cfg::DUMMY, OOB);   // keeps every wave's vmcnt arithmetic the same
            }
        }
    };

    // ---- transposing-read lane roles (as in conv_wgrad.hip) ----
    const int gi = lane >> 4, li = lane & 15, tq = li >> 2, tp = li & 3;
    const int kofs = 8 * (gi >> 1) + tq;          // slot inside a 16-slot k-step (first read; the second is 4 on)
    const int cofs = (gi & 1) * 16 + 4 * tp;      // channel inside the wave's 32
    // per-lane byte offsets inside a stage buffer, k-step 0
    const int a_base = kofs * 128 + ((kb ^ ((kofs >> 1) & 1)) << 6) + cofs * 2;
    int b_base[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int toff = (t / 3) * RS + (t % 3) - 1;                  // x slot = dy slot + toff
        const int s0 = kofs + toff;                                   // >= -1
        b_base[t] = Cfg::DY_BYTES + s0 * 128 + ((cb ^ ((s0 >> 1) & 1)) << 6) + cofs * 2;
    }

    floatx16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // three stage buffers, two tiles in flight: tile t + 2 is issued right after the barrier of iteration t, when every
    // wave has left the MFMAs of tile t - 1, whose buffer it takes - one barrier per tile
    issue(t_begin, 0);
    if (t_begin + 1 < t_end) issue(t_begin + 1, 1);
    int buf = 0, nbuf = 2;
    for (int t = t_begin; t < t_end; ++t) {
        if (t + 1 < t_end) __builtin_amdgcn_s_waitcnt(waitcnt_vm_only(Cfg::LPT));   // tile t has landed (this wave's pieces)
        else               asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                              // ... and every other wave's
        if (t + 2 < t_end) issue(t + 2, nbuf);
        const unsigned char* st = smem + buf * Cfg::STAGE;
#pragma unroll
        for (int ks = 0; ks < Cfg::KSTEPS / 2; ++ks) {
            const int kk = 2 * ks + kpar;                             // wave-uniform
            const unsigned char* sk = st + kk * 2048;
            short4v alo = tr_read(sk + a_base), ahi = tr_read(sk + a_base + 512);
            const bf16x8 af = __builtin_bit_cast(bf16x8, __builtin_shufflevector(alo, ahi, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
            for (int tt = 0; tt < 9; ++tt) {
                short4v blo = tr_read(sk + b_base[tt]), bhi = tr_read(sk + b_base[tt] + 512);
                const bf16x8 bf = __builtin_bit_cast(bf16x8, __builtin_shufflevector(blo, bhi, 0, 1, 2, 3, 4, 5, 6, 7));
                acc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc[tt], 0, 0, 0);
            }
        }
        buf = buf == 2 ? 0 : buf + 1;
        nbuf = nbuf == 2 ? 0 : nbuf + 1;
    }

    // ---- the two k-step parities of a (kb, cb) pair are folded through LDS (the stage buffers are free now), then ONE
    //      wave of the pair adds into fp32 dw [K][3][3][C]: every workgroup costs one fp32 copy of its 64 x 9 x 64 slab
    //      in float-atomic traffic (chip-wide ~1.3 TB/s of added bytes), not two ----
    __syncthreads();
    float* fold = (float*)smem + ((wave & 3) * 144) * 64 + lane;
    if (kpar == 1) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) fold[(t * 16 + r) * 64] = acc[t][r];
    }
    __syncthreads();
    if (kpar == 1) return;
    const int ccol = lane & 31, rhalf = (lane >> 5) * 4;
    const long Kg = 9L * a.C;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int cg = t * a.C + c0 + cb * 32 + ccol;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = k0 + kb * 32 + (r & 3) + 8 * (r >> 2) + rhalf;
            atomicAdd(a.dw + (long)row * Kg + cg, acc[t][r] + fold[(t * 16 + r) * 64]);
        }
    }
}

int halo_mode() {
    static const int v = [] { const char* e = getenv("DPA_WGRAD_HALO"); return e ? atoi(e) : 1; }();
    return v;
}

template <int W, int TR>
void launch_halo(const WgradArgs& a, hipStream_t st) {
    using Cfg = HaloCfg<W, TR>;
    const int slabs_k = a.K / 64, slabs_c = a.C / 64, slabs = slabs_k * slabs_c;
    const int tiles_total = a.N * (a.H / TR);
    int wg_per_slab = 256 / slabs;                      // one 8-wave workgroup per CU (three stage buffers fill its LDS)
    if (wg_per_slab < 1) wg_per_slab = 1;
    if (wg_per_slab > tiles_total) wg_per_slab = tiles_total;
    auto kernel = wgrad3x3_halo_kernel<W, TR>;
    static bool attr_set = false;
    if (!attr_set) {
        const hipError_t err = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
        if (err != hipSuccess) {      // a launch without the attribute would fail silently and leave dw untouched
            fprintf(stderr, "conv_wgrad_halo: cannot allow %d bytes of LDS: %s\n", Cfg::LDS_BYTES, hipGetErrorString(err));
            abort();
        }
        attr_set = true;
    }
    hipLaunchKernelGGL(kernel, dim3(slabs * wg_per_slab), dim3(512), Cfg::LDS_BYTES, st, a, slabs_c, wg_per_slab, tiles_total);
}

}  // namespace

// 3 x 3, stride 1, pad 1, 64-channel slabs, image widths with a tuned row grouping; the statistics-free,
// non-deterministic (atomic) path only
bool conv_wgrad_halo_ok(const WgradArgs& a) {
    if (halo_mode() == 0 || deterministic()) return false;
    if (a.R != 3 || a.S != 3 || a.stride != 1 || a.pad != 1 || a.asym || a.Ho != a.H || a.Wo != a.W) return false;
    if (a.C % 64 != 0 || a.K % 64 != 0 || a.dbias != nullptr) return false;
    if ((long)a.N * a.H * a.W * (a.C > a.K ? a.C : a.K) >= (1L << 30)) return false;
    // Measured alone at batch 512 (scripts/bench_wgrad_halo.py): 56 x 56 x 64 -> 64: 295 -> 181 us; 28 x 28 x 128 -> 128:
    // 153 -> 164 us (there the implicit GEMM has 128 x 128 tiles' worth of operand reuse and no row padding of 30 / 28),
    // so 28-wide maps stay on conv_wgrad.hip unless DPA_WGRAD_HALO=2.
    if (a.W == 56) return a.H % 2 == 0;
    if (a.W == 28) return halo_mode() >= 2 && a.H % 4 == 0;
    return false;
}

void conv_wgrad_halo(const WgradArgs& a, hipStream_t st) {
    if (a.W == 56) launch_halo<56, 2>(a, st);
    else           launch_halo<28, 4>(a, st);
}

}  // namespace dpa
