// fp32 NCHW kernels for the reference's MNIST networks (Net / Net1 / Net2).
//
// The whole model is ~0.5 MMAC per sample, i.e. launch-latency bound on MI355X, so
// the design goal is FEW launches, not MFMA tiling.  A training step of Net is 12 launches: conv + bias + 2x2 max-pool
// + ReLU (+ channel dropout drawn in line from the counter hash) forward x2, Linear + bias + ReLU + element dropout,
// Linear + log-softmax, NLL; backward: NLL, (log-softmax + Linear) dx|dw in one role-split launch, Linear dx|dw,
// conv2 dx|dw, conv1 dw; step counter; fused SGD.  Dropout masks are never stored: forward draws them from
// (seed, step, element), backward only needs "output > 0" and the constant 1/(1-p).  Weight gradients are ADDED into
// whatever the caller passes (the flat gradient buffer when a sink is installed): no ATen add or fill kernels.
// (Reference ops: multinode-DP/multinodeDP.py:88-95, multinode-MP/multinodeMP.py:84-100.)
#include <stdlib.h>
#include "common.h"
#include "kernels.h"

namespace dpa {
namespace {

struct DropArg { float p; uint64_t seed; uint64_t offset; const int64_t* step; };   // p == 0: no dropout

constexpr int KS = 5;  // 5x5 filters, stride 1, no padding (DP.py:82-83)

// one thread per pooled output (n, co, ph, pw)
__global__ __launch_bounds__(256) void conv5_pool_relu_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                  const float* __restrict__ b, float* __restrict__ y,
                                                                  uint8_t* __restrict__ idx, int N, int Cin, int H, int W,
                                                                  int Cout, DropArg d) {
    const int PH = (H - KS + 1) / 2, PW = (W - KS + 1) / 2;
    const long total = (long)N * Cout * PH * PW;
    const uint64_t doff = d.offset + (d.step != nullptr ? (uint64_t)d.step[0] * 2000003ull : 0ull);
    const float keep_scale = d.p > 0.f ? 1.f / (1.f - d.p) : 1.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        int pw = (int)(i % PW); long t = i / PW;
        int ph = (int)(t % PH); t /= PH;
        int co = (int)(t % Cout); int n = (int)(t / Cout);
        float a00 = 0.f, a01 = 0.f, a10 = 0.f, a11 = 0.f;
        const float* wp = w + (long)co * Cin * KS * KS;
        for (int ci = 0; ci < Cin; ++ci) {
            const float* xp = x + (((long)n * Cin + ci) * H + ph * 2) * W + pw * 2;
            float patch[6][6];
#pragma unroll
            for (int r = 0; r < 6; ++r)
#pragma unroll
                for (int s = 0; s < 6; ++s) patch[r][s] = xp[r * W + s];
#pragma unroll
            for (int r = 0; r < KS; ++r)
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    const float wv = wp[(ci * KS + r) * KS + s];
                    a00 += wv * patch[r][s];     a01 += wv * patch[r][s + 1];
                    a10 += wv * patch[r + 1][s]; a11 += wv * patch[r + 1][s + 1];
                }
        }
        float best = a00; int bi = 0;
        if (a01 > best) { best = a01; bi = 1; }
        if (a10 > best) { best = a10; bi = 2; }
        if (a11 > best) { best = a11; bi = 3; }
        best = fmaxf(best + b[co], 0.f);
        if (d.p > 0.f) best *= dropout_keep(d.seed, doff, (uint64_t)(n * Cout + co), d.p, keep_scale);   // Dropout2d: per (sample, channel)
        y[i] = best;
        idx[i] = (uint8_t)bi;
    }
}

// gradient w.r.t. the conv output at the arg-max position of pooled cell i
// (a dropped channel has y == 0, a kept one was scaled by 1/(1-p) = gscale: no mask needed in backward)
__device__ __forceinline__ float pooled_grad(const float* dy, const float* y, float gscale, long i) {
    return y[i] > 0.f ? dy[i] * gscale : 0.f;
}

// dw: blockIdx.x = (co, ci), blockIdx.y = slice of the (n, ph, pw) cells; 25 accumulators per thread, block
// reduction, then one atomic per filter tap into the zero-initialised dw / db (a 10-filter layer would
// otherwise run on 10 workgroups of a 256-CU chip)
__device__ __forceinline__ void conv5_dw_body(int bx, int by, int ny, const float* __restrict__ x, const float* __restrict__ dy,
                                              const float* __restrict__ y, const uint8_t* __restrict__ idx,
                                              float* __restrict__ dw, float* __restrict__ db, int N, int Cin,
                                              int H, int W, int Cout, float gscale, float (*red)[KS * KS + 1]) {
    const int PH = (H - KS + 1) / 2, PW = (W - KS + 1) / 2;
    const int co = bx / Cin, ci = bx % Cin;
    float acc[KS * KS];
#pragma unroll
    for (int k = 0; k < KS * KS; ++k) acc[k] = 0.f;
    float bsum = 0.f;
    const int cells = N * PH * PW;
    const int per = (cells + ny - 1) / ny;
    const int c0 = by * per, c1 = min(cells, c0 + per);
    // Latency-bound: a cell is a three-deep dependent chain (dy, y, idx -> patch address -> 25 patch loads).  Four cells
    // go through it together - 12 independent loads, then 100 - and a cell without gradient (ReLU / dropout / past the
    // end) multiplies a valid patch by zero instead of branching around its loads.
    for (int cb = c0 + threadIdx.x; cb < c1; cb += 256 * 4) {
        float g[4]; const float* xp[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = cb + 256 * u;
            const bool ok = c < c1;
            const int cc = ok ? c : c0;
            const int pw = cc % PW; int t = cc / PW; const int ph = t % PH; const int n = t / PH;
            const long i = (((long)n * Cout + co) * PH + ph) * PW + pw;
            const float gv = pooled_grad(dy, y, gscale, i);
            const int bi = idx[i];
            g[u] = ok ? gv : 0.f;
            const int oh = ph * 2 + (bi >> 1), ow = pw * 2 + (bi & 1);
            xp[u] = x + (((long)n * Cin + ci) * H + oh) * W + ow;
        }
        float patch[4][KS * KS];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int r = 0; r < KS; ++r)
#pragma unroll
                for (int s = 0; s < KS; ++s) patch[u][r * KS + s] = xp[u][r * W + s];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            bsum += g[u];
#pragma unroll
            for (int k = 0; k < KS * KS; ++k) acc[k] += g[u] * patch[u][k];
        }
    }
#pragma unroll
    for (int k = 0; k < KS * KS; ++k) {
        const float v = wave_reduce_sum(acc[k]);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = v;
    }
    {
        const float v = wave_reduce_sum(bsum);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][KS * KS] = v;
    }
    __syncthreads();
    if (threadIdx.x < KS * KS)
        atomicAdd(dw + ((long)co * Cin + ci) * KS * KS + threadIdx.x,
                  red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
    if (threadIdx.x == 0 && ci == 0) atomicAdd(db + co, red[0][KS * KS] + red[1][KS * KS] + red[2][KS * KS] + red[3][KS * KS]);
}

__global__ __launch_bounds__(256) void conv5_dw_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                       const float* __restrict__ y, const uint8_t* __restrict__ idx,
                                                       float* __restrict__ dw, float* __restrict__ db, int N, int Cin,
                                                       int H, int W, int Cout, float gscale) {
    __shared__ float red[4][KS * KS + 1];
    conv5_dw_body(blockIdx.x, blockIdx.y, gridDim.y, x, dy, y, idx, dw, db, N, Cin, H, W, Cout, gscale, red);
}

// dx, LDS-tiled, shapes fixed at compile time (the reference's conv2: 10 -> 20 channels on 12x12): one
// workgroup per sample.  The sample's conv-output gradient is scattered into a ZERO-PADDED dense
// [Cout][OH+8][OW+8] LDS image (non-zero only at the pooling arg-max positions; the padding makes every
// filter tap in range, so the hot loop has no conditions), the layer's filters sit in LDS as well, and
// each thread produces one input ROW (ci, h, 0..W-1) in registers: per (co, r) it reads one padded
// gradient row (4 x ds_read_b128) and five taps for 5*W FMAs of straight-line code.
template <int CIN, int COUT, int HH, int WW>
struct Conv5DxTile {
    static constexpr int OH = HH - KS + 1, OW = WW - KS + 1, PH = OH / 2, PW = OW / 2;
    static constexpr int GH = OH + 2 * (KS - 1), GW = (OW + 2 * (KS - 1) + 3) / 4 * 4;    // padded image, rows 16-byte multiples
    static constexpr int G_FLOATS = COUT * GH * GW, W_FLOATS = COUT * CIN * KS * KS, ROWS = CIN * HH;
    static_assert(GW >= WW + KS - 1, "padded row must cover every tap of every output column");
    static_assert(2 * ROWS <= 256 && COUT % 2 == 0, "two thread groups, each half of the filters");
    static_assert(ROWS * WW <= G_FLOATS, "the partial sums of the second group reuse the gradient image");

    // Threads [0, ROWS) take filters [0, COUT/2), threads [ROWS, 2 ROWS) the other half of the same input row (the
    // layer has 120 rows per sample: one group would leave half the workgroup idle on a latency-bound loop); the
    // second group's partial row goes through LDS.
    static __device__ __forceinline__ void run(int n, const float* __restrict__ w, const float* __restrict__ dy,
                                               const float* __restrict__ y, const uint8_t* __restrict__ idx,
                                               float* __restrict__ dx, float gscale, float* G, float* Wl) {
        for (int i = threadIdx.x; i < G_FLOATS; i += 256) G[i] = 0.f;
        for (int i = threadIdx.x; i < W_FLOATS; i += 256) Wl[i] = w[i];
        __syncthreads();
        for (int i = threadIdx.x; i < COUT * PH * PW; i += 256) {
            const int pw = i % PW; int t = i / PW; const int ph = t % PH; const int co = t / PH;
            const long o = (long)n * COUT * PH * PW + i;
            const int bi = idx[o];
            G[(co * GH + ph * 2 + (bi >> 1) + KS - 1) * GW + pw * 2 + (bi & 1) + KS - 1] = pooled_grad(dy, y, gscale, o);
        }
        __syncthreads();
        const int half = threadIdx.x >= ROWS ? 1 : 0, row = threadIdx.x - half * ROWS;
        const bool active = threadIdx.x < 2 * ROWS;
        const int ci = row / HH, h = row - ci * HH;
        float acc[WW];
#pragma unroll
        for (int v = 0; v < WW; ++v) acc[v] = 0.f;
        if (active) {
            for (int co = half * (COUT / 2); co < (half + 1) * (COUT / 2); ++co) {
#pragma unroll
                for (int r = 0; r < KS; ++r) {
                    // dx[h][v] += sum_s G[h - r][v - s] * w[r][s];  padded coordinates: row h - r + 4, column v - s + 4
                    const float4* gp4 = (const float4*)(G + (co * GH + h - r + KS - 1) * GW);
                    float gr[GW];
#pragma unroll
                    for (int q = 0; q < GW / 4; ++q) {
                        const float4 t4 = gp4[q];
                        gr[4 * q] = t4.x; gr[4 * q + 1] = t4.y; gr[4 * q + 2] = t4.z; gr[4 * q + 3] = t4.w;
                    }
                    const float* wp = Wl + ((co * CIN + ci) * KS + r) * KS;
#pragma unroll
                    for (int sft = 0; sft < KS; ++sft) {
                        const float wv = wp[sft];
#pragma unroll
                        for (int v = 0; v < WW; ++v) acc[v] += gr[v - sft + KS - 1] * wv;
                    }
                }
            }
        }
        __syncthreads();                                   // everyone is done reading G: its front becomes the exchange buffer
        if (active && half == 1) {
#pragma unroll
            for (int v = 0; v < WW; ++v) G[row * WW + v] = acc[v];
        }
        __syncthreads();
        if (active && half == 0) {
            float* dst = dx + (((long)n * CIN + ci) * HH + h) * WW;
#pragma unroll
            for (int v = 0; v < WW; ++v) dst[v] = acc[v] + G[row * WW + v];
        }
    }
};

// conv backward of the reference's conv2 in ONE launch: workgroups [0, N) produce dx (one sample each), the rest dw / db
// (filter pair x slice of the cells).  The two halves are independent; one launch instead of two, running side by side.
template <int CIN, int COUT, int HH, int WW>
__global__ __launch_bounds__(256) void conv5_bwd_fused_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ dy, const float* __restrict__ y,
                                                              const uint8_t* __restrict__ idx, float* __restrict__ dx,
                                                              float* __restrict__ dw, float* __restrict__ db, int N, float gscale,
                                                              int split) {
    using T = Conv5DxTile<CIN, COUT, HH, WW>;
    __shared__ __attribute__((aligned(16))) float G[T::G_FLOATS];
    __shared__ __attribute__((aligned(16))) float Wl[T::W_FLOATS];
    if ((int)blockIdx.x < N) { T::run(blockIdx.x, w, dy, y, idx, dx, gscale, G, Wl); return; }
    const int b = blockIdx.x - N;
    conv5_dw_body(b / split, b % split, split, x, dy, y, idx, dw, db, N, CIN, HH, WW, COUT, gscale, (float (*)[KS * KS + 1])G);
}

// dx: one thread per input element; walks the (co, r, s) taps and keeps those that hit an arg-max
__global__ __launch_bounds__(256) void conv5_dx_kernel(const float* __restrict__ w, const float* __restrict__ dy,
                                                       const float* __restrict__ y, const uint8_t* __restrict__ idx,
                                                       float* __restrict__ dx, int N, int Cin, int H, int W, int Cout,
                                                       float gscale) {
    const int OH = H - KS + 1, OW = W - KS + 1, PH = OH / 2, PW = OW / 2;
    const long total = (long)N * Cin * H * W;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        int wv = (int)(i % W); long t = i / W;
        int h = (int)(t % H); t /= H;
        int ci = (int)(t % Cin); int n = (int)(t / Cin);
        float acc = 0.f;
        for (int co = 0; co < Cout; ++co) {
            const float* wp = w + ((long)co * Cin + ci) * KS * KS;
            const long obase = ((long)n * Cout + co) * PH * PW;
#pragma unroll
            for (int r = 0; r < KS; ++r) {
                const int oh = h - r;
                if (oh < 0 || oh >= PH * 2) continue;
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    const int ow = wv - s;
                    if (ow < 0 || ow >= PW * 2) continue;
                    const long o = obase + (oh >> 1) * PW + (ow >> 1);
                    if (idx[o] == ((oh & 1) * 2 + (ow & 1)) && y[o] > 0.f) acc += dy[o] * gscale * wp[r * KS + s];
                }
            }
        }
        dx[i] = acc;
    }
}

// y = dropout(act(x @ w^T + b)): 8 lanes per (row, out) split the K loop (a 320-long serial FMA chain per thread made the
// 128 x 50 layer run 12 us on 25 workgroups), folded with three lane shuffles
__global__ __launch_bounds__(256) void linear_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ b, float* __restrict__ y, int rows,
                                                         int in, int out, int relu, DropArg d) {
    const long total = (long)rows * out;
    const int sub = threadIdx.x & 7;
    const uint64_t doff = d.offset + (d.step != nullptr ? (uint64_t)d.step[0] * 2000003ull : 0ull);
    const float keep_scale = d.p > 0.f ? 1.f / (1.f - d.p) : 1.f;
    const long groups = ((long)gridDim.x * 256) >> 3;
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) >> 3; i < total; i += groups) {   // the 8 lanes of a group share i
        const int o = (int)(i % out); const long r = i / out;
        const float* xp = x + r * in; const float* wp = w + (long)o * in;
        float acc = 0.f;
        if ((in & 3) == 0) {
            for (int k = sub * 4; k < in; k += 32) {
                const float4 a = *(const float4*)(xp + k), c = *(const float4*)(wp + k);
                acc += a.x * c.x + a.y * c.y + a.z * c.z + a.w * c.w;
            }
        } else {
            for (int k = sub; k < in; k += 8) acc += xp[k] * wp[k];
        }
        acc += __shfl_xor(acc, 1, 64); acc += __shfl_xor(acc, 2, 64); acc += __shfl_xor(acc, 4, 64);
        if (sub == 0) {
            float v = acc + (b != nullptr ? b[o] : 0.f);
            if (relu) v = fmaxf(v, 0.f);
            if (d.p > 0.f) v *= dropout_keep(d.seed, doff, (uint64_t)i, d.p, keep_scale);
            y[i] = v;
        }
    }
}

// logp = log_softmax(x @ w^T + b) for a narrow output layer (out <= 64): one wave per row, lane o owns class o
__global__ __launch_bounds__(256) void linear_lsm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ b, float* __restrict__ logp, int rows,
                                                             int in, int out) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* xp = x + (long)row * in;
    float v = -INFINITY;
    if (lane < out) {
        const float* wp = w + (long)lane * in;
        float acc = b != nullptr ? b[lane] : 0.f;
        for (int k = 0; k < in; ++k) acc += xp[k] * wp[k];
        v = acc;
    }
    const float m = wave_reduce_max(v);
    const float s = wave_reduce_sum(lane < out ? __expf(v - m) : 0.f);
    if (lane < out) logp[(long)row * out + lane] = v - m - __logf(s);
}

// The classifier head and its loss in ONE single-workgroup launch (out <= 16):  logp = log_softmax(x @ w^T + b),
// loss = -mean_r logp[r][t_r]  (reference DP.py:94-95,152 - three framework calls, here one kernel; the loss needs no
// atomics and no zeroed accumulator because one workgroup sees every row).  16 lanes = one row, lane = class; x and w are
// staged through LDS with one batch of independent loads (x was written by the previous kernel, possibly on another XCD:
// ~2 us per DEPENDENT step), in row chunks of `chunk` rows.  `step` (optional): the device-resident dropout step counter
// of models/mnist.py, bumped here - this is the last forward kernel, every reader of the counter has run - so a replayed
// hipGraph needs no one-thread kernel for it.
__global__ __launch_bounds__(1024) void linear_lsm_nll_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                  const float* __restrict__ b, const int64_t* __restrict__ t,
                                                                  float* __restrict__ logp, float* __restrict__ loss,
                                                                  int64_t* step, int rows, int in, int out, int chunk) {
    extern __shared__ float lsm_smem[];
    float* ws = lsm_smem;                       // [16][in]
    float* xs = lsm_smem + 16 * in;             // [chunk][in]
    const int tid = threadIdx.x, cls = tid & 15, rsub = tid >> 4;
    for (int i = tid; i < out * in; i += 1024) ws[i] = w[i];
    float lacc = 0.f;
    for (int r0 = 0; r0 < rows; r0 += chunk) {
        const int nr = rows - r0 < chunk ? rows - r0 : chunk;
        __syncthreads();
        for (int i = tid; i < nr * in; i += 1024) xs[i] = x[(long)r0 * in + i];
        __syncthreads();
        for (int rl = rsub; rl < nr; rl += 64) {          // a 16-lane group stays together: the shuffles below are safe
            float v = -INFINITY;
            if (cls < out) {
                float acc = b != nullptr ? b[cls] : 0.f;
                const float* xr = xs + rl * in;
                const float* wr = ws + cls * in;
                for (int k = 0; k < in; ++k) acc += xr[k] * wr[k];
                v = acc;
            }
            float m = v;
            m = fmaxf(m, __shfl_xor(m, 1, 64)); m = fmaxf(m, __shfl_xor(m, 2, 64));
            m = fmaxf(m, __shfl_xor(m, 4, 64)); m = fmaxf(m, __shfl_xor(m, 8, 64));
            float e = cls < out ? __expf(v - m) : 0.f;
            e += __shfl_xor(e, 1, 64); e += __shfl_xor(e, 2, 64); e += __shfl_xor(e, 4, 64); e += __shfl_xor(e, 8, 64);
            const long r = r0 + rl;
            if (cls < out) {
                const float lp = v - m - __logf(e);
                logp[r * out + cls] = lp;
                if ((int64_t)cls == t[r]) lacc -= lp;
            }
        }
    }
    lacc = wave_reduce_sum(lacc);
    __shared__ float red[16];
    if ((tid & 63) == 0) red[tid >> 6] = lacc;
    __syncthreads();
    if (tid == 0) {
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) sum += red[i];
        loss[0] = sum / (float)rows;
        if (step != nullptr) step[0] += 1;
    }
}

// Upstream gradient of a Linear's pre-activation, element (r, o):
//   plain:  g = dy * (relu ? (y > 0) * gscale : 1)          (y = the layer's stored output; gscale = 1/(1-p) of a dropout behind it)
//   lsm 1:  g = dy - exp(y) * sum_c dy[r][c]                 (y = log-probabilities: log-softmax backward in line)
//   lsm 2:  g = (exp(y) - [o == t_r]) * g0 / rows            (log-softmax AND mean-NLL backward in line: dy = the scalar
//                                                             loss gradient g0, t = the targets; no row sum to fetch)
// These loops are LATENCY-bound (a few hundred loads per thread, one wave per SIMD): every helper below issues its loads
// in independent batches of 8 before anything consumes them - a runtime-trip-count loop with one load per iteration
// costs one L2 round trip per iteration (measured: 40 us for the 128 x 10 head, 18 us for the 128 x 50 layer).
// NARROW (out <= 16, the classifier head): one predicated batch of 16, no loop - every load of a thread's work item is
// then independent of every other and the whole item costs ONE memory round trip (data written by the previous kernel
// comes from another XCD's L2 through the fabric: ~2 us per dependent step, which is what these kernels are made of)
template <bool NARROW>
__device__ __forceinline__ float lin_rowsum(const float* __restrict__ dy, long r, int out) {
    float s = 0.f;
    if (NARROW) {
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = u < out ? dy[r * out + u] : 0.f;
#pragma unroll
        for (int u = 0; u < 16; ++u) s += v[u];
        return s;
    }
    for (int c0 = 0; c0 < out; c0 += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = c0 + u < out ? dy[r * out + c0 + u] : 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    return s;
}
__device__ __forceinline__ float lin_g1(float dyv, float yv, int relu, float gscale, int lsm, float rowsum, bool hit = false) {
    if (lsm == 2) return (__expf(yv) - (hit ? 1.f : 0.f)) * rowsum;      // rowsum carries g0 / rows
    if (lsm) return dyv - __expf(yv) * rowsum;
    if (relu) return yv > 0.f ? dyv * gscale : 0.f;
    return dyv;
}

// Linear backward in ONE launch.  Workgroups [0, dx_blocks): dx[r][k] = sum_o g[r][o] * w[o][k], one thread per element.
// The rest: dw[o][k] (+)= sum_r g[r][o] * x[r][k], db[o] (+)= sum_r g[r][o]; block = (o, 64-wide k tile); the 4 waves take
// interleaved rows, folded through LDS.
template <bool NARROW>
__global__ __launch_bounds__(256) void linear_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ y, const float* __restrict__ dy,
                                                         float* __restrict__ dx, float* __restrict__ dw, float* __restrict__ db,
                                                         int rows, int in, int out, int relu, float gscale, int lsm,
                                                         int accumulate, int dx_blocks, const int64_t* __restrict__ tgt) {
    const float gs = lsm == 2 ? dy[0] / (float)rows : 0.f;
    if ((int)blockIdx.x < dx_blocks) {
        const long total = (long)rows * in;
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)dx_blocks * 256) {
            const int k = (int)(i % in); const long r = i / in;
            const float rs = lsm == 1 ? lin_rowsum<NARROW>(dy, r, out) : gs;
            float acc = 0.f;
            if (NARROW) {
                float dv[16], yv[16], wv[16];
                const int tr = lsm == 2 ? (int)tgt[r] : -1;
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const bool ok = u < out;
                    const long e = r * out + (ok ? u : 0);
                    dv[u] = (ok && lsm != 2) ? dy[e] : 0.f;
                    yv[u] = (relu || lsm) ? y[e] : 1.f;
                    wv[u] = ok ? w[(long)u * in + k] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 16; ++u) acc += (u < out ? lin_g1(dv[u], yv[u], relu, gscale, lsm, rs, u == tr) : 0.f) * wv[u];
                dx[i] = acc;
                continue;
            }
            for (int o0 = 0; o0 < out; o0 += 8) {
                float dv[8], yv[8], wv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const bool ok = o0 + u < out;
                    const long e = r * out + (ok ? o0 + u : 0);
                    dv[u] = ok ? dy[e] : 0.f;
                    yv[u] = (relu || lsm) ? y[e] : 1.f;
                    wv[u] = ok ? w[(long)(o0 + u) * in + k] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) acc += (o0 + u < out ? lin_g1(dv[u], yv[u], relu, gscale, lsm, rs) : 0.f) * wv[u];
            }
            dx[i] = acc;
        }
        return;
    }
    const int bid = blockIdx.x - dx_blocks;
    const int ktiles = (in + 63) / 64;
    const int o = bid / ktiles, k = (bid % ktiles) * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6;
    const bool kok = k < in;
    float acc = 0.f, bs = 0.f;
    for (int r = rg; r < rows; r += 32) {
        float dv[8], yv[8], xv[8], rs[8];
        int tr[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const long rr = r + 4 * u;
            const bool ok = rr < rows;
            const long e = (ok ? rr : 0) * out + o;
            dv[u] = (ok && lsm != 2) ? dy[e] : 0.f;
            yv[u] = (relu || lsm) ? y[e] : 1.f;
            xv[u] = (ok && kok) ? x[rr * in + k] : 0.f;
            rs[u] = (lsm == 1 && ok) ? lin_rowsum<NARROW>(dy, rr, out) : gs;
            tr[u] = (lsm == 2 && ok) ? (int)tgt[rr] : -1;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float g = r + 4 * u < rows ? lin_g1(dv[u], yv[u], relu, gscale, lsm, rs[u], tr[u] == o) : 0.f;
            acc += g * xv[u]; bs += g;
        }
    }
    __shared__ float red[2][4][64];
    red[0][rg][threadIdx.x & 63] = acc;
    red[1][rg][threadIdx.x & 63] = bs;
    __syncthreads();
    if (rg == 0) {
        const int l = threadIdx.x;
        if (kok) {
            const float v = red[0][0][l] + red[0][1][l] + red[0][2][l] + red[0][3][l];
            float* dst = dw + (long)o * in + k;
            *dst = accumulate ? *dst + v : v;              // one writer per element
        }
        if (l == 0 && bid % ktiles == 0 && db != nullptr) {
            const float v = red[1][0][0] + red[1][1][0] + red[1][2][0] + red[1][3][0];
            db[o] = accumulate ? db[o] + v : v;
        }
    }
}

// log_softmax over rows of a small [rows, V] fp32 matrix: one wave per row
__global__ __launch_bounds__(256) void log_softmax_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int rows, int V) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    float m = -INFINITY;
    for (int i = lane; i < V; i += 64) m = fmaxf(m, x[(long)row * V + i]);
    m = wave_reduce_max(m);
    float s = 0.f;
    for (int i = lane; i < V; i += 64) s += __expf(x[(long)row * V + i] - m);
    s = wave_reduce_sum(s);
    const float lse = m + __logf(s);
    for (int i = lane; i < V; i += 64) y[(long)row * V + i] = x[(long)row * V + i] - lse;
}

// dx = dy - exp(y) * sum(dy)
__global__ __launch_bounds__(256) void log_softmax_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                              float* __restrict__ dx, int rows, int V) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    float s = 0.f;
    for (int i = lane; i < V; i += 64) s += dy[(long)row * V + i];
    s = wave_reduce_sum(s);
    for (int i = lane; i < V; i += 64) dx[(long)row * V + i] = dy[(long)row * V + i] - __expf(y[(long)row * V + i]) * s;
}

// loss = -mean_r logp[r][t_r]  (mean over all rows, like F.nll_loss default)
__global__ __launch_bounds__(256) void nll_fwd_kernel(const float* __restrict__ logp, const int64_t* __restrict__ t,
                                                      float* __restrict__ loss, int rows, int V) {
    float s = 0.f;
    for (int r = threadIdx.x; r < rows; r += 256) s -= logp[(long)r * V + t[r]];
    s = wave_reduce_sum(s);
    __shared__ float red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) loss[0] = (red[0] + red[1] + red[2] + red[3]) / (float)rows;
}

__global__ __launch_bounds__(256) void nll_bwd_kernel(const float* __restrict__ g, const int64_t* __restrict__ t,
                                                      float* __restrict__ dlogp, int rows, int V) {
    const long total = (long)rows * V;
    const float v = -g[0] / (float)rows;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % V); const long r = i / V;
        dlogp[i] = (t[r] == c) ? v : 0.f;
    }
}

inline int grid_for(long n) { long b = (n + 255) / 256; if (b > 2048) b = 2048; if (b < 1) b = 1; return (int)b; }

}  // namespace

void conv5_pool_relu_fwd(const float* x, const float* w, const float* b, float* y, uint8_t* idx,
                         int N, int Cin, int H, int W, int Cout, float drop_p, uint64_t seed, uint64_t offset,
                         const int64_t* step, hipStream_t st) {
    const int PH = (H - KS + 1) / 2, PW = (W - KS + 1) / 2;
    hipLaunchKernelGGL(conv5_pool_relu_fwd_kernel, dim3(grid_for((long)N * Cout * PH * PW)), dim3(256), 0, st, x, w, b, y, idx,
                       N, Cin, H, W, Cout, DropArg{drop_p, seed, offset, step});
}

void conv5_pool_relu_bwd(const float* x, const float* w, const float* dy, const float* y, const uint8_t* idx,
                         float* dx, float* dw, float* db, int N, int Cin, int H, int W, int Cout,
                         float gscale, hipStream_t st) {
    // dw / db are ADDED to with atomics (pixel split): zero-initialised by the caller, or the flat gradient buffer
    const int PH = (H - KS + 1) / 2, PW = (W - KS + 1) / 2;
    const int cells = N * PH * PW;
    // every split adds one more same-address float atomic per filter tap (they serialise in L2: conv1's 10 filters
    // at 51 splits spent 10 of its 16 us there), so the cells are cut only until each thread has ~4 of them (one batch of the loop below)
    static const int cells_per_thread = [] { const char* e = getenv("DPA_CONV5_CPT"); return e ? atoi(e) : 4; }();
    int split = 512 / (Cout * Cin);
    if (split > (cells + 256 * cells_per_thread - 1) / (256 * cells_per_thread)) split = (cells + 256 * cells_per_thread - 1) / (256 * cells_per_thread);
    if (split < 1 || deterministic()) split = 1;            // one writer per filter tap (runtime.hip)
    if (dx != nullptr && Cin == 10 && Cout == 20 && H == 12 && W == 12) {   // the reference's conv2 (DP.py:83): dx | dw in one launch
        hipLaunchKernelGGL((conv5_bwd_fused_kernel<10, 20, 12, 12>), dim3(N + Cout * Cin * split), dim3(256), 0, st, x, w, dy, y,
                           idx, dx, dw, db, N, gscale, split);
        return;
    }
    hipLaunchKernelGGL(conv5_dw_kernel, dim3(Cout * Cin, split), dim3(256), 0, st, x, dy, y, idx, dw, db, N, Cin, H, W, Cout, gscale);
    if (dx != nullptr)
        hipLaunchKernelGGL(conv5_dx_kernel, dim3(grid_for((long)N * Cin * H * W)), dim3(256), 0, st, w, dy, y, idx, dx, N, Cin,
                           H, W, Cout, gscale);
}

void linear_f32_fwd(const float* x, const float* w, const float* b, float* y, int rows, int in, int out, int relu,
                    float drop_p, uint64_t seed, uint64_t offset, const int64_t* step, hipStream_t st) {
    hipLaunchKernelGGL(linear_fwd_kernel, dim3(grid_for((long)rows * out * 8)), dim3(256), 0, st, x, w, b, y, rows, in, out, relu,
                       DropArg{drop_p, seed, offset, step});
}

void linear_lsm_fwd(const float* x, const float* w, const float* b, float* logp, int rows, int in, int out, hipStream_t st) {
    hipLaunchKernelGGL(linear_lsm_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, x, w, b, logp, rows, in, out);
}

int linear_lsm_nll_chunk(int rows, int in) {               // rows of x staged per pass: 32 KiB of LDS
    int c = 8192 / (in > 0 ? in : 1);
    if (c > rows) c = rows;
    return c < 1 ? 1 : c;
}

void linear_lsm_nll_fwd(const float* x, const float* w, const float* b, const int64_t* t, float* logp, float* loss,
                        int64_t* step, int rows, int in, int out, hipStream_t st) {
    const int chunk = linear_lsm_nll_chunk(rows, in);
    const size_t lds = (size_t)(16 + chunk) * in * sizeof(float);
    hipLaunchKernelGGL(linear_lsm_nll_fwd_kernel, dim3(1), dim3(1024), lds, st, x, w, b, t, logp, loss, step, rows, in, out, chunk);
}

void linear_f32_bwd(const float* x, const float* w, const float* y, const float* dy, float* dx, float* dw, float* db,
                    int rows, int in, int out, int relu, float gscale, int lsm, int accumulate, const int64_t* tgt, hipStream_t st) {
    const int dxb = dx != nullptr ? grid_for((long)rows * in) : 0;
    const int dwb = out * ((in + 63) / 64);
    if (out <= 16)
        hipLaunchKernelGGL(linear_bwd_kernel<true>, dim3((unsigned)(dxb + dwb)), dim3(256), 0, st, x, w, y, dy, dx, dw, db, rows, in,
                           out, relu, gscale, lsm, accumulate, dxb, tgt);
    else
        hipLaunchKernelGGL(linear_bwd_kernel<false>, dim3((unsigned)(dxb + dwb)), dim3(256), 0, st, x, w, y, dy, dx, dw, db, rows, in,
                           out, relu, gscale, lsm, accumulate, dxb, tgt);
}

void log_softmax_fwd(const float* x, float* y, int rows, int V, hipStream_t st) {
    hipLaunchKernelGGL(log_softmax_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, x, y, rows, V);
}
void log_softmax_bwd(const float* dy, const float* y, float* dx, int rows, int V, hipStream_t st) {
    hipLaunchKernelGGL(log_softmax_bwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, dy, y, dx, rows, V);
}
void nll_fwd(const float* logp, const int64_t* t, float* loss, int rows, int V, hipStream_t st) {
    hipLaunchKernelGGL(nll_fwd_kernel, dim3(1), dim3(256), 0, st, logp, t, loss, rows, V);
}
void nll_bwd(const float* g, const int64_t* t, float* dlogp, int rows, int V, hipStream_t st) {
    hipLaunchKernelGGL(nll_bwd_kernel, dim3(grid_for((long)rows * V)), dim3(256), 0, st, g, t, dlogp, rows, V);
}

}  // namespace dpa
