// fp32 NCHW kernels for the reference's MNIST networks (Net / Net1 / Net2).
//
// The whole model is ~0.5 MMAC per sample, i.e. launch-latency bound on MI355X, so
// the design goal is FEW launches, not MFMA tiling: conv + bias + 2x2 max-pool +
// ReLU (+ channel dropout) is one kernel forward and three backward, a Linear with
// its bias and ReLU is one kernel, log-softmax + NLL is one kernel.
// (Reference ops: multinode-DP/multinodeDP.py:88-95, multinode-MP/multinodeMP.py:84-100.)
#include "common.h"
#include "kernels.h"

namespace dpa {
namespace {

constexpr int KS = 5;  // 5x5 filters, stride 1, no padding (DP.py:82-83)

// one thread per pooled output (n, co, ph, pw)
__global__ __launch_bounds__(256) void conv5_pool_relu_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                  const float* __restrict__ b, float* __restrict__ y,
                                                                  uint8_t* __restrict__ idx, int N, int Cin, int H, int W,
                                                                  int Cout, const float* __restrict__ drop) {
    const int PH = (H - KS + 1) / 2, PW = (W - KS + 1) / 2;
    const long total = (long)N * Cout * PH * PW;
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
        if (drop != nullptr) best *= drop[n * Cout + co];
        y[i] = best;
        idx[i] = (uint8_t)bi;
    }
}

// gradient w.r.t. the conv output at the arg-max position of pooled cell i
__device__ __forceinline__ float pooled_grad(const float* dy, const float* y, const float* drop, long i, int nc) {
    float g = y[i] > 0.f ? dy[i] : 0.f;
    if (drop != nullptr) g *= drop[nc];
    return g;
}

// dw: blockIdx.x = (co, ci), blockIdx.y = slice of the (n, ph, pw) cells; 25 accumulators per thread, block
// reduction, then one atomic per filter tap into the zero-initialised dw / db (a 10-filter layer would
// otherwise run on 10 workgroups of a 256-CU chip)
__global__ __launch_bounds__(256) void conv5_dw_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                       const float* __restrict__ y, const uint8_t* __restrict__ idx,
                                                       float* __restrict__ dw, float* __restrict__ db, int N, int Cin,
                                                       int H, int W, int Cout, const float* __restrict__ drop) {
    const int PH = (H - KS + 1) / 2, PW = (W - KS + 1) / 2;
    const int co = blockIdx.x / Cin, ci = blockIdx.x % Cin;
    float acc[KS * KS];
#pragma unroll
    for (int k = 0; k < KS * KS; ++k) acc[k] = 0.f;
    float bsum = 0.f;
    const int cells = N * PH * PW;
    const int per = (cells + gridDim.y - 1) / gridDim.y;
    const int c0 = blockIdx.y * per, c1 = min(cells, c0 + per);
    for (int c = c0 + threadIdx.x; c < c1; c += 256) {
        const int pw = c % PW; int t = c / PW; const int ph = t % PH; const int n = t / PH;
        const long i = (((long)n * Cout + co) * PH + ph) * PW + pw;
        float g = dy[i];
        if (drop != nullptr) { const float dm = drop[n * Cout + co]; g = (y[i] > 0.f) ? g * dm : 0.f; }
        else g = (y[i] > 0.f) ? g : 0.f;
        if (g == 0.f) continue;
        const int bi = idx[i];
        const int oh = ph * 2 + (bi >> 1), ow = pw * 2 + (bi & 1);
        const float* xp = x + (((long)n * Cin + ci) * H + oh) * W + ow;
        bsum += g;
#pragma unroll
        for (int r = 0; r < KS; ++r)
#pragma unroll
            for (int s = 0; s < KS; ++s) acc[r * KS + s] += g * xp[r * W + s];
    }
    __shared__ float red[4][KS * KS + 1];
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

// dx, LDS-tiled, shapes fixed at compile time (the reference's conv2: 10 -> 20 channels on 12x12): one
// workgroup per sample.  The sample's conv-output gradient is scattered into a ZERO-PADDED dense
// [Cout][OH+8][OW+8] LDS image (non-zero only at the pooling arg-max positions; the padding makes every
// filter tap in range, so the hot loop has no conditions), the layer's filters sit in LDS as well, and
// each thread produces one input ROW (ci, h, 0..W-1) in registers: per (co, r) it reads one padded
// gradient row (4 x ds_read_b128) and five taps for 5*W FMAs of straight-line code.
template <int CIN, int COUT, int HH, int WW>
__global__ __launch_bounds__(256) void conv5_dx_tiled_kernel(const float* __restrict__ w, const float* __restrict__ dy,
                                                             const float* __restrict__ y, const uint8_t* __restrict__ idx,
                                                             float* __restrict__ dx, const float* __restrict__ drop) {
    constexpr int OH = HH - KS + 1, OW = WW - KS + 1, PH = OH / 2, PW = OW / 2;
    constexpr int GH = OH + 2 * (KS - 1), GW = (OW + 2 * (KS - 1) + 3) / 4 * 4;    // padded image, rows 16-byte multiples
    static_assert(GW >= WW + KS - 1, "padded row must cover every tap of every output column");
    __shared__ __attribute__((aligned(16))) float G[COUT * GH * GW];
    __shared__ __attribute__((aligned(16))) float Wl[COUT * CIN * KS * KS];
    const int n = blockIdx.x;
    for (int i = threadIdx.x; i < COUT * GH * GW; i += 256) G[i] = 0.f;
    for (int i = threadIdx.x; i < COUT * CIN * KS * KS; i += 256) Wl[i] = w[i];
    __syncthreads();
    for (int i = threadIdx.x; i < COUT * PH * PW; i += 256) {
        const int pw = i % PW; int t = i / PW; const int ph = t % PH; const int co = t / PH;
        const long o = (long)n * COUT * PH * PW + i;
        float g = y[o] > 0.f ? dy[o] : 0.f;
        if (drop != nullptr) g *= drop[n * COUT + co];
        const int bi = idx[o];
        G[(co * GH + ph * 2 + (bi >> 1) + KS - 1) * GW + pw * 2 + (bi & 1) + KS - 1] = g;
    }
    __syncthreads();
    const int row = threadIdx.x;
    if (row >= CIN * HH) return;
    const int ci = row / HH, h = row - ci * HH;
    float acc[WW];
#pragma unroll
    for (int v = 0; v < WW; ++v) acc[v] = 0.f;
    for (int co = 0; co < COUT; ++co) {
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
    float* dst = dx + (((long)n * CIN + ci) * HH + h) * WW;
#pragma unroll
    for (int v = 0; v < WW; ++v) dst[v] = acc[v];
}

// dx: one thread per input element; walks the (co, r, s) taps and keeps those that hit an arg-max
__global__ __launch_bounds__(256) void conv5_dx_kernel(const float* __restrict__ w, const float* __restrict__ dy,
                                                       const float* __restrict__ y, const uint8_t* __restrict__ idx,
                                                       float* __restrict__ dx, int N, int Cin, int H, int W, int Cout,
                                                       const float* __restrict__ drop) {
    const int OH = H - KS + 1, OW = W - KS + 1, PH = OH / 2, PW = OW / 2;
    const long total = (long)N * Cin * H * W;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        int wv = (int)(i % W); long t = i / W;
        int h = (int)(t % H); t /= H;
        int ci = (int)(t % Cin); int n = (int)(t / Cin);
        float acc = 0.f;
        for (int co = 0; co < Cout; ++co) {
            const float dm = drop != nullptr ? drop[n * Cout + co] : 1.f;
            if (dm == 0.f) continue;
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
                    if (idx[o] == ((oh & 1) * 2 + (ow & 1)) && y[o] > 0.f) acc += dy[o] * dm * wp[r * KS + s];
                }
            }
        }
        dx[i] = acc;
    }
}

// y = act(x @ w^T + b): one thread per (row, out)
__global__ __launch_bounds__(256) void linear_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ b, float* __restrict__ y, int rows,
                                                         int in, int out, int relu) {
    const long total = (long)rows * out;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int o = (int)(i % out); const long r = i / out;
        const float* xp = x + r * in; const float* wp = w + (long)o * in;
        float acc = b != nullptr ? b[o] : 0.f;
        int k = 0;
        if (((in & 3) == 0)) {
            for (; k < in; k += 4) {
                const float4 a = *(const float4*)(xp + k), c = *(const float4*)(wp + k);
                acc += a.x * c.x + a.y * c.y + a.z * c.z + a.w * c.w;
            }
        } else {
            for (; k < in; ++k) acc += xp[k] * wp[k];
        }
        y[i] = relu ? fmaxf(acc, 0.f) : acc;
    }
}

// dx[r][k] = sum_o g[r][o] * w[o][k],  g = dy * (relu ? y>0 : 1)
__global__ __launch_bounds__(256) void linear_dx_kernel(const float* __restrict__ w, const float* __restrict__ y,
                                                        const float* __restrict__ dy, float* __restrict__ dx, int rows,
                                                        int in, int out, int relu) {
    const long total = (long)rows * in;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int k = (int)(i % in); const long r = i / in;
        float acc = 0.f;
        for (int o = 0; o < out; ++o) {
            float g = dy[r * out + o];
            if (relu && !(y[r * out + o] > 0.f)) g = 0.f;
            acc += g * w[(long)o * in + k];
        }
        dx[i] = acc;
    }
}

// dw[o][k] = sum_r g[r][o] * x[r][k] ; db[o] = sum_r g[r][o].  Block = (o, 64-wide k tile); the 4 waves take
// interleaved rows (8 independent loads in flight per thread - the loop is latency-bound, not
// bandwidth-bound) and are folded through LDS.  x rows are read coalesced over k; g[r][o] is a broadcast.
__global__ __launch_bounds__(256) void linear_dw_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                        const float* __restrict__ dy, float* __restrict__ dw,
                                                        float* __restrict__ db, int rows, int in, int out, int relu) {
    const int ktiles = (in + 63) / 64;
    const int o = blockIdx.x / ktiles, k = (blockIdx.x % ktiles) * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6;
    const bool kok = k < in;
    float acc = 0.f, bs = 0.f;
    int r = rg;
    for (; r + 28 < rows; r += 32) {
        float g[8], xv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const long rr = r + 4 * u;
            g[u] = dy[rr * out + o];
            if (relu && !(y[rr * out + o] > 0.f)) g[u] = 0.f;
            xv[u] = kok ? x[rr * in + k] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) { acc += g[u] * xv[u]; bs += g[u]; }
    }
    for (; r < rows; r += 4) {
        float g = dy[(long)r * out + o];
        if (relu && !(y[(long)r * out + o] > 0.f)) g = 0.f;
        acc += g * (kok ? x[(long)r * in + k] : 0.f);
        bs += g;
    }
    __shared__ float red[2][4][64];
    red[0][rg][threadIdx.x & 63] = acc;
    red[1][rg][threadIdx.x & 63] = bs;
    __syncthreads();
    if (rg == 0) {
        const int l = threadIdx.x;
        if (kok) dw[(long)o * in + k] = red[0][0][l] + red[0][1][l] + red[0][2][l] + red[0][3][l];
        if (l == 0 && blockIdx.x % ktiles == 0 && db != nullptr) db[o] = red[1][0][0] + red[1][1][0] + red[1][2][0] + red[1][3][0];
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
                         int N, int Cin, int H, int W, int Cout, const float* drop_mask, hipStream_t st) {
    const int PH = (H - KS + 1) / 2, PW = (W - KS + 1) / 2;
    hipLaunchKernelGGL(conv5_pool_relu_fwd_kernel, dim3(grid_for((long)N * Cout * PH * PW)), dim3(256), 0, st, x, w, b, y, idx,
                       N, Cin, H, W, Cout, drop_mask);
}

void conv5_pool_relu_bwd(const float* x, const float* w, const float* dy, const float* y, const uint8_t* idx,
                         float* dx, float* dw, float* db, int N, int Cin, int H, int W, int Cout,
                         const float* drop_mask, hipStream_t st) {
    // dw / db are accumulated with atomics: the caller passes them zero-initialised
    const int PH = (H - KS + 1) / 2, PW = (W - KS + 1) / 2;
    const int cells = N * PH * PW;
    int split = 512 / (Cout * Cin);
    if (split > (cells + 255) / 256) split = (cells + 255) / 256;
    if (split < 1) split = 1;
    hipLaunchKernelGGL(conv5_dw_kernel, dim3(Cout * Cin, split), dim3(256), 0, st, x, dy, y, idx, dw, db, N, Cin, H, W, Cout, drop_mask);
    if (dx != nullptr) {
        if (Cin == 10 && Cout == 20 && H == 12 && W == 12)   // the reference's conv2 (DP.py:83)
            hipLaunchKernelGGL((conv5_dx_tiled_kernel<10, 20, 12, 12>), dim3(N), dim3(256), 0, st, w, dy, y, idx, dx, drop_mask);
        else
            hipLaunchKernelGGL(conv5_dx_kernel, dim3(grid_for((long)N * Cin * H * W)), dim3(256), 0, st, w, dy, y, idx, dx, N, Cin,
                               H, W, Cout, drop_mask);
    }
}

void linear_f32_fwd(const float* x, const float* w, const float* b, float* y, int rows, int in, int out, int relu, hipStream_t st) {
    hipLaunchKernelGGL(linear_fwd_kernel, dim3(grid_for((long)rows * out)), dim3(256), 0, st, x, w, b, y, rows, in, out, relu);
}

void linear_f32_bwd(const float* x, const float* w, const float* y, const float* dy, float* dx, float* dw, float* db,
                    int rows, int in, int out, int relu, hipStream_t st) {
    if (dx != nullptr)
        hipLaunchKernelGGL(linear_dx_kernel, dim3(grid_for((long)rows * in)), dim3(256), 0, st, w, y, dy, dx, rows, in, out, relu);
    hipLaunchKernelGGL(linear_dw_kernel, dim3((unsigned)(out * ((in + 63) / 64))), dim3(256), 0, st, x, y, dy, dw, db, rows, in, out, relu);
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
