// Host-side launcher API of the gfx950 kernels.  Plain C++ (no torch headers) so
// the .hip translation units compile in seconds; bindings.cpp adapts tensors.
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>

namespace dpa {

// ---------------------------------------------------------------- switches (runtime.hip)
bool deterministic();            // DPA_DETERMINISTIC=1: single-writer routes instead of float atomics (bit-reproducible)
void set_deterministic(bool on);

// ---------------------------------------------------------------- conv / GEMM
// Implicit-GEMM convolution on NHWC bf16 activations and KRSC bf16 filters.
//   mode 0 (fprop):  y[n,ho,wo,k] = sum_{r,s,c} x[n, ho*stride-pad+r, wo*stride-pad+s, c] * w[k,r,s,c]
//   mode 1 (dgrad):  y[n,h,w,k]   = sum_{r,s,c} x[n, (h+pad-r)/stride, (w+pad-s)/stride, c] * w[k,r,s,c]
//                    (taps whose quotient is not integral / out of range contribute 0;
//                     x is dy, w is the filter with its K and C axes swapped, k runs
//                     over the forward conv's input channels)
// A Linear layer is the R=S=1, H=W=1 case (N = rows).
struct ConvArgs {
    const void* x;      // bf16 [N, H, W, C]
    const void* w;      // bf16 [K, R, S, C]
    void* y;            // bf16 [N, Ho, Wo, K]
    const float* bias;  // fp32 [K] or nullptr
    float* stats;       // fp32 [stats_rows][2][K]: (sum, sumsq) per channel are ADDED (atomics) into replica
                        // (pixel tile mod stats_rows); must be zero before the launch; or nullptr
    int stats_rows;
    const void* addend; // bf16 [N, Ho, Wo, K] added to the output (residual-gradient fusion), or nullptr
    const unsigned char* addend_mask;  // optional bit mask (1 bit per addend element, 8 per byte): masked-out
                                       // elements of addend count as 0 (ReLU gradient of the residual join)
    int N, H, W, C, K, R, S, Ho, Wo;
    int stride, pad;
    int asym, stride_w, pad_w;   // asym != 0 (fprop only): the WIDTH axis uses stride_w / pad_w instead of stride / pad
                                 // (the pair-packed stem: 7 x 4 taps, stride 2 x 1, pad 3 x 2 - ops/block.py)
    int mode;           // 0 fprop, 1 dgrad
    int act;            // 0 none, 1 relu, 2 gelu(tanh)
    // Fused BatchNorm epilogues (fprop only), c = this conv's bf16-rounded output, never written when epi != 0:
    //   epi 1: y = relu(c * ep_a[k] + ep_b[k] + ep_src), mask bits -> ep_mask_out
    //   epi 2: no y; g = ep_src under ep_mask_in; (sum g, sum g*c) per channel are ADDED into ep_stats
    //          ([ep_stats_rows][2][K] fp32, zero before the launch)
    //   epi 3: y = ep_a[k] * g + ep_b[k] * c + ep_c[k]
    // y == nullptr with epi == 0: statistics-only pass (only `stats` is produced).
    int epi;
    const float* ep_a; const float* ep_b; const float* ep_c;
    const void* ep_src;                 // bf16 [N, Ho, Wo, K]
    const unsigned char* ep_mask_in;
    unsigned char* ep_mask_out;
    float* ep_stats; int ep_stats_rows;
    //   epi 5 (dgrad, stride 1): y = dgrad * gelu'(ep_src), ep_src = the input of the tanh-GELU in front of this layer
    void* y_act;        // fprop, epi 0: optional SECOND output gelu_tanh(y) (bf16, y's shape) - the activated copy the next
                        // GEMM reads while y itself is kept for the GELU's backward; nullptr = none
};
void conv_igemm(const ConvArgs& a, hipStream_t st);
int conv_stats_rows(long M);  // replica rows of the stats buffer
// 256 x 256-tile GEMM main loop (gemm_wide.hip) for the plain-GEMM launches (1x1, stride 1) whose grid fills the chip
// with such tiles; conv_igemm() routes to it when wide_gemm_eligible().  set_wide_gemm_min_tiles(0) turns it off.
bool wide_gemm_supported(const ConvArgs& a);
bool wide_gemm_eligible(const ConvArgs& a);
void gemm_wide(const ConvArgs& a, hipStream_t st);
void set_wide_gemm_min_tiles(int n);
void set_wide_gemm_splitk(int n);   // 0 = chosen per launch by a cost model, n > 0 = forced (tests)
int wide_gemm_min_tiles();
void set_wide_gemm_bn(int n);       // output channels per tile: 0 = chosen per launch, 256 or 192 = forced (tests, benchmarks)
// Temporary device memory of the kernels' launchers (gemm_wide's split-K slabs): alloc(bytes, stream) returns a block that
// may be used by work enqueued on `stream` until free(ptr); the framework registers torch's caching allocator
typedef void* (*WorkspaceAlloc)(size_t bytes, hipStream_t stream);
typedef void (*WorkspaceFree)(void* ptr);
void set_workspace_allocator(WorkspaceAlloc alloc, WorkspaceFree free_);

// halo-tiled 3 x 3 / stride 1 / pad 1 forward + input gradient for 64 -> 64 channels on 56-wide maps (conv3x3_halo.hip);
// conv_igemm routes eligible launches to it
bool conv3x3_halo_ok(const ConvArgs& a);
void conv3x3_halo(const ConvArgs& a, hipStream_t st);
void set_conv3x3_halo(int on);      // tests / A-B runs: route eligible launches to the halo kernel (1) or to conv_igemm (0)
int conv3x3_halo_enabled();

// wgrad: dw[k,r,s,c] += sum_{n,ho,wo} dy[n,ho,wo,k] * x[n, ho*stride-pad+r, wo*stride-pad+s, c]
// dw is fp32 and is accumulated into with float atomics (split over the pixel axis).
struct WgradArgs {
    const void* x;    // bf16 [N, H, W, C]
    const void* dy;   // bf16 [N, Ho, Wo, K]
    float* dw;        // fp32 [K, R, S, C]
    int N, H, W, C, K, R, S, Ho, Wo;
    int stride, pad;
    int asym, stride_w, pad_w;   // as in ConvArgs
    float* dbias;     // optional fp32 [K]: += column sums of dy (bias gradient), computed on the MFMA beside dw
};
void conv_wgrad(const WgradArgs& a, hipStream_t st);
// halo-tiled 3 x 3 / stride 1 / pad 1 weight gradient (conv_wgrad_halo.hip); conv_wgrad routes eligible problems to it
bool conv_wgrad_halo_ok(const WgradArgs& a);
void conv_wgrad_halo(const WgradArgs& a, hipStream_t st);
// true when conv_wgrad's kernel for this problem also produces a.dbias (else the caller runs colsum_bf16)
bool conv_wgrad_fuses_bias(const WgradArgs& a);

// Wide-tile weight gradient of Linear-shaped problems, several layers over the same tokens in ONE launch
// (wgrad_wide.hip): 256 x 256 tiles, token split folded in a fixed order (no float atomics on dw).
constexpr int WGRAD_WIDE_MAX = 8;                    // layers per launch
bool wgrad_wide_supported(const WgradArgs& a);       // can run there
bool wgrad_wide_eligible(const WgradArgs& a);        // ... and pays (DPA_WGRAD_WIDE=0: never)
bool wgrad_wide_alone_pays(const WgradArgs& a);      // ... even in a launch of its own: conv_wgrad() then routes it here
long wgrad_wide_tiles(const WgradArgs& a);
void wgrad_wide_group(const WgradArgs* probs, int nprob, hipStream_t st);
void set_wgrad_wide_splitk(int n);                   // 0 = chosen per launch, n > 0 = forced (tests)
// temporary device memory of a launcher, usable by work enqueued on `st` until released (the registered allocator, or a
// grow-only block per stream)
void* launcher_workspace(size_t bytes, hipStream_t st);
void launcher_workspace_release(void* p);

// Backward of "1x1 conv (CM -> CO = 4 CM channels) -> BatchNorm -> ReLU mask" in one kernel (tail_bwd.hip):
//   g = cA * (dout under kout) + cB * craw + cC  (never stored);  dw += g^T xin;  dx = g w;
//   optionally the reduction pass of the BatchNorm behind dx: stats += (sum dx', sum dx' * sx), dx' = dx under smask
struct TailBwdArgs {
    const void* dout;            // bf16 [P][CO]
    const unsigned char* kout;   // 1 bit per element of dout, or nullptr (no mask)
    const void* craw;            // bf16 [P][CO]  the BatchNorm's input (raw conv output)
    const float* cA; const float* cB; const float* cC;   // fp32 [CO]  (bn_bwd_finalize)
    const void* xin;             // bf16 [P][CM]  the conv's input
    const void* wt;              // bf16 [CM][CO] the conv's filter with K and C swapped
    float* dw;                   // fp32 [CO][CM] accumulated into (atomics)
    void* dx;                    // bf16 [P][CM]
    const void* sx; const unsigned char* smask; float* stats; int stats_rows;   // all four or none ([rows][2][CM])
    long P; int CM, CO;
};
bool tail_bwd_supported(const TailBwdArgs& a);
void tail_bwd(const TailBwdArgs& a, hipStream_t st);

// [K, RS, C] bf16 -> [C, RS, K] bf16 (filter for dgrad)
void transpose_filter(const void* w, void* wt, int K, int RS, int C, hipStream_t st);
// all filters of a model in one launch; table [nconv][6] int64 on device = (src off, dst off, K, RS, C, first tile)
void transpose_filters_batched(const void* src, void* dst, const long* table, int nconv, long total_tiles, hipStream_t st);
int transpose_tile();   // tile edge the table's tile counts are made of
// column sums of a bf16 [M, K] matrix into fp32 [K] (bias gradient), accumulates.
// wide matrices (K > 2048) need a [colsum_workspace_rows][K] fp32 scratch buffer
int colsum_workspace_rows(long M, int K);
void colsum_bf16(const void* x, float* out, long M, int K, float* workspace, hipStream_t st);

// ---------------------------------------------------------------- batch norm (NHWC, rows = N*H*W)
// partial (sum,sumsq) per block of rows -> stats[rows][2][C]
int bn_stats_rows(long P);
void bn_stats(const void* x, float* stats, long P, int C, hipStream_t st);
// out[o][:] = sum of in rows [o*chunk, (o+1)*chunk); cols floats per row (fold a tall partial buffer)
void fold_rows(const float* in, int rows, int cols, int chunk, float* out, hipStream_t st);
// reduce partials, produce per-channel scale/shift + saved mean/invstd, update running stats
void bn_finalize(float* stats, int rows, long P, int C, const float* gamma, const float* beta,
                 float* running_mean, float* running_var, float momentum, float eps,
                 float* mean, float* invstd, float* scale, float* shift, int reset_stats, hipStream_t st);
// y = act(x*scale[c] + shift[c] (+ residual))
// relu: optionally also emits the bit mask (y > 0), one byte per 8 elements, that the backward kernels read
void bn_apply(const void* x, const void* residual, void* y, const float* scale, const float* shift,
              unsigned char* mask, long P, int C, int relu,
              const float* rscale, const float* rshift /* optional per-channel affine of the residual, or nullptr */,
              hipStream_t st);
// backward pass 1: g = dy * (relu ? mask : 1); partial sums of g and g*xhat per block of rows
// (replicas > 0: added with atomics into a zeroed [replicas][2][C] buffer instead)
void bn_bwd_stats(const void* dy, const unsigned char* mask, const void* x, const float* mean, const float* invstd,
                  float* stats, long P, int C, int relu, int replicas, hipStream_t st);
// reduce partials -> dgamma, dbeta (accumulate=0: overwrite) and coefficients A,B,C0 with dx = A*g + B*x + C0
// raw_x: the second statistic is sum(g * x) (raw conv output) instead of sum(g * xhat); converted here
void bn_bwd_finalize(float* stats, int rows, long P, int C, const float* gamma, const float* mean,
                     const float* invstd, float* dgamma, float* dbeta, int accumulate, float* coefA, float* coefB,
                     float* coefC, int reset_stats, int raw_x, hipStream_t st);
// backward pass 2: dx = A*g + B*x + C0
void bn_bwd_apply(const void* dy, const unsigned char* mask, const void* x, const float* coefA, const float* coefB,
                  const float* coefC, void* dx, long P, int C, int relu, hipStream_t st);
// out = dy where the mask bit is set, else 0 (n % 8 == 0)
void relu_mask_mul(const void* dy, const unsigned char* mask, void* out, long n, hipStream_t st);

// ---------------------------------------------------------------- pooling / layout
// stem: y = maxpool3x3/2(relu(x * scale + shift)) without materialising the activation; idx = winning tap (9: none)
void bn_relu_maxpool_fwd(const void* x, const float* scale, const float* shift, void* y, unsigned char* idx,
                         int N, int H, int W, int C, int Ho, int Wo, hipStream_t st);
// ... and its BatchNorm backward, gathering the pooled gradient through idx (no un-pooled gradient tensor)
void pool_bn_bwd_stats(const void* dp, const unsigned char* idx, const void* x, const float* mean, const float* invstd,
                       float* stats, int N, int H, int W, int C, int Ho, int Wo, int replicas, hipStream_t st);
void pool_bn_bwd_apply(const void* dp, const unsigned char* idx, const void* x, const float* coefA, const float* coefB,
                       const float* coefC, void* dx, int N, int H, int W, int C, int Ho, int Wo, hipStream_t st);
void maxpool3x3s2_fwd(const void* x, void* y, uint8_t* idx, int N, int H, int W, int C, int Ho, int Wo, hipStream_t st);
void maxpool3x3s2_bwd(const void* dy, const uint8_t* idx, void* dx, int N, int H, int W, int C, int Ho, int Wo, hipStream_t st);
void avgpool_fwd(const void* x, void* y, int N, int HW, int C, hipStream_t st);
void avgpool_bwd(const void* dy, void* dx, int N, int HW, int C, hipStream_t st);
// NCHW (fp32 or bf16) -> NHWC bf16 with channel padding to Cpad
void nchw_to_nhwc_pad(const void* x, int x_is_bf16, void* y, int N, int C, int H, int W, int Cpad, hipStream_t st);
// pair-packed stem (elementwise.hip): NCHW image (C <= 3, W even) -> [N, H, W/2, 8] pixel pairs; filter [K][R][7][8] ->
// [K][R][4][8]; filter gradient back (dw += unpack(dwp)).  KR = K * R.
void nchw_to_pairs(const void* x, int x_is_bf16, void* y, int N, int C, int H, int W, hipStream_t st);
void stem_pack_filter(const void* w, void* wp, int KR, int C, hipStream_t st);
void stem_unpack_dw(const float* dwp, float* dw, int KR, int C, hipStream_t st);
// elementwise on bf16: y = a + b ; relu variants
void add_bf16(const void* a, const void* b, void* y, long n, hipStream_t st);
void add_relu_fwd(const void* a, const void* b, void* y, long n, hipStream_t st);
void relu_bwd(const void* dy, const void* y, void* dx, long n, hipStream_t st);
void gelu_fwd(const void* x, void* y, long n, hipStream_t st);
void gelu_bwd(const void* dy, const void* x, void* dx, long n, hipStream_t st);

// ---------------------------------------------------------------- loss
// logits [B, V] (bf16 or fp32, row stride ld); loss_sum += sum_b (lse_b - logit_b[target_b]) * scale
void softmax_ce_fwd(const void* logits, int is_bf16, const int64_t* targets, float* lse, float* loss_sum,
                    long B, int V, long ld, float scale, int64_t ignore_index, hipStream_t st);
// dlogits = (softmax - onehot) * (*gscale_ptr) * scale ; may alias logits
void softmax_ce_bwd(const void* logits, int is_bf16, const int64_t* targets, const float* lse, const float* gscale_ptr,
                    void* dlogits, long B, int V, long ld, float scale, int64_t ignore_index, hipStream_t st);
// one-pass form (loss.hip): the bf16 logits are overwritten by their gradient for an upstream gradient of 1;
// scale_unless_one() multiplies by the real upstream gradient when it is not 1 (device-side test, no host sync)
bool softmax_ce_fused_supported(int is_bf16, int V, long ld, const void* logits);
void softmax_ce_fused(void* logits, const int64_t* targets, float* loss_sum, long B, int V, long ld, float scale,
                      int64_t ignore_index, hipStream_t st);
void scale_unless_one(void* d_bf16, const float* g, long n, hipStream_t st);

// ---------------------------------------------------------------- optimizer
// fused SGD-momentum over a flat buffer:
//   g' = g*grad_scale + wd*p ; m = mu*m + g' ; p -= lr*m ; p_lp = bf16(p) ; g = 0 (if zero_grad)
// optional segment table: element i uses seg_wd[s] for the first s with seg_ends[s] > i (device arrays)
void sgd_momentum(float* p, float* g, float* m, void* p_lp, long n, float lr, float mu, float wd,
                  float grad_scale, int zero_grad, int nesterov, const long* seg_ends, const float* seg_wd, int nseg,
                  const float* grad_scale_dev /* optional device scalar multiplied into grad_scale (clip coefficient) */,
                  hipStream_t st);
void adamw(float* p, float* g, float* m, float* v, void* p_lp, long n, float lr, float b1, float b2, float eps,
           float wd, float grad_scale, float bc1, float bc2, int zero_grad, const float* grad_scale_dev, hipStream_t st);
void cast_f32_to_bf16(const float* x, void* y, long n, hipStream_t st);
void cast_bf16_to_f32(const void* x, float* y, long n, hipStream_t st);
void scale_f32(float* x, long n, float s, hipStream_t st);

// ---------------------------------------------------------------- transformer pieces
void layernorm_fwd(const void* x, const float* g, const float* b, void* y, float* mean, float* rstd,
                   long rows, int C, float eps, hipStream_t st);
void layernorm_bwd(const void* dy, const void* x, const float* g, const float* mean, const float* rstd,
                   void* dx, float* dg_partial, float* db_partial, int partial_rows, long rows, int C,
                   const void* addend /* bf16 [rows, C] added to dx, or nullptr */, hipStream_t st);
int layernorm_partial_rows(long rows);
void reduce_partials(const float* partial, int rows, int C, float* out, int accumulate, hipStream_t st);
// partial = [2][rows][C]: both halves folded by one launch into out_a / out_b
void reduce_partials2(const float* partial, int rows, int C, float* out_a, float* out_b, int accumulate, hipStream_t st);
void embedding_fwd(const int64_t* tok, const void* wte, const void* wpe, void* y, long B, int T, int C, long V, hipStream_t st);
void embedding_bwd(const int64_t* tok, const void* dy, float* dwte, float* dwpe, long B, int T, int C, long V, hipStream_t st);
// causal flash attention, qkv packed [B, T, 3, H, D] bf16, D = 64
// drop_p > 0: attention dropout, element (b, h, q, k) kept by the counter hash at index ((b*H + h)*T + q)*T + k of the
// stream (seed, offset) - dropout_mask() with n = B*H*T*T reproduces the mask
void set_attention_fwd_pipe(int on);   // 1 = the software-pipelined half-tile forward kernel (D = 64, no dropout); default 0 / DPA_ATTN_FWD_PIPE
void attention_fwd(const void* qkv, void* out, float* lse, int B, int T, int H, int D, float drop_p, uint64_t seed,
                   uint64_t offset, hipStream_t st);
void attention_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, float* delta,
                   int B, int T, int H, int D, float drop_p, uint64_t seed, uint64_t offset, hipStream_t st);
// y = addend + x * keep(seed, offset, i) / (1 - p)  (bf16, n % 8 == 0; addend optional): element dropout with the mask
// drawn in line, fused with the residual add that follows it in a transformer block
void dropout_bf16(const void* x, const void* addend, void* y, long n, float p, uint64_t seed, uint64_t offset, hipStream_t st);

// ---------------------------------------------------------------- small MNIST CNN (fp32, NCHW; reference model)
// y = dropout2d(relu(maxpool2(conv5x5(x) + b))) fused ; idx = argmax position in the 2x2 window.  drop_p > 0: one keep/drop
// draw per (sample, channel) from the counter hash (seed, offset + *step * 2000003, n * Cout + co); nothing is stored
void conv5_pool_relu_fwd(const float* x, const float* w, const float* b, float* y, uint8_t* idx,
                         int N, int Cin, int H, int W, int Cout, float drop_p, uint64_t seed, uint64_t offset,
                         const int64_t* step, hipStream_t st);
// dw / db are ADDED to (atomics); gscale = 1/(1-p) of the forward's dropout (a dropped channel has y == 0 by itself)
void conv5_pool_relu_bwd(const float* x, const float* w, const float* dy, const float* y, const uint8_t* idx,
                         float* dx, float* dw, float* db, int N, int Cin, int H, int W, int Cout,
                         float gscale, hipStream_t st);
// y = dropout(act(x @ w^T + b)) fp32 small GEMM (rows x in -> rows x out); element dropout drawn in line as above
void linear_f32_fwd(const float* x, const float* w, const float* b, float* y, int rows, int in, int out, int relu,
                    float drop_p, uint64_t seed, uint64_t offset, const int64_t* step, hipStream_t st);
// logp = log_softmax(x @ w^T + b), out <= 64
void linear_lsm_fwd(const float* x, const float* w, const float* b, float* logp, int rows, int in, int out, hipStream_t st);
// one launch: dx (may be nullptr) | dw, db (overwritten, or added to with accumulate).  lsm: y holds log-probabilities and
// dy is the gradient w.r.t. them (log-softmax backward in line); else relu / gscale as for the conv
// lsm: 0 plain, 1 y = log-probabilities and dy their gradient, 2 additionally the mean-NLL backward: dy = ONE float (the
// loss gradient), tgt = the targets (out <= 16 only)
void linear_f32_bwd(const float* x, const float* w, const float* y, const float* dy, float* dx, float* dw, float* db,
                    int rows, int in, int out, int relu, float gscale, int lsm, int accumulate, const int64_t* tgt, hipStream_t st);
// logp = log_softmax(x @ w^T + b), loss = -mean_r logp[r][t_r] in one single-workgroup launch (out <= 16, in <= 448: LDS);
// step (optional): device counter incremented by one
void linear_lsm_nll_fwd(const float* x, const float* w, const float* b, const int64_t* t, float* logp, float* loss,
                        int64_t* step, int rows, int in, int out, hipStream_t st);
void log_softmax_fwd(const float* x, float* y, int rows, int V, hipStream_t st);
void log_softmax_bwd(const float* dy, const float* y, float* dx, int rows, int V, hipStream_t st);
void nll_fwd(const float* logp, const int64_t* t, float* loss, int rows, int V, hipStream_t st);
void nll_bwd(const float* g, const int64_t* t, float* dlogp, int rows, int V, hipStream_t st);
void dropout_mask(float* mask, long n, float p, uint64_t seed, uint64_t offset, const int64_t* step, hipStream_t st);
void counter_add(int64_t* c, int64_t v, hipStream_t st);
// dst[i] += sum_{k<nsrc} src[k*stride + i], i < n (fp32; local step of the direct all-reduce)
void reduce_shards(float* dst, const float* src, int nsrc, long stride, long n, hipStream_t st);   // device-resident step counter (graph-replay safe)
void mul_f32(const float* a, const float* b, float* y, long n, hipStream_t st);
// dst[i] += (*g) * src[i], g a device scalar
void axpy_dev(float* dst, const float* src, long n, const float* g, hipStream_t st);
// dst[i] += sum_{k<nsrc} bf16 src[k*stride + i] in fp32; out[i] = bf16(dst[i])  (bf16 wire format of the direct all-reduce)
void reduce_shards_bf16(float* dst, const void* src, int nsrc, long stride, long n, void* out, hipStream_t st);

// ---------------------------------------------------------------- peer-mapped all-reduce (peer_allreduce.hip, peer_comm.cpp)
constexpr int PEER_MAX_RANKS = 8;          // one xGMI-connected node
constexpr int PEER_MAX_BLOCKS = 64;        // workgroups of one launch = rows of the signal arrays
constexpr int PEER_ERR_WORD = 2 * PEER_MAX_BLOCKS * 16;       // signal block: A[blocks][16], B[blocks][16], error word, ...
constexpr int PEER_SIGNAL_WORDS = PEER_ERR_WORD + 64;
constexpr unsigned PEER_SEQ_WINDOW = 1u << 20;  // a slot value counts as "raised" only within this many launches above the awaited one
struct PeerArgs {
    int rank, world;
    unsigned seq;                          // launch number of the group (same on every rank), counts from 1
    float scale;                           // applied to the sum (1 = plain sum)
    long n;                                // fp32 elements of the tensor
    long shard;                            // elements per rank's shard, a multiple of 8, world * shard >= n
    float* user;                           // the tensor, 16-byte aligned, reduced in place
    void* stage_self;                      // this launch's half of my staging buffer
    void* stage_peer[PEER_MAX_RANKS - 1];  // ... of the other ranks, in rank order (peer-mapped)
    unsigned* sig_self;                    // my signal block (uncached memory)
    unsigned* sig_peer[PEER_MAX_RANKS - 1];
    unsigned long long timeout_ticks;      // bound of every poll, in 100 MHz ticks
};
int peer_allreduce_blocks(long n, int max_blocks);
void peer_allreduce(const PeerArgs& a, int blocks, int bf16_wire, hipStream_t st);

}  // namespace dpa
