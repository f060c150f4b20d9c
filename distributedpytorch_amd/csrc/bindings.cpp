// Python bindings: adapt torch tensors to the raw-pointer launcher API of kernels.h.
// Every entry point validates dtype / shape / contiguity / alignment on the host
// BEFORE launching (an out-of-bounds kernel access can reset the whole GPU), and
// launches on PyTorch's current HIP stream so ordering with autograd is implicit.
#include <torch/extension.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <c10/hip/HIPCachingAllocator.h>

#include "kernels.h"

namespace {

using torch::Tensor;

inline hipStream_t cur_stream() { return c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream(); }

#define CHECK_DEV(t) TORCH_CHECK((t).is_cuda(), #t " must be a GPU tensor")
#define CHECK_CONTIG(t) TORCH_CHECK((t).is_contiguous(), #t " must be contiguous")
#define CHECK_BF16(t) do { CHECK_DEV(t); CHECK_CONTIG(t); TORCH_CHECK((t).scalar_type() == at::kBFloat16, #t " must be bf16"); \
                           TORCH_CHECK(((uintptr_t)(t).data_ptr()) % 16 == 0, #t " must be 16-byte aligned"); } while (0)
#define CHECK_F32(t) do { CHECK_DEV(t); CHECK_CONTIG(t); TORCH_CHECK((t).scalar_type() == at::kFloat, #t " must be fp32"); } while (0)
#define CHECK_I64(t) do { CHECK_DEV(t); CHECK_CONTIG(t); TORCH_CHECK((t).scalar_type() == at::kLong, #t " must be int64"); } while (0)

struct Guard {
    c10::hip::HIPGuardMasqueradingAsCUDA g;
    explicit Guard(const Tensor& t) : g(t.device()) {}
};

// ---------------------------------------------------------------- conv
// x [N,H,W,C] bf16, w [K,R,S,C] bf16 -> y [N,Ho,Wo,K] bf16 (+ optional BN partial stats [rows,2,K])
// DPA_DETERMINISTIC (csrc/runtime.hip): statistics that a conv epilogue adds with float atomics into a few replica rows go to
// one row per contributing workgroup instead (a single add per entry: order-free) and are folded in a fixed order into
// row 0 of the caller's accumulator afterwards.
struct DetStats {
    Tensor tmp, user;
    void redirect(float*& ptr, int& rows, const Tensor& user_buf, int64_t slots) {
        if (!dpa::deterministic() || ptr == nullptr) return;
        user = user_buf;
        tmp = torch::zeros({slots, user_buf.size(1), user_buf.size(2)}, user_buf.options());
        ptr = tmp.data_ptr<float>(); rows = (int)slots;
    }
    void finish() {
        if (user.defined()) user.select(0, 0).add_(tmp.sum(0));
    }
};
static int64_t det_slots(int64_t rows_out, int64_t stride) { return ((rows_out + 63) / 64 + 1) * stride * stride; }

// `addend` (bf16, the output's shape): y = conv(x) + bias + addend - a residual connection in the GEMM epilogue.
// stride_w / pad_w >= 0: the width axis has its own stride / padding (the pair-packed stem convolution).
std::vector<Tensor> conv_fprop(const Tensor& x, const Tensor& w, const c10::optional<Tensor>& bias, int64_t stride,
                               int64_t pad, int64_t act, bool want_stats, const c10::optional<Tensor>& stats_buf,
                               const c10::optional<Tensor>& addend, bool gelu_twin, int64_t stride_w, int64_t pad_w, int64_t out_w) {
    CHECK_BF16(x); CHECK_BF16(w);
    TORCH_CHECK(x.dim() == 4 && w.dim() == 4, "conv_fprop: x and w must be 4-D (NHWC / KRSC)");
    const int64_t N = x.size(0), H = x.size(1), W = x.size(2), C = x.size(3);
    const int64_t K = w.size(0), R = w.size(1), S = w.size(2);
    TORCH_CHECK(w.size(3) == C, "conv_fprop: channel mismatch ", w.size(3), " vs ", C);
    TORCH_CHECK(C % 8 == 0 && K % 8 == 0, "conv_fprop: C and K must be multiples of 8 (pad channels)");
    TORCH_CHECK(stride >= 1 && pad >= 0, "bad stride/pad");
    const bool asym = stride_w >= 0 || pad_w >= 0;
    TORCH_CHECK(!asym || (stride_w >= 1 && pad_w >= 0), "conv_fprop: stride_w and pad_w go together");
    const int64_t Ho = (H + 2 * pad - R) / stride + 1,
                  Wo_full = asym ? (W + 2 * pad_w - S) / stride_w + 1 : (W + 2 * pad - S) / stride + 1;
    // out_w (asym only): fewer output columns than the padding would allow (the packed stem pads 2 pairs on both sides
    // but its last output column needs only one)
    TORCH_CHECK(out_w < 0 || (asym && out_w >= 1 && out_w <= Wo_full), "conv_fprop: out_w");
    const int64_t Wo = out_w > 0 ? out_w : Wo_full;
    TORCH_CHECK(Ho > 0 && Wo > 0, "conv_fprop: empty output");
    TORCH_CHECK(x.numel() < (1ll << 30) && w.numel() < (1ll << 30) && N * Ho * Wo * K < (1ll << 31) - 16,
                "conv_fprop: operands must have fewer than 2^31 elements (32-bit gather offsets)");
    Guard g(x);
    Tensor y = torch::empty({N, Ho, Wo, K}, x.options());
    dpa::ConvArgs a{};
    a.x = x.data_ptr(); a.w = w.data_ptr(); a.y = y.data_ptr();
    a.bias = nullptr;
    if (bias.has_value() && bias->defined()) {
        CHECK_F32((*bias)); TORCH_CHECK(bias->numel() == K, "bias size");
        TORCH_CHECK(!want_stats, "stats are computed on the bias-free output");
        a.bias = bias->data_ptr<float>();
    }
    Tensor stats;
    DetStats det;
    a.stats = nullptr;
    const int64_t M = N * Ho * Wo;
    a.stats_rows = 0; a.addend = nullptr;
    if (want_stats) {
        TORCH_CHECK(act == 0, "stats need a linear epilogue");
        if (stats_buf.has_value() && stats_buf->defined()) {   // persistent buffer: zero on entry, re-zeroed by bn_finalize
            stats = *stats_buf;
            CHECK_F32(stats);
            TORCH_CHECK(stats.dim() == 3 && stats.size(1) == 2 && stats.size(2) == K && stats.size(0) >= 1, "stats_buf must be [rows,2,K]");
        } else {
            stats = torch::zeros({dpa::conv_stats_rows(M), 2, K}, x.options().dtype(at::kFloat));
        }
        a.stats = stats.data_ptr<float>();
        a.stats_rows = stats.size(0);
        det.redirect(a.stats, a.stats_rows, stats, det_slots(M, 1));
    }
    if (addend.has_value() && addend->defined()) {
        CHECK_BF16((*addend));
        TORCH_CHECK(addend->numel() == y.numel() && !want_stats && act == 0, "conv_fprop: addend must match the output; no stats / activation with it");
        a.addend = addend->data_ptr();
    }
    a.N = N; a.H = H; a.W = W; a.C = C; a.K = K; a.R = R; a.S = S; a.Ho = Ho; a.Wo = Wo;
    a.stride = stride; a.pad = pad; a.mode = 0; a.act = act;
    a.asym = asym ? 1 : 0; a.stride_w = asym ? (int)stride_w : 0; a.pad_w = asym ? (int)pad_w : 0;
    Tensor y_act;
    if (gelu_twin) {      // second output gelu_tanh(y): the next GEMM's operand, y itself stays for the GELU's backward
        TORCH_CHECK(!want_stats && act == 0, "conv_fprop: gelu_twin goes with a linear epilogue and no statistics");
        y_act = torch::empty_like(y);
        a.y_act = y_act.data_ptr();
    }
    dpa::conv_igemm(a, cur_stream());
    det.finish();
    if (want_stats) return {y, stats};
    if (gelu_twin) return {y, y_act};
    return {y};
}

// ---- recompute-instead-of-store: a 1x1 conv in front of a BatchNorm, with the BatchNorm fused into the
//      conv epilogue (csrc/kernels.h, ConvArgs::epi).  The raw conv output is never written. ----
static dpa::ConvArgs fused_args(const Tensor& x, const Tensor& w, int64_t stride, int64_t pad, const char* who) {
    CHECK_BF16(x); CHECK_BF16(w);
    TORCH_CHECK(x.dim() == 4 && w.dim() == 4 && w.size(3) == x.size(3), who, ": NHWC x / KRSC w expected");
    const int64_t N = x.size(0), H = x.size(1), W = x.size(2), C = x.size(3);
    const int64_t K = w.size(0), R = w.size(1), S = w.size(2);
    const int64_t Ho = (H + 2 * pad - R) / stride + 1, Wo = (W + 2 * pad - S) / stride + 1;
    TORCH_CHECK(C % 8 == 0 && K % 8 == 0 && Ho > 0 && Wo > 0, who, ": bad geometry");
    TORCH_CHECK(x.numel() < (1ll << 30) && w.numel() < (1ll << 30) && N * Ho * Wo * K < (1ll << 31) - 16, who, ": operands too large");
    dpa::ConvArgs a{};
    a.x = x.data_ptr(); a.w = w.data_ptr();
    a.N = N; a.H = H; a.W = W; a.C = C; a.K = K; a.R = R; a.S = S; a.Ho = Ho; a.Wo = Wo;
    a.stride = stride; a.pad = pad; a.mode = 0; a.act = 0;
    return a;
}
static void check_stats_buf(const Tensor& sb, int64_t K) {
    CHECK_F32(sb);
    TORCH_CHECK(sb.dim() == 3 && sb.size(1) == 2 && sb.size(2) == K && sb.size(0) >= 1, "stats buffer must be [rows,2,K]");
}
static void check_like_output(const Tensor& t, const dpa::ConvArgs& a, const char* what) {
    CHECK_BF16(t);
    TORCH_CHECK(t.numel() == (int64_t)a.N * a.Ho * a.Wo * a.K, what, " must have the conv output's shape");
}
static const uint8_t* check_mask(const Tensor& m, const dpa::ConvArgs& a) {
    TORCH_CHECK(m.is_cuda() && m.scalar_type() == at::kByte && m.is_contiguous() &&
                m.numel() * 8 == (int64_t)a.N * a.Ho * a.Wo * a.K, "mask must be uint8 with one bit per output element");
    return m.data_ptr<uint8_t>();
}

// statistics-only pass: (sum, sumsq) of conv(x, w) per channel are ADDED into stats_buf; nothing else is written
void conv_fprop_stats(const Tensor& x, const Tensor& w, int64_t stride, int64_t pad, Tensor& stats_buf) {
    dpa::ConvArgs a = fused_args(x, w, stride, pad, "conv_fprop_stats");
    check_stats_buf(stats_buf, a.K);
    Guard g(x);
    a.y = nullptr; a.stats = stats_buf.data_ptr<float>(); a.stats_rows = stats_buf.size(0);
    DetStats det;
    det.redirect(a.stats, a.stats_rows, stats_buf, det_slots((int64_t)a.N * a.Ho * a.Wo, 1));
    dpa::conv_igemm(a, cur_stream());
    det.finish();
}

// out = relu(conv(x, w) * scale + shift + residual), plus its 1-bit mask
std::vector<Tensor> conv_bn_res_relu(const Tensor& x, const Tensor& w, int64_t stride, int64_t pad, const Tensor& scale,
                                     const Tensor& shift, const Tensor& residual) {
    dpa::ConvArgs a = fused_args(x, w, stride, pad, "conv_bn_res_relu");
    CHECK_F32(scale); CHECK_F32(shift);
    TORCH_CHECK(scale.numel() == a.K && shift.numel() == a.K, "conv_bn_res_relu: scale/shift size");
    check_like_output(residual, a, "residual");
    Guard g(x);
    Tensor y = torch::empty({a.N, a.Ho, a.Wo, a.K}, x.options());
    Tensor mask = torch::empty({y.numel() / 8}, x.options().dtype(at::kByte));
    a.y = y.data_ptr(); a.epi = 1; a.ep_a = scale.data_ptr<float>(); a.ep_b = shift.data_ptr<float>();
    a.ep_src = residual.data_ptr(); a.ep_mask_out = mask.data_ptr<uint8_t>();
    dpa::conv_igemm(a, cur_stream());
    return {y, mask};
}

// BN backward, reduction pass on the recomputed conv output: adds (sum g, sum g*c) into stats_buf, g = dout under mask
void conv_bnb_stats(const Tensor& x, const Tensor& w, int64_t stride, int64_t pad, const Tensor& dout, const Tensor& mask,
                    Tensor& stats_buf) {
    dpa::ConvArgs a = fused_args(x, w, stride, pad, "conv_bnb_stats");
    check_like_output(dout, a, "dout");
    check_stats_buf(stats_buf, a.K);
    Guard g(x);
    a.y = nullptr; a.epi = 2; a.ep_src = dout.data_ptr(); a.ep_mask_in = check_mask(mask, a);
    a.ep_stats = stats_buf.data_ptr<float>(); a.ep_stats_rows = stats_buf.size(0);
    DetStats det;
    det.redirect(a.ep_stats, a.ep_stats_rows, stats_buf, det_slots((int64_t)a.N * a.Ho * a.Wo, 1));
    dpa::conv_igemm(a, cur_stream());
    det.finish();
}

// BN backward, apply pass: d(conv out) = cA * g + cB * c + cC with c recomputed
Tensor conv_bnb_apply(const Tensor& x, const Tensor& w, int64_t stride, int64_t pad, const Tensor& dout, const Tensor& mask,
                      const Tensor& cA, const Tensor& cB, const Tensor& cC) {
    dpa::ConvArgs a = fused_args(x, w, stride, pad, "conv_bnb_apply");
    check_like_output(dout, a, "dout");
    CHECK_F32(cA); CHECK_F32(cB); CHECK_F32(cC);
    TORCH_CHECK(cA.numel() == a.K && cB.numel() == a.K && cC.numel() == a.K, "conv_bnb_apply: coefficient sizes");
    Guard g(x);
    Tensor y = torch::empty({a.N, a.Ho, a.Wo, a.K}, x.options());
    a.y = y.data_ptr(); a.epi = 3; a.ep_a = cA.data_ptr<float>(); a.ep_b = cB.data_ptr<float>(); a.ep_c = cC.data_ptr<float>();
    a.ep_src = dout.data_ptr(); a.ep_mask_in = check_mask(mask, a);
    dpa::conv_igemm(a, cur_stream());
    return y;
}

// (sum g, sum g*x) accumulated by conv_bnb_stats -> dgamma, dbeta (accumulated into the targets when given) and
// the apply-pass coefficients; re-zeroes stats_buf.  Returns [5, C]: dgamma, dbeta, cA, cB, cC.
Tensor bn_bwd_coefs(Tensor& stats_buf, int64_t P, const Tensor& gamma, const Tensor& mean, const Tensor& invstd,
                    const c10::optional<Tensor>& dgamma_acc, const c10::optional<Tensor>& dbeta_acc) {
    CHECK_F32(gamma); CHECK_F32(mean); CHECK_F32(invstd);
    const int64_t C = gamma.numel();
    check_stats_buf(stats_buf, C);
    TORCH_CHECK(mean.numel() == C && invstd.numel() == C, "bn_bwd_coefs: sizes");
    Guard g(gamma);
    Tensor co = torch::empty({5, C}, gamma.options());
    float* o = co.data_ptr<float>();
    float* dg = o; float* db = o + C; int accumulate = 0;
    if (dgamma_acc.has_value() && dgamma_acc->defined()) {
        CHECK_F32((*dgamma_acc)); CHECK_F32((*dbeta_acc));
        TORCH_CHECK(dgamma_acc->numel() == C && dbeta_acc->numel() == C, "bn_bwd_coefs: accumulate targets");
        dg = dgamma_acc->data_ptr<float>(); db = dbeta_acc->data_ptr<float>(); accumulate = 1;
    }
    dpa::bn_bwd_finalize(stats_buf.data_ptr<float>(), stats_buf.size(0), P, C, gamma.data_ptr<float>(), mean.data_ptr<float>(),
                         invstd.data_ptr<float>(), dg, db, accumulate, o + 2 * C, o + 3 * C, o + 4 * C, 1, 1, cur_stream());
    return co;
}

// dgrad: dy [N,Ho,Wo,K] bf16, wt [C,R,S,K] bf16 (filter with K and C swapped) -> dx [N,H,W,C]
// Optional bn_x / bn_mask / bn_stats: the kernel also adds sum(g'), sum(g' * bn_x) per channel
// into bn_stats ([rows,2,C] fp32, zeroed), g' = dx (addend included) under the ReLU bit mask bn_mask - the reduction pass of the
// backward of the BatchNorm whose output this dx is the gradient of (finish with bn_bwd_coefs + bn_bwd_apply).
// Optional gelu_x (bf16, dx's shape; stride 1, nothing else fused): dx = dgrad * gelu'(gelu_x) - the backward of a
// tanh-GELU in front of this layer, in the epilogue.
Tensor conv_dgrad(const Tensor& dy, const Tensor& wt, int64_t H, int64_t W, int64_t stride, int64_t pad,
                  const c10::optional<Tensor>& addend, const c10::optional<Tensor>& addend_mask,
                  const c10::optional<Tensor>& bn_x, const c10::optional<Tensor>& bn_mask, const c10::optional<Tensor>& bn_stats,
                  const c10::optional<Tensor>& gelu_x) {
    CHECK_BF16(dy); CHECK_BF16(wt);
    TORCH_CHECK(dy.dim() == 4 && wt.dim() == 4, "conv_dgrad: 4-D tensors expected");
    const int64_t N = dy.size(0), Ho = dy.size(1), Wo = dy.size(2), K = dy.size(3);
    const int64_t C = wt.size(0), R = wt.size(1), S = wt.size(2);
    TORCH_CHECK(wt.size(3) == K, "conv_dgrad: channel mismatch");
    TORCH_CHECK(C % 8 == 0 && K % 8 == 0, "conv_dgrad: channels must be multiples of 8");
    TORCH_CHECK((H + 2 * pad - R) / stride + 1 == Ho && (W + 2 * pad - S) / stride + 1 == Wo, "conv_dgrad: geometry mismatch");
    Guard g(dy);
    Tensor dx = torch::empty({N, H, W, C}, dy.options());
    dpa::ConvArgs a{};
    DetStats det;
    a.x = dy.data_ptr(); a.w = wt.data_ptr(); a.y = dx.data_ptr(); a.bias = nullptr; a.stats = nullptr;
    a.stats_rows = 0; a.addend = nullptr;
    TORCH_CHECK(dy.numel() < (1ll << 30) && wt.numel() < (1ll << 30) && N * H * W * C < (1ll << 31) - 16,
                "conv_dgrad: operands must have fewer than 2^30 elements");
    if (addend.has_value() && addend->defined()) {
        CHECK_BF16((*addend));
        TORCH_CHECK(addend->numel() == N * H * W * C, "conv_dgrad: addend shape");
        a.addend = addend->data_ptr();
        if (addend_mask.has_value() && addend_mask->defined()) {
            TORCH_CHECK(addend_mask->scalar_type() == at::kByte && addend_mask->is_contiguous() && addend_mask->numel() * 8 == addend->numel(),
                        "conv_dgrad: addend_mask must be uint8 with one bit per addend element");
            a.addend_mask = addend_mask->data_ptr<uint8_t>();
        }
    }
    a.N = N; a.H = Ho; a.W = Wo; a.C = K; a.K = C; a.R = R; a.S = S; a.Ho = H; a.Wo = W;
    a.stride = stride; a.pad = pad; a.mode = 1; a.act = 0;
    if (bn_x.has_value() && bn_x->defined()) {
        TORCH_CHECK(bn_mask.has_value() && bn_mask->defined() && bn_stats.has_value() && bn_stats->defined(), "conv_dgrad: bn_mask / bn_stats missing");
        CHECK_BF16((*bn_x)); CHECK_F32((*bn_stats));
        TORCH_CHECK(bn_x->numel() == dx.numel() && bn_mask->scalar_type() == at::kByte && bn_mask->is_contiguous() &&
                    bn_mask->numel() * 8 == dx.numel(), "conv_dgrad: bn_x / bn_mask shapes");
        TORCH_CHECK(bn_stats->dim() == 3 && bn_stats->size(1) == 2 && bn_stats->size(2) == C, "conv_dgrad: bn_stats must be [rows,2,C]");
        a.epi = 4; a.ep_src = bn_x->data_ptr(); a.ep_mask_in = bn_mask->data_ptr<uint8_t>();
        a.ep_stats = bn_stats->data_ptr<float>(); a.ep_stats_rows = (int)bn_stats->size(0);
        det.redirect(a.ep_stats, a.ep_stats_rows, *bn_stats, det_slots(N * H * W, stride));
    }
    if (gelu_x.has_value() && gelu_x->defined()) {
        CHECK_BF16((*gelu_x));
        TORCH_CHECK(stride == 1 && a.addend == nullptr && a.epi == 0 && gelu_x->numel() == dx.numel(),
                    "conv_dgrad: gelu_x needs stride 1, dx's shape and no other fused epilogue");
        a.epi = 5; a.ep_src = gelu_x->data_ptr();
    }
    dpa::conv_igemm(a, cur_stream());
    det.finish();
    return dx;
}

// the apply pass of a BatchNorm backward alone: dx = A * dy' + B * x + C per channel (co = bn_bwd_coefs(...): rows 2..4),
// dy' = dy under the ReLU bit mask
Tensor bn_bwd_apply(const Tensor& dy, const c10::optional<Tensor>& mask, const Tensor& x, const Tensor& co, bool relu) {
    CHECK_BF16(dy); CHECK_BF16(x); CHECK_F32(co);
    const int64_t C = x.size(-1), P = x.numel() / C;
    TORCH_CHECK(dy.numel() == x.numel() && co.dim() == 2 && co.size(0) == 5 && co.size(1) == C && C % 8 == 0, "bn_bwd_apply: shapes");
    const uint8_t* mk = nullptr;
    if (relu) {
        TORCH_CHECK(mask.has_value() && mask->defined() && mask->scalar_type() == at::kByte && mask->is_contiguous() &&
                    mask->numel() * 8 == x.numel(), "bn_bwd_apply: relu needs the uint8 bit mask from bn_apply");
        mk = mask->data_ptr<uint8_t>();
    }
    Guard g(x);
    Tensor dx = torch::empty_like(x);
    const float* o = co.data_ptr<float>();
    dpa::bn_bwd_apply(dy.data_ptr(), mk, x.data_ptr(), o + 2 * C, o + 3 * C, o + 4 * C, dx.data_ptr(), P, C, relu, cur_stream());
    return dx;
}

// Backward of "1x1 conv -> BatchNorm -> ReLU mask" in one kernel (csrc/tail_bwd.hip): the BatchNorm-backward apply pass
// (co = bn_bwd_coefs(...)), the conv's filter gradient (accumulated into dw) and its input gradient (returned), plus -
// with bn_x / bn_mask / bn_stats - the reduction pass of the BatchNorm behind the returned gradient, as conv_dgrad does.
bool tail_bwd_ok(int64_t P, int64_t CM, int64_t CO) {
    dpa::TailBwdArgs a{};
    a.P = P; a.CM = (int)CM; a.CO = (int)CO;
    return dpa::tail_bwd_supported(a);
}
Tensor tail_bwd(const Tensor& dout, const c10::optional<Tensor>& kout, const Tensor& craw, const Tensor& co, const Tensor& xin,
                const Tensor& wt, Tensor& dw, const c10::optional<Tensor>& bn_x, const c10::optional<Tensor>& bn_mask,
                const c10::optional<Tensor>& bn_stats) {
    CHECK_BF16(dout); CHECK_BF16(craw); CHECK_BF16(xin); CHECK_BF16(wt); CHECK_F32(co); CHECK_F32(dw);
    const int64_t CO = craw.size(-1), CM = xin.size(-1), P = craw.numel() / CO;
    TORCH_CHECK(dout.numel() == craw.numel() && xin.numel() == P * CM, "tail_bwd: dout / craw / xin shapes");
    TORCH_CHECK(wt.numel() == CM * CO && wt.size(0) == CM && wt.size(-1) == CO, "tail_bwd: wt must be [CM,1,1,CO]");
    TORCH_CHECK(dw.numel() == CM * CO && dw.size(0) == CO, "tail_bwd: dw must be [CO,1,1,CM]");
    TORCH_CHECK(co.dim() == 2 && co.size(0) == 5 && co.size(1) == CO, "tail_bwd: co must be bn_bwd_coefs' [5,CO]");
    dpa::TailBwdArgs a{};
    a.dout = dout.data_ptr(); a.craw = craw.data_ptr(); a.xin = xin.data_ptr(); a.wt = wt.data_ptr();
    a.dw = dw.data_ptr<float>();
    const float* o = co.data_ptr<float>();
    a.cA = o + 2 * CO; a.cB = o + 3 * CO; a.cC = o + 4 * CO;
    a.P = P; a.CM = (int)CM; a.CO = (int)CO;
    if (kout.has_value() && kout->defined()) {
        TORCH_CHECK(kout->is_cuda() && kout->scalar_type() == at::kByte && kout->is_contiguous() && kout->numel() * 8 == dout.numel(),
                    "tail_bwd: kout must be uint8 with one bit per element of dout");
        a.kout = kout->data_ptr<uint8_t>();
    }
    TORCH_CHECK(dpa::tail_bwd_supported(a), "tail_bwd: unsupported shape (ask tail_bwd_ok first)");
    Guard g(dout);
    std::vector<int64_t> shp(xin.sizes().begin(), xin.sizes().end());
    Tensor dx = torch::empty(shp, xin.options());
    a.dx = dx.data_ptr();
    if (bn_x.has_value() && bn_x->defined()) {
        TORCH_CHECK(bn_mask.has_value() && bn_mask->defined() && bn_stats.has_value() && bn_stats->defined(), "tail_bwd: bn_mask / bn_stats missing");
        CHECK_BF16((*bn_x)); CHECK_F32((*bn_stats));
        TORCH_CHECK(bn_x->numel() == dx.numel() && bn_mask->is_cuda() && bn_mask->scalar_type() == at::kByte && bn_mask->is_contiguous() &&
                    bn_mask->numel() * 8 == dx.numel(), "tail_bwd: bn_x / bn_mask shapes");
        TORCH_CHECK(bn_stats->dim() == 3 && bn_stats->size(1) == 2 && bn_stats->size(2) == CM && bn_stats->size(0) >= 1, "tail_bwd: bn_stats must be [rows,2,CM]");
        a.sx = bn_x->data_ptr(); a.smask = bn_mask->data_ptr<uint8_t>();
        a.stats = bn_stats->data_ptr<float>(); a.stats_rows = (int)bn_stats->size(0);
    }
    dpa::tail_bwd(a, cur_stream());
    return dx;
}

// dw [K,R,S,C] fp32 += wgrad(x, dy).  dw must be initialised by the caller.  With dbias (fp32 [K]) the bias
// gradient dbias += column sums of dy is produced too: inside the wgrad kernel where that variant exists, else by
// the column-sum kernels.
void conv_wgrad(const Tensor& x, const Tensor& dy, Tensor& dw, int64_t stride, int64_t pad, const c10::optional<Tensor>& dbias,
                int64_t stride_w, int64_t pad_w) {
    CHECK_BF16(x); CHECK_BF16(dy); CHECK_F32(dw);
    TORCH_CHECK(x.dim() == 4 && dy.dim() == 4 && dw.dim() == 4, "conv_wgrad: 4-D tensors expected");
    const int64_t N = x.size(0), H = x.size(1), W = x.size(2), C = x.size(3);
    const int64_t Ho = dy.size(1), Wo = dy.size(2), K = dy.size(3);
    const int64_t R = dw.size(1), S = dw.size(2);
    TORCH_CHECK(dy.size(0) == N && dw.size(0) == K && dw.size(3) == C, "conv_wgrad: shape mismatch");
    const bool asym = stride_w >= 0 || pad_w >= 0;
    TORCH_CHECK(!asym || (stride_w >= 1 && pad_w >= 0), "conv_wgrad: stride_w and pad_w go together");
    TORCH_CHECK((H + 2 * pad - R) / stride + 1 == Ho &&
                (asym ? Wo >= 1 && Wo <= (W + 2 * pad_w - S) / stride_w + 1 : (W + 2 * pad - S) / stride + 1 == Wo),
                "conv_wgrad: geometry mismatch");
    TORCH_CHECK(C % 8 == 0 && K % 8 == 0, "conv_wgrad: channels must be multiples of 8");
    TORCH_CHECK(x.numel() < (1ll << 30) - (1ll << 24) && dy.numel() < (1ll << 30) - (1ll << 24),
                "conv_wgrad: operands must stay below 2^31 bytes (32-bit buffer offsets)");
    Guard g(x);
    dpa::WgradArgs a{};
    a.x = x.data_ptr(); a.dy = dy.data_ptr(); a.dw = dw.data_ptr<float>();
    a.N = N; a.H = H; a.W = W; a.C = C; a.K = K; a.R = R; a.S = S; a.Ho = Ho; a.Wo = Wo; a.stride = stride; a.pad = pad;
    a.asym = asym ? 1 : 0; a.stride_w = asym ? (int)stride_w : 0; a.pad_w = asym ? (int)pad_w : 0;
    a.dbias = nullptr;
    const bool want_bias = dbias.has_value() && dbias->defined();
    if (want_bias) {
        CHECK_F32((*dbias));
        TORCH_CHECK(dbias->numel() == K, "conv_wgrad: dbias must have K elements");
        if (dpa::conv_wgrad_fuses_bias(a)) a.dbias = dbias->data_ptr<float>();
    }
    dpa::conv_wgrad(a, cur_stream());
    if (want_bias && a.dbias == nullptr) {
        const int64_t M = dy.numel() / K;
        const int wr = dpa::colsum_workspace_rows(M, K);
        Tensor ws;
        if (wr > 0) ws = torch::empty({wr, K}, dbias->options());
        dpa::colsum_bf16(dy.data_ptr(), dbias->data_ptr<float>(), M, K, wr > 0 ? ws.data_ptr<float>() : nullptr, cur_stream());
    }
}

// Weight gradients of several Linear layers over the same tokens in one launch (csrc/wgrad_wide.hip):
// dw[q] [K,1,1,C] fp32 += dy[q]^T x[q]; dbias[q] (fp32 [K], optional per layer) += column sums of dy[q].
void wgrad_wide_group(const std::vector<Tensor>& xs, const std::vector<Tensor>& dys, std::vector<Tensor> dws,
                      const std::vector<c10::optional<Tensor>>& dbiases) {
    const size_t n = xs.size();
    TORCH_CHECK(n >= 1 && n <= (size_t)dpa::WGRAD_WIDE_MAX && dys.size() == n && dws.size() == n && dbiases.size() == n,
                "wgrad_wide_group: 1..", dpa::WGRAD_WIDE_MAX, " layers, one entry per list each");
    std::vector<dpa::WgradArgs> probs(n);
    int64_t M = -1;
    for (size_t q = 0; q < n; ++q) {
        const Tensor &x = xs[q], &dy = dys[q];
        Tensor& dw = dws[q];
        CHECK_BF16(x); CHECK_BF16(dy); CHECK_F32(dw);
        const int64_t C = x.size(-1), K = dy.size(-1), rows = x.numel() / C;
        TORCH_CHECK(dy.numel() / K == rows && dw.numel() == K * C, "wgrad_wide_group: layer ", q, " shape mismatch");
        TORCH_CHECK(M < 0 || rows == M, "wgrad_wide_group: every layer must reduce over the same rows");
        M = rows;
        dpa::WgradArgs a{};
        a.x = x.data_ptr(); a.dy = dy.data_ptr(); a.dw = dw.data_ptr<float>();
        a.N = (int)rows; a.H = a.W = a.Ho = a.Wo = 1; a.C = (int)C; a.K = (int)K; a.R = a.S = 1; a.stride = 1; a.pad = 0;
        a.dbias = nullptr;
        if (dbiases[q].has_value() && dbiases[q]->defined()) {
            CHECK_F32((*dbiases[q]));
            TORCH_CHECK(dbiases[q]->numel() == K, "wgrad_wide_group: dbias must have K elements");
            a.dbias = dbiases[q]->data_ptr<float>();
        }
        TORCH_CHECK(dpa::wgrad_wide_supported(a), "wgrad_wide_group: layer ", q, " (", rows, " x ", K, " x ", C, ") is not supported");
        probs[q] = a;
    }
    Guard g(xs[0]);
    dpa::wgrad_wide_group(probs.data(), (int)n, cur_stream());
}

// whether conv_wgrad's dispatcher would send this Linear-shaped problem to the wide kernel, and its 256 x 256 tile count
std::tuple<bool, int64_t> wgrad_wide_query(int64_t rows, int64_t K, int64_t C) {
    dpa::WgradArgs a{};
    a.N = (int)rows; a.H = a.W = a.Ho = a.Wo = 1; a.C = (int)C; a.K = (int)K; a.R = a.S = 1; a.stride = 1; a.pad = 0;
    return {dpa::wgrad_wide_eligible(a), (int64_t)dpa::wgrad_wide_tiles(a)};
}

Tensor transpose_filter(const Tensor& w) {
    CHECK_BF16(w);
    TORCH_CHECK(w.dim() == 4, "transpose_filter: KRSC expected");
    Guard g(w);
    Tensor wt = torch::empty({w.size(3), w.size(1), w.size(2), w.size(0)}, w.options());
    dpa::transpose_filter(w.data_ptr(), wt.data_ptr(), w.size(0), w.size(1) * w.size(2), w.size(3), cur_stream());
    return wt;
}

void transpose_filters_batched(const Tensor& src, Tensor& dst, const Tensor& table, int64_t total_tiles) {
    CHECK_BF16(src); CHECK_BF16(dst); CHECK_I64(table);
    TORCH_CHECK(table.dim() == 2 && table.size(1) == 6 && dst.numel() == src.numel() && total_tiles > 0, "transpose_filters_batched: shapes");
    Guard g(src);
    dpa::transpose_filters_batched(src.data_ptr(), dst.data_ptr(), (const long*)table.data_ptr<int64_t>(), (int)table.size(0),
                                   total_tiles, cur_stream());
}

void colsum(const Tensor& x, Tensor& out) {
    CHECK_BF16(x); CHECK_F32(out);
    const int64_t K = x.size(-1);
    TORCH_CHECK(K % 8 == 0 && out.numel() == K, "colsum: bad shapes");
    Guard g(x);
    const int64_t M = x.numel() / K;
    const int wr = dpa::colsum_workspace_rows(M, K);
    Tensor ws;
    if (wr > 0) ws = torch::empty({wr, K}, out.options());
    dpa::colsum_bf16(x.data_ptr(), out.data_ptr<float>(), M, K, wr > 0 ? ws.data_ptr<float>() : nullptr, cur_stream());
}

// ---------------------------------------------------------------- batch norm
Tensor bn_stats(const Tensor& x) {
    CHECK_BF16(x);
    const int64_t C = x.size(-1), P = x.numel() / C;
    TORCH_CHECK(C % 8 == 0 && C <= 2048, "bn_stats: C must be a multiple of 8 and <= 2048");
    Guard g(x);
    Tensor stats = torch::empty({dpa::bn_stats_rows(P), 2, C}, x.options().dtype(at::kFloat));
    dpa::bn_stats(x.data_ptr(), stats.data_ptr<float>(), P, C, cur_stream());
    return stats;
}

// stats [rows,2,C] -> (mean, invstd, scale, shift); updates running stats in place
std::vector<Tensor> bn_finalize(const Tensor& stats_in, int64_t P, const Tensor& gamma, const Tensor& beta,
                                const c10::optional<Tensor>& running_mean, const c10::optional<Tensor>& running_var,
                                double momentum, double eps, bool reset_stats) {
    CHECK_F32(stats_in); CHECK_F32(gamma); CHECK_F32(beta);
    TORCH_CHECK(stats_in.dim() == 3 && stats_in.size(1) == 2, "bn_finalize: stats must be [rows,2,C]");
    const int64_t C = stats_in.size(2);
    TORCH_CHECK(gamma.numel() == C && beta.numel() == C, "bn_finalize: affine size");
    Guard g(stats_in);
    Tensor stats = stats_in;
    int64_t rows = stats.size(0);
    if (rows > 256) {  // fold tall partial buffers in parallel first
        const int chunk = 64;
        Tensor folded = torch::empty({(rows + chunk - 1) / chunk, 2, C}, stats.options());
        dpa::fold_rows(stats.data_ptr<float>(), rows, 2 * C, chunk, folded.data_ptr<float>(), cur_stream());
        if (reset_stats) stats_in.zero_();
        stats = folded; rows = folded.size(0); reset_stats = false;
    }
    Tensor out = torch::empty({4, C}, stats.options());
    float* rm = nullptr; float* rv = nullptr;
    if (running_mean.has_value() && running_mean->defined()) {
        CHECK_F32((*running_mean)); CHECK_F32((*running_var));
        TORCH_CHECK(running_mean->numel() == C && running_var->numel() == C, "running stats size");
        rm = running_mean->data_ptr<float>(); rv = running_var->data_ptr<float>();
    }
    float* o = out.data_ptr<float>();
    dpa::bn_finalize(stats.data_ptr<float>(), rows, P, C, gamma.data_ptr<float>(), beta.data_ptr<float>(), rm, rv,
                     (float)momentum, (float)eps, o, o + C, o + 2 * C, o + 3 * C, reset_stats ? 1 : 0, cur_stream());
    return {out[0], out[1], out[2], out[3]};
}

// returns (y, mask): mask is the uint8 bit mask (y > 0) when relu && want_mask, else undefined
// res_scale / res_shift: the residual is a RAW conv output with its own BatchNorm (projection shortcut) - its affine is
// applied on the fly, so that BatchNorm needs no apply pass of its own.
std::vector<Tensor> bn_apply(const Tensor& x, const c10::optional<Tensor>& residual, const Tensor& scale, const Tensor& shift,
                             bool relu, bool want_mask, const c10::optional<Tensor>& res_scale, const c10::optional<Tensor>& res_shift) {
    CHECK_BF16(x); CHECK_F32(scale); CHECK_F32(shift);
    const int64_t C = x.size(-1), P = x.numel() / C;
    TORCH_CHECK(C % 8 == 0 && scale.numel() == C && shift.numel() == C, "bn_apply: bad channel count");
    TORCH_CHECK(((uintptr_t)scale.data_ptr()) % 16 == 0 && ((uintptr_t)shift.data_ptr()) % 16 == 0, "scale/shift alignment");
    const void* res = nullptr;
    if (residual.has_value() && residual->defined()) {
        CHECK_BF16((*residual)); TORCH_CHECK(residual->numel() == x.numel(), "residual shape");
        res = residual->data_ptr();
    }
    Guard g(x);
    Tensor y = torch::empty_like(x);
    Tensor mask;
    if (relu && want_mask) mask = torch::empty({x.numel() / 8}, x.options().dtype(at::kByte));
    const float* rs = nullptr; const float* rb = nullptr;
    if (res_scale.has_value() && res_scale->defined()) {
        TORCH_CHECK(res != nullptr && res_shift.has_value() && res_shift->defined(), "bn_apply: res_scale needs residual and res_shift");
        CHECK_F32((*res_scale)); CHECK_F32((*res_shift));
        TORCH_CHECK(res_scale->numel() == C && res_shift->numel() == C, "bn_apply: res_scale / res_shift size");
        rs = res_scale->data_ptr<float>(); rb = res_shift->data_ptr<float>();
    }
    dpa::bn_apply(x.data_ptr(), res, y.data_ptr(), scale.data_ptr<float>(), shift.data_ptr<float>(),
                  mask.defined() ? mask.data_ptr<uint8_t>() : nullptr, P, C, relu, rs, rb, cur_stream());
    return {y, mask};
}

// returns (dx, dgamma, dbeta).  relu: `mask` is the bit mask written by bn_apply (the gradient
// of a residual branch is dy under the same mask: relu_mask_mul, or the dgrad addend_mask).
// the reduction + finalize half of bn_backward: returns co [5, C] = dgamma, dbeta, cA, cB, cC (dx = cA * dy' + cB * x + cC)
Tensor bn_backward_coefs(const Tensor& dy, const c10::optional<Tensor>& mask, const Tensor& x, const Tensor& gamma,
                         const Tensor& mean, const Tensor& invstd, bool relu, const c10::optional<Tensor>& dgamma_acc,
                         const c10::optional<Tensor>& dbeta_acc, const c10::optional<Tensor>& stats_buf) {
    CHECK_BF16(dy); CHECK_BF16(x); CHECK_F32(gamma); CHECK_F32(mean); CHECK_F32(invstd);
    const int64_t C = x.size(-1), P = x.numel() / C;
    TORCH_CHECK(dy.numel() == x.numel(), "bn_backward: shape mismatch");
    TORCH_CHECK(C % 8 == 0 && C <= 2048 && gamma.numel() == C && mean.numel() == C && invstd.numel() == C, "bn_backward: channels");
    const uint8_t* mk = nullptr;
    if (relu) {
        TORCH_CHECK(mask.has_value() && mask->defined() && mask->scalar_type() == at::kByte && mask->is_contiguous() &&
                    mask->numel() * 8 == x.numel(), "bn_backward: relu needs the uint8 bit mask from bn_apply");
        mk = mask->data_ptr<uint8_t>();
    }
    Guard g(x);
    auto fopt = x.options().dtype(at::kFloat);
    int rows; int replicas = 0; int reset = 0;
    Tensor stats;
    if (stats_buf.has_value() && stats_buf->defined() && !dpa::deterministic()) {     // persistent zeroed accumulator (re-zeroed by the finalize kernel)
        stats = *stats_buf;
        CHECK_F32(stats);
        TORCH_CHECK(stats.dim() == 3 && stats.size(1) == 2 && stats.size(2) == C, "bn_backward: stats_buf must be [rows,2,C]");
        rows = stats.size(0); replicas = rows; reset = 1;
    } else {
        rows = dpa::bn_stats_rows(P);
        stats = torch::empty({rows, 2, C}, fopt);
    }
    dpa::bn_bwd_stats(dy.data_ptr(), mk, x.data_ptr(), mean.data_ptr<float>(), invstd.data_ptr<float>(),
                      stats.data_ptr<float>(), P, C, relu, replicas, cur_stream());
    if (replicas == 0 && rows > 256) {
        const int chunk = 64;
        Tensor folded = torch::empty({(rows + chunk - 1) / chunk, 2, C}, fopt);
        dpa::fold_rows(stats.data_ptr<float>(), rows, 2 * C, chunk, folded.data_ptr<float>(), cur_stream());
        stats = folded; rows = folded.size(0);
    }
    Tensor co = torch::empty({5, C}, fopt);  // dgamma, dbeta, A, B, C0
    float* o = co.data_ptr<float>();
    float* dg = o; float* db = o + C; int accumulate = 0;
    if (dgamma_acc.has_value() && dgamma_acc->defined()) {     // add straight into the flat gradient buffer
        CHECK_F32((*dgamma_acc)); CHECK_F32((*dbeta_acc));
        TORCH_CHECK(dgamma_acc->numel() == C && dbeta_acc->numel() == C, "bn_backward: accumulate targets");
        dg = dgamma_acc->data_ptr<float>(); db = dbeta_acc->data_ptr<float>(); accumulate = 1;
    }
    dpa::bn_bwd_finalize(stats.data_ptr<float>(), rows, P, C, gamma.data_ptr<float>(), mean.data_ptr<float>(),
                         invstd.data_ptr<float>(), dg, db, accumulate, o + 2 * C, o + 3 * C, o + 4 * C, reset, 0, cur_stream());
    return co;
}

std::vector<Tensor> bn_backward(const Tensor& dy, const c10::optional<Tensor>& mask, const Tensor& x, const Tensor& gamma,
                                const Tensor& mean, const Tensor& invstd, bool relu, const c10::optional<Tensor>& dgamma_acc,
                                const c10::optional<Tensor>& dbeta_acc, const c10::optional<Tensor>& stats_buf) {
    Tensor co = bn_backward_coefs(dy, mask, x, gamma, mean, invstd, relu, dgamma_acc, dbeta_acc, stats_buf);
    const int64_t C = x.size(-1), P = x.numel() / C;
    Guard g(x);
    const float* o = co.data_ptr<float>();
    Tensor dx = torch::empty_like(x);
    dpa::bn_bwd_apply(dy.data_ptr(), relu ? mask->data_ptr<uint8_t>() : nullptr, x.data_ptr(), o + 2 * C, o + 3 * C, o + 4 * C,
                      dx.data_ptr(), P, C, relu, cur_stream());
    return {dx, co[0], co[1]};
}

Tensor relu_mask_mul(const Tensor& dy, const Tensor& mask) {
    CHECK_BF16(dy);
    TORCH_CHECK(mask.scalar_type() == at::kByte && mask.is_contiguous() && mask.numel() * 8 == dy.numel(), "relu_mask_mul: mask size");
    Guard g(dy);
    Tensor out = torch::empty_like(dy);
    dpa::relu_mask_mul(dy.data_ptr(), mask.data_ptr<uint8_t>(), out.data_ptr(), dy.numel(), cur_stream());
    return out;
}

// stem: (pooled y, idx) = maxpool3x3/2(relu(x * scale + shift)), x = raw conv output [N,H,W,C]
std::vector<Tensor> bn_relu_maxpool_fwd(const Tensor& x, const Tensor& scale, const Tensor& shift) {
    CHECK_BF16(x); CHECK_F32(scale); CHECK_F32(shift);
    TORCH_CHECK(x.dim() == 4 && x.size(3) % 8 == 0 && scale.numel() == x.size(3) && shift.numel() == x.size(3), "bn_relu_maxpool_fwd: shapes");
    TORCH_CHECK(x.numel() < (1ll << 31), "bn_relu_maxpool_fwd: tensor too large for 32-bit indexing");
    const int64_t N = x.size(0), H = x.size(1), W = x.size(2), C = x.size(3);
    const int64_t Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    Guard g(x);
    Tensor y = torch::empty({N, Ho, Wo, C}, x.options());
    Tensor idx = torch::empty({N, Ho, Wo, C}, x.options().dtype(at::kByte));
    dpa::bn_relu_maxpool_fwd(x.data_ptr(), scale.data_ptr<float>(), shift.data_ptr<float>(), y.data_ptr(), idx.data_ptr<uint8_t>(),
                             N, H, W, C, Ho, Wo, cur_stream());
    return {y, idx};
}

// backward of the above down to the conv output: returns (dx, dgamma, dbeta); accumulators / stats_buf as bn_backward
std::vector<Tensor> pool_bn_backward(const Tensor& dp, const Tensor& idx, const Tensor& x, const Tensor& gamma, const Tensor& mean,
                                     const Tensor& invstd, const c10::optional<Tensor>& dgamma_acc,
                                     const c10::optional<Tensor>& dbeta_acc, const Tensor& stats_buf) {
    CHECK_BF16(dp); CHECK_BF16(x); CHECK_F32(gamma); CHECK_F32(mean); CHECK_F32(invstd); CHECK_F32(stats_buf);
    CHECK_DEV(idx); CHECK_CONTIG(idx);
    TORCH_CHECK(x.dim() == 4 && dp.dim() == 4, "pool_bn_backward: NHWC tensors expected");
    const int64_t N = x.size(0), H = x.size(1), W = x.size(2), C = x.size(3), Ho = dp.size(1), Wo = dp.size(2);
    TORCH_CHECK(dp.size(0) == N && dp.size(3) == C && (H - 1) / 2 + 1 == Ho && (W - 1) / 2 + 1 == Wo && idx.numel() == dp.numel() &&
                idx.scalar_type() == at::kByte, "pool_bn_backward: geometry");
    TORCH_CHECK(C % 8 == 0 && C <= 2048 && gamma.numel() == C && mean.numel() == C && invstd.numel() == C && x.numel() < (1ll << 31),
                "pool_bn_backward: channels");
    TORCH_CHECK(stats_buf.dim() == 3 && stats_buf.size(1) == 2 && stats_buf.size(2) == C, "pool_bn_backward: stats_buf must be [rows,2,C]");
    Guard g(x);
    int rows = stats_buf.size(0);
    const int64_t P = N * H * W;
    Tensor stats = stats_buf;
    if (dpa::deterministic()) {        // one partial row per workgroup (plain stores), folded in order by the finalize kernel
        rows = dpa::bn_stats_rows(P);
        stats = torch::empty({rows, 2, C}, stats_buf.options());
    }
    dpa::pool_bn_bwd_stats(dp.data_ptr(), idx.data_ptr<uint8_t>(), x.data_ptr(), mean.data_ptr<float>(), invstd.data_ptr<float>(),
                           stats.data_ptr<float>(), N, H, W, C, Ho, Wo, dpa::deterministic() ? 0 : rows, cur_stream());
    Tensor co = torch::empty({5, C}, x.options().dtype(at::kFloat));
    float* o = co.data_ptr<float>();
    float* dg = o; float* db = o + C; int accumulate = 0;
    if (dgamma_acc.has_value() && dgamma_acc->defined()) {
        CHECK_F32((*dgamma_acc)); CHECK_F32((*dbeta_acc));
        TORCH_CHECK(dgamma_acc->numel() == C && dbeta_acc->numel() == C, "pool_bn_backward: accumulate targets");
        dg = dgamma_acc->data_ptr<float>(); db = dbeta_acc->data_ptr<float>(); accumulate = 1;
    }
    dpa::bn_bwd_finalize(stats.data_ptr<float>(), rows, P, C, gamma.data_ptr<float>(), mean.data_ptr<float>(),
                         invstd.data_ptr<float>(), dg, db, accumulate, o + 2 * C, o + 3 * C, o + 4 * C, 1, 0, cur_stream());
    Tensor dx = torch::empty_like(x);
    dpa::pool_bn_bwd_apply(dp.data_ptr(), idx.data_ptr<uint8_t>(), x.data_ptr(), o + 2 * C, o + 3 * C, o + 4 * C, dx.data_ptr(),
                           N, H, W, C, Ho, Wo, cur_stream());
    return {dx, co[0], co[1]};
}

// ---------------------------------------------------------------- pooling / layout / elementwise
std::vector<Tensor> maxpool_fwd(const Tensor& x) {
    CHECK_BF16(x); TORCH_CHECK(x.dim() == 4 && x.size(3) % 8 == 0, "maxpool: NHWC with C % 8 == 0");
    const int64_t N = x.size(0), H = x.size(1), W = x.size(2), C = x.size(3);
    const int64_t Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    Guard g(x);
    Tensor y = torch::empty({N, Ho, Wo, C}, x.options());
    Tensor idx = torch::empty({N, Ho, Wo, C}, x.options().dtype(at::kByte));
    dpa::maxpool3x3s2_fwd(x.data_ptr(), y.data_ptr(), idx.data_ptr<uint8_t>(), N, H, W, C, Ho, Wo, cur_stream());
    return {y, idx};
}

Tensor maxpool_bwd(const Tensor& dy, const Tensor& idx, int64_t H, int64_t W) {
    CHECK_BF16(dy); CHECK_DEV(idx); CHECK_CONTIG(idx);
    TORCH_CHECK(idx.scalar_type() == at::kByte && idx.numel() == dy.numel(), "maxpool_bwd: idx");
    const int64_t N = dy.size(0), Ho = dy.size(1), Wo = dy.size(2), C = dy.size(3);
    TORCH_CHECK((H - 1) / 2 + 1 == Ho && (W - 1) / 2 + 1 == Wo, "maxpool_bwd: geometry");
    Guard g(dy);
    Tensor dx = torch::empty({N, H, W, C}, dy.options());
    dpa::maxpool3x3s2_bwd(dy.data_ptr(), idx.data_ptr<uint8_t>(), dx.data_ptr(), N, H, W, C, Ho, Wo, cur_stream());
    return dx;
}

Tensor avgpool_fwd(const Tensor& x) {
    CHECK_BF16(x); TORCH_CHECK(x.dim() == 4 && x.size(3) % 8 == 0, "avgpool: NHWC with C % 8 == 0");
    Guard g(x);
    Tensor y = torch::empty({x.size(0), x.size(3)}, x.options());
    dpa::avgpool_fwd(x.data_ptr(), y.data_ptr(), x.size(0), x.size(1) * x.size(2), x.size(3), cur_stream());
    return y;
}

Tensor avgpool_bwd(const Tensor& dy, int64_t H, int64_t W) {
    CHECK_BF16(dy); TORCH_CHECK(dy.dim() == 2 && dy.size(1) % 8 == 0, "avgpool_bwd: [N,C]");
    Guard g(dy);
    Tensor dx = torch::empty({dy.size(0), H, W, dy.size(1)}, dy.options());
    dpa::avgpool_bwd(dy.data_ptr(), dx.data_ptr(), dy.size(0), H * W, dy.size(1), cur_stream());
    return dx;
}

Tensor nchw_to_nhwc_pad(const Tensor& x, int64_t Cpad) {
    CHECK_DEV(x); CHECK_CONTIG(x);
    TORCH_CHECK(x.dim() == 4, "NCHW expected");
    TORCH_CHECK(x.scalar_type() == at::kFloat || x.scalar_type() == at::kBFloat16, "fp32 or bf16 input");
    TORCH_CHECK(Cpad % 8 == 0 && Cpad >= x.size(1), "Cpad");
    Guard g(x);
    Tensor y = torch::empty({x.size(0), x.size(2), x.size(3), Cpad}, x.options().dtype(at::kBFloat16));
    dpa::nchw_to_nhwc_pad(x.data_ptr(), x.scalar_type() == at::kBFloat16, y.data_ptr(), x.size(0), x.size(1), x.size(2),
                          x.size(3), Cpad, cur_stream());
    return y;
}

// pair-packed stem: see elementwise.hip
Tensor nchw_to_pairs(const Tensor& x) {
    CHECK_DEV(x); CHECK_CONTIG(x);
    TORCH_CHECK(x.dim() == 4 && x.size(1) >= 1 && x.size(1) <= 3 && x.size(3) % 2 == 0, "nchw_to_pairs: NCHW, C <= 3, even W");
    TORCH_CHECK(x.scalar_type() == at::kFloat || x.scalar_type() == at::kBFloat16, "fp32 or bf16 input");
    Guard g(x);
    Tensor y = torch::empty({x.size(0), x.size(2), x.size(3) / 2, 8}, x.options().dtype(at::kBFloat16));
    dpa::nchw_to_pairs(x.data_ptr(), x.scalar_type() == at::kBFloat16, y.data_ptr(), x.size(0), x.size(1), x.size(2), x.size(3), cur_stream());
    return y;
}
Tensor stem_pack_filter(const Tensor& w, int64_t C) {
    CHECK_BF16(w);
    TORCH_CHECK(w.dim() == 4 && w.size(2) == 7 && w.size(3) == 8 && C >= 1 && C <= 3, "stem_pack_filter: [K,R,7,8] filter, C <= 3");
    Guard g(w);
    Tensor wp = torch::empty({w.size(0), w.size(1), 4, 8}, w.options());
    dpa::stem_pack_filter(w.data_ptr(), wp.data_ptr(), w.size(0) * w.size(1), C, cur_stream());
    return wp;
}
void stem_unpack_dw(const Tensor& dwp, Tensor& dw, int64_t C) {
    CHECK_F32(dwp); CHECK_F32(dw);
    TORCH_CHECK(dwp.dim() == 4 && dw.dim() == 4 && dwp.size(2) == 4 && dwp.size(3) == 8 && dw.size(2) == 7 && dw.size(3) == 8 &&
                dwp.size(0) == dw.size(0) && dwp.size(1) == dw.size(1) && C >= 1 && C <= 3, "stem_unpack_dw: shapes");
    Guard g(dw);
    dpa::stem_unpack_dw(dwp.data_ptr<float>(), dw.data_ptr<float>(), dw.size(0) * dw.size(1), C, cur_stream());
}

Tensor ew2(const Tensor& a, const Tensor& b, int64_t op) {
    CHECK_BF16(a); CHECK_BF16(b); TORCH_CHECK(a.numel() == b.numel(), "elementwise: size mismatch");
    Guard g(a);
    Tensor y = torch::empty_like(a);
    const long n = a.numel();
    switch (op) {
        case 0: dpa::add_bf16(a.data_ptr(), b.data_ptr(), y.data_ptr(), n, cur_stream()); break;
        case 1: dpa::add_relu_fwd(a.data_ptr(), b.data_ptr(), y.data_ptr(), n, cur_stream()); break;
        case 2: dpa::relu_bwd(a.data_ptr(), b.data_ptr(), y.data_ptr(), n, cur_stream()); break;
        case 3: dpa::gelu_fwd(a.data_ptr(), y.data_ptr(), n, cur_stream()); break;
        case 4: dpa::gelu_bwd(a.data_ptr(), b.data_ptr(), y.data_ptr(), n, cur_stream()); break;
        default: TORCH_CHECK(false, "unknown elementwise op");
    }
    return y;
}

// ---------------------------------------------------------------- loss
// returns (loss_sum_scaled [1] fp32, lse [B] fp32)
std::vector<Tensor> softmax_ce_fwd(const Tensor& logits, const Tensor& targets, int64_t V, double scale, int64_t ignore_index) {
    CHECK_DEV(logits); CHECK_CONTIG(logits); CHECK_I64(targets);
    TORCH_CHECK(logits.dim() == 2, "logits must be [B, ld]");
    const bool bf = logits.scalar_type() == at::kBFloat16;
    TORCH_CHECK(bf || logits.scalar_type() == at::kFloat, "logits must be bf16 or fp32");
    const int64_t B = logits.size(0), ld = logits.size(1);
    TORCH_CHECK(V >= 1 && V <= ld && targets.numel() == B, "softmax_ce: shapes");
    Guard g(logits);
    auto fopt = logits.options().dtype(at::kFloat);
    Tensor loss = torch::zeros({1}, fopt), lse = torch::empty({B}, fopt);
    dpa::softmax_ce_fwd(logits.data_ptr(), bf, targets.data_ptr<int64_t>(), lse.data_ptr<float>(), loss.data_ptr<float>(),
                        B, V, ld, (float)scale, ignore_index, cur_stream());
    if (dpa::deterministic()) {     // the kernel adds one term per row with a float atomic: re-sum them in a fixed order
        Tensor valid = (targets != ignore_index).logical_and(targets >= 0).logical_and(targets < V);
        Tensor picked = logits.narrow(1, 0, V).gather(1, targets.clamp(0, V - 1).unsqueeze(1)).squeeze(1).to(at::kFloat);
        loss = ((lse - picked) * valid.to(at::kFloat)).sum().mul(scale).reshape({1});
    }
    return {loss, lse};
}

Tensor softmax_ce_bwd(const Tensor& logits, const Tensor& targets, const Tensor& lse, const Tensor& gscale, int64_t V,
                      double scale, int64_t ignore_index, bool inplace) {
    CHECK_DEV(logits); CHECK_CONTIG(logits); CHECK_I64(targets); CHECK_F32(lse); CHECK_F32(gscale);
    const bool bf = logits.scalar_type() == at::kBFloat16;
    TORCH_CHECK(bf || logits.scalar_type() == at::kFloat, "logits must be bf16 or fp32");
    const int64_t B = logits.size(0), ld = logits.size(1);
    TORCH_CHECK(V >= 1 && V <= ld && targets.numel() == B && lse.numel() == B && gscale.numel() == 1, "softmax_ce_bwd: shapes");
    Guard g(logits);
    Tensor d = inplace ? logits : torch::empty_like(logits);
    dpa::softmax_ce_bwd(logits.data_ptr(), bf, targets.data_ptr<int64_t>(), lse.data_ptr<float>(), gscale.data_ptr<float>(),
                        d.data_ptr(), B, V, ld, (float)scale, ignore_index, cur_stream());
    return d;
}

// One-pass cross-entropy: returns the loss [1]; `logits` (bf16 [B, ld]) is OVERWRITTEN by d loss / d logits for an
// upstream gradient of 1.  softmax_ce_fused_ok() says whether the shape / mode allows it.
bool softmax_ce_fused_ok(const Tensor& logits, int64_t V) {
    return logits.is_cuda() && logits.is_contiguous() && logits.dim() == 2 &&
           dpa::softmax_ce_fused_supported(logits.scalar_type() == at::kBFloat16, (int)V, logits.size(1), logits.data_ptr());
}
Tensor softmax_ce_fused(Tensor& logits, const Tensor& targets, int64_t V, double scale, int64_t ignore_index) {
    CHECK_I64(targets);
    TORCH_CHECK(softmax_ce_fused_ok(logits, V) && V <= logits.size(1) && targets.numel() == logits.size(0), "softmax_ce_fused: shapes");
    Guard g(logits);
    Tensor loss = torch::zeros({1}, logits.options().dtype(at::kFloat));
    dpa::softmax_ce_fused(logits.data_ptr(), targets.data_ptr<int64_t>(), loss.data_ptr<float>(), logits.size(0), (int)V,
                          logits.size(1), (float)scale, ignore_index, cur_stream());
    return loss;
}
void scale_unless_one(Tensor& d, const Tensor& gscale) {
    CHECK_BF16(d); CHECK_F32(gscale);
    TORCH_CHECK(gscale.numel() == 1 && d.numel() % 8 == 0, "scale_unless_one: shapes");
    Guard g(d);
    dpa::scale_unless_one(d.data_ptr(), gscale.data_ptr<float>(), d.numel(), cur_stream());
}

// ---------------------------------------------------------------- optimizer / casts
void sgd_momentum(Tensor& p, Tensor& grad, const c10::optional<Tensor>& m, const c10::optional<Tensor>& p_lp, double lr,
                  double mu, double wd, double grad_scale, bool zero_grad, bool nesterov,
                  const c10::optional<Tensor>& seg_ends, const c10::optional<Tensor>& seg_wd,
                  const c10::optional<Tensor>& scale_dev) {
    CHECK_F32(p); CHECK_F32(grad);
    const int64_t n = p.numel();
    TORCH_CHECK(grad.numel() == n, "sgd: grad size");
    TORCH_CHECK(((uintptr_t)p.data_ptr()) % 16 == 0 && ((uintptr_t)grad.data_ptr()) % 16 == 0, "sgd: 16-byte alignment");
    float* mp = nullptr; void* lp = nullptr;
    if (mu != 0.0) {
        TORCH_CHECK(m.has_value() && m->defined(), "sgd: momentum buffer required");
        CHECK_F32((*m)); TORCH_CHECK(m->numel() == n && ((uintptr_t)m->data_ptr()) % 16 == 0, "sgd: momentum buffer");
        mp = m->data_ptr<float>();
    }
    if (p_lp.has_value() && p_lp->defined()) {
        CHECK_DEV((*p_lp)); CHECK_CONTIG((*p_lp));
        TORCH_CHECK(p_lp->scalar_type() == at::kBFloat16 && p_lp->numel() == n && ((uintptr_t)p_lp->data_ptr()) % 8 == 0, "sgd: bf16 shadow");
        lp = p_lp->data_ptr();
    }
    const long* se = nullptr; const float* sw = nullptr; int nseg = 0;
    if (seg_ends.has_value() && seg_ends->defined()) {
        CHECK_I64((*seg_ends)); CHECK_F32((*seg_wd));
        nseg = seg_ends->numel();
        TORCH_CHECK(seg_wd->numel() == nseg && nseg >= 1, "sgd: segment table");
        se = (const long*)seg_ends->data_ptr<int64_t>(); sw = seg_wd->data_ptr<float>();
    }
    const float* sd = nullptr;
    if (scale_dev.has_value() && scale_dev->defined()) { CHECK_F32((*scale_dev)); TORCH_CHECK(scale_dev->numel() == 1, "sgd: scale_dev is one float"); sd = scale_dev->data_ptr<float>(); }
    Guard g(p);
    dpa::sgd_momentum(p.data_ptr<float>(), grad.data_ptr<float>(), mp, lp, n, (float)lr, (float)mu, (float)wd,
                      (float)grad_scale, zero_grad, nesterov, se, sw, nseg, sd, cur_stream());
}

void adamw(Tensor& p, Tensor& grad, Tensor& m, Tensor& v, const c10::optional<Tensor>& p_lp, double lr, double b1, double b2,
           double eps, double wd, double grad_scale, int64_t step, bool zero_grad, const c10::optional<Tensor>& scale_dev) {
    CHECK_F32(p); CHECK_F32(grad); CHECK_F32(m); CHECK_F32(v);
    const int64_t n = p.numel();
    TORCH_CHECK(grad.numel() == n && m.numel() == n && v.numel() == n, "adamw: sizes");
    void* lp = nullptr;
    if (p_lp.has_value() && p_lp->defined()) {
        CHECK_DEV((*p_lp)); CHECK_CONTIG((*p_lp));
        TORCH_CHECK(p_lp->scalar_type() == at::kBFloat16 && p_lp->numel() == n, "adamw: bf16 shadow");
        lp = p_lp->data_ptr();
    }
    const float* sd = nullptr;
    if (scale_dev.has_value() && scale_dev->defined()) { CHECK_F32((*scale_dev)); TORCH_CHECK(scale_dev->numel() == 1, "adamw: scale_dev is one float"); sd = scale_dev->data_ptr<float>(); }
    Guard g(p);
    const double bc1 = 1.0 - std::pow(b1, (double)step), bc2 = 1.0 - std::pow(b2, (double)step);
    dpa::adamw(p.data_ptr<float>(), grad.data_ptr<float>(), m.data_ptr<float>(), v.data_ptr<float>(), lp, n, (float)lr,
               (float)b1, (float)b2, (float)eps, (float)wd, (float)grad_scale, (float)bc1, (float)bc2, zero_grad, sd, cur_stream());
}

void cast_f32_to_bf16(const Tensor& x, Tensor& y) {
    CHECK_F32(x); CHECK_DEV(y); CHECK_CONTIG(y);
    TORCH_CHECK(y.scalar_type() == at::kBFloat16 && y.numel() == x.numel(), "cast: shapes");
    Guard g(x);
    dpa::cast_f32_to_bf16(x.data_ptr<float>(), y.data_ptr(), x.numel(), cur_stream());
}

Tensor cast_bf16_to_f32(const Tensor& x) {
    CHECK_DEV(x); CHECK_CONTIG(x); TORCH_CHECK(x.scalar_type() == at::kBFloat16, "bf16 expected");
    Guard g(x);
    Tensor y = torch::empty(x.sizes(), x.options().dtype(at::kFloat));
    dpa::cast_bf16_to_f32(x.data_ptr(), y.data_ptr<float>(), x.numel(), cur_stream());
    return y;
}

// ---------------------------------------------------------------- MNIST nets (fp32 NCHW)
// dropout drawn inside the kernels: p, seed, offset and an optional device step counter (hipGraph-replay safe)
static const int64_t* step_ptr(const c10::optional<Tensor>& step) {
    if (!(step.has_value() && step->defined())) return nullptr;
    TORCH_CHECK(step->is_cuda() && step->scalar_type() == at::kLong && step->numel() == 1, "step must be a device int64 scalar");
    return step->data_ptr<int64_t>();
}
// weight / bias gradient destinations: the caller's accumulators (flat gradient buffer slices) or fresh tensors
static void grad_dst(const c10::optional<Tensor>& acc, const Tensor& like, bool zero, Tensor& out, bool& accumulate) {
    if (acc.has_value() && acc->defined()) {
        CHECK_F32((*acc)); TORCH_CHECK(acc->numel() == like.numel(), "gradient accumulator size");
        out = *acc; accumulate = true;
    } else {
        out = zero ? torch::zeros_like(like) : torch::empty_like(like); accumulate = false;
    }
}

std::vector<Tensor> conv5_pool_relu_fwd(const Tensor& x, const Tensor& w, const Tensor& b, double drop_p, int64_t seed,
                                        int64_t offset, const c10::optional<Tensor>& step) {
    CHECK_F32(x); CHECK_F32(w); CHECK_F32(b);
    TORCH_CHECK(x.dim() == 4 && w.dim() == 4 && w.size(2) == 5 && w.size(3) == 5 && w.size(1) == x.size(1), "conv5: shapes");
    const int64_t N = x.size(0), Cin = x.size(1), H = x.size(2), W = x.size(3), Cout = w.size(0);
    TORCH_CHECK(H >= 6 && W >= 6 && (H - 4) % 2 == 0 && (W - 4) % 2 == 0 && b.numel() == Cout, "conv5: geometry");
    TORCH_CHECK(drop_p >= 0.0 && drop_p < 1.0, "conv5: dropout probability");
    Guard g(x);
    Tensor y = torch::empty({N, Cout, (H - 4) / 2, (W - 4) / 2}, x.options());
    Tensor idx = torch::empty(y.sizes(), x.options().dtype(at::kByte));
    dpa::conv5_pool_relu_fwd(x.data_ptr<float>(), w.data_ptr<float>(), b.data_ptr<float>(), y.data_ptr<float>(),
                             idx.data_ptr<uint8_t>(), N, Cin, H, W, Cout, (float)drop_p, (uint64_t)seed, (uint64_t)offset,
                             step_ptr(step), cur_stream());
    return {y, idx};
}

// returns (dx, dw, db); with dw_acc / db_acc the filter gradients are ADDED there instead (gradient sink) and returned as is
std::vector<Tensor> conv5_pool_relu_bwd(const Tensor& x, const Tensor& w, const Tensor& dy, const Tensor& y, const Tensor& idx,
                                        double gscale, bool need_dx, const c10::optional<Tensor>& dw_acc,
                                        const c10::optional<Tensor>& db_acc) {
    CHECK_F32(x); CHECK_F32(w); CHECK_F32(dy); CHECK_F32(y); CHECK_DEV(idx); CHECK_CONTIG(idx);
    const int64_t N = x.size(0), Cin = x.size(1), H = x.size(2), W = x.size(3), Cout = w.size(0);
    TORCH_CHECK(dy.numel() == y.numel() && idx.numel() == y.numel() && idx.scalar_type() == at::kByte, "conv5 bwd: shapes");
    TORCH_CHECK(y.numel() == N * Cout * ((H - 4) / 2) * ((W - 4) / 2), "conv5 bwd: geometry");
    Guard g(x);
    Tensor dw, db, dx; bool acc_w, acc_b;
    grad_dst(dw_acc, w, true, dw, acc_w);                     // accumulated with atomics either way
    grad_dst(db_acc, torch::empty({Cout}, w.options()), true, db, acc_b);
    if (need_dx) dx = torch::empty_like(x);
    dpa::conv5_pool_relu_bwd(x.data_ptr<float>(), w.data_ptr<float>(), dy.data_ptr<float>(), y.data_ptr<float>(),
                             idx.data_ptr<uint8_t>(), need_dx ? dx.data_ptr<float>() : nullptr, dw.data_ptr<float>(),
                             db.data_ptr<float>(), N, Cin, H, W, Cout, (float)gscale, cur_stream());
    return {dx, dw, db};
}

Tensor linear_f32_fwd(const Tensor& x, const Tensor& w, const Tensor& b, bool relu, double drop_p, int64_t seed, int64_t offset,
                      const c10::optional<Tensor>& step) {
    CHECK_F32(x); CHECK_F32(w); CHECK_F32(b);
    TORCH_CHECK(x.dim() == 2 && w.dim() == 2 && w.size(1) == x.size(1) && b.numel() == w.size(0), "linear_f32: shapes");
    TORCH_CHECK(drop_p >= 0.0 && drop_p < 1.0, "linear_f32: dropout probability");
    Guard g(x);
    Tensor y = torch::empty({x.size(0), w.size(0)}, x.options());
    dpa::linear_f32_fwd(x.data_ptr<float>(), w.data_ptr<float>(), b.data_ptr<float>(), y.data_ptr<float>(), x.size(0),
                        x.size(1), w.size(0), relu, (float)drop_p, (uint64_t)seed, (uint64_t)offset, step_ptr(step), cur_stream());
    return y;
}

// logp = log_softmax(x @ w^T + b, dim=1) in one kernel (classifier heads with <= 64 classes)
Tensor linear_log_softmax_fwd(const Tensor& x, const Tensor& w, const Tensor& b) {
    CHECK_F32(x); CHECK_F32(w); CHECK_F32(b);
    TORCH_CHECK(x.dim() == 2 && w.dim() == 2 && w.size(1) == x.size(1) && b.numel() == w.size(0) && w.size(0) <= 64, "linear_log_softmax: shapes");
    Guard g(x);
    Tensor y = torch::empty({x.size(0), w.size(0)}, x.options());
    dpa::linear_lsm_fwd(x.data_ptr<float>(), w.data_ptr<float>(), b.data_ptr<float>(), y.data_ptr<float>(), x.size(0), x.size(1),
                        w.size(0), cur_stream());
    return y;
}

// (dx, dw, db) in one launch.  lsm: y = log-probabilities, dy = gradient w.r.t. them.  dw_acc / db_acc as for the conv.
std::vector<Tensor> linear_f32_bwd(const Tensor& x, const Tensor& w, const Tensor& y, const Tensor& dy, bool relu, double gscale,
                                   bool lsm, bool need_dx, const c10::optional<Tensor>& dw_acc, const c10::optional<Tensor>& db_acc) {
    CHECK_F32(x); CHECK_F32(w); CHECK_F32(y); CHECK_F32(dy);
    TORCH_CHECK(dy.sizes() == y.sizes() && y.size(0) == x.size(0) && y.size(1) == w.size(0), "linear_f32 bwd: shapes");
    Guard g(x);
    Tensor dw, db, dx; bool acc_w, acc_b;
    grad_dst(dw_acc, w, false, dw, acc_w);
    grad_dst(db_acc, torch::empty({w.size(0)}, w.options()), false, db, acc_b);
    TORCH_CHECK(acc_w == acc_b, "linear_f32 bwd: pass both accumulators or neither");
    if (need_dx) dx = torch::empty_like(x);
    dpa::linear_f32_bwd(x.data_ptr<float>(), w.data_ptr<float>(), y.data_ptr<float>(), dy.data_ptr<float>(),
                        need_dx ? dx.data_ptr<float>() : nullptr, dw.data_ptr<float>(), db.data_ptr<float>(), x.size(0),
                        x.size(1), w.size(0), relu, (float)gscale, lsm, acc_w, nullptr, cur_stream());
    return {dx, dw, db};
}

// The classifier head with its loss: (loss [], logp [rows, out]) = (mean NLL, log_softmax(x @ w^T + b)) in one launch.
// `step`: the dropout stream's device counter, incremented here (models/mnist.py) - or None.
std::vector<Tensor> linear_lsm_nll_fwd(const Tensor& x, const Tensor& w, const Tensor& b, const Tensor& target,
                                       const c10::optional<Tensor>& step) {
    CHECK_F32(x); CHECK_F32(w); CHECK_F32(b); CHECK_I64(target);
    TORCH_CHECK(x.dim() == 2 && w.dim() == 2 && w.size(1) == x.size(1) && b.numel() == w.size(0) && w.size(0) <= 16 &&
                x.size(1) <= 448 && target.numel() == x.size(0) && x.size(0) >= 1, "linear_lsm_nll: shapes (out <= 16, in <= 448)");
    Guard g(x);
    Tensor logp = torch::empty({x.size(0), w.size(0)}, x.options()), loss = torch::empty({}, x.options());
    dpa::linear_lsm_nll_fwd(x.data_ptr<float>(), w.data_ptr<float>(), b.data_ptr<float>(), target.data_ptr<int64_t>(),
                            logp.data_ptr<float>(), loss.data_ptr<float>(), const_cast<int64_t*>(step_ptr(step)), x.size(0),
                            x.size(1), w.size(0), cur_stream());
    return {loss, logp};
}

// its backward: log-softmax, mean-NLL and the Linear's dx | dw | db in one launch; gloss = the scalar loss gradient
std::vector<Tensor> linear_lsm_nll_bwd(const Tensor& x, const Tensor& w, const Tensor& logp, const Tensor& gloss, const Tensor& target,
                                       bool need_dx, const c10::optional<Tensor>& dw_acc, const c10::optional<Tensor>& db_acc) {
    CHECK_F32(x); CHECK_F32(w); CHECK_F32(logp); CHECK_F32(gloss); CHECK_I64(target);
    TORCH_CHECK(gloss.numel() == 1 && logp.size(0) == x.size(0) && logp.size(1) == w.size(0) && w.size(0) <= 16 &&
                target.numel() == x.size(0), "linear_lsm_nll bwd: shapes");
    Guard g(x);
    Tensor dw, db, dx; bool acc_w, acc_b;
    grad_dst(dw_acc, w, false, dw, acc_w);
    grad_dst(db_acc, torch::empty({w.size(0)}, w.options()), false, db, acc_b);
    TORCH_CHECK(acc_w == acc_b, "linear_lsm_nll bwd: pass both accumulators or neither");
    if (need_dx) dx = torch::empty_like(x);
    dpa::linear_f32_bwd(x.data_ptr<float>(), w.data_ptr<float>(), logp.data_ptr<float>(), gloss.data_ptr<float>(),
                        need_dx ? dx.data_ptr<float>() : nullptr, dw.data_ptr<float>(), db.data_ptr<float>(), x.size(0),
                        x.size(1), w.size(0), false, 1.f, 2, acc_w, target.data_ptr<int64_t>(), cur_stream());
    return {dx, dw, db};
}

Tensor log_softmax_fwd(const Tensor& x) {
    CHECK_F32(x); TORCH_CHECK(x.dim() == 2, "log_softmax: 2-D");
    Guard g(x);
    Tensor y = torch::empty_like(x);
    dpa::log_softmax_fwd(x.data_ptr<float>(), y.data_ptr<float>(), x.size(0), x.size(1), cur_stream());
    return y;
}
Tensor log_softmax_bwd(const Tensor& dy, const Tensor& y) {
    CHECK_F32(dy); CHECK_F32(y); TORCH_CHECK(dy.sizes() == y.sizes(), "log_softmax bwd: shapes");
    Guard g(y);
    Tensor dx = torch::empty_like(y);
    dpa::log_softmax_bwd(dy.data_ptr<float>(), y.data_ptr<float>(), dx.data_ptr<float>(), y.size(0), y.size(1), cur_stream());
    return dx;
}
Tensor nll_fwd(const Tensor& logp, const Tensor& t) {
    CHECK_F32(logp); CHECK_I64(t); TORCH_CHECK(logp.dim() == 2 && t.numel() == logp.size(0), "nll: shapes");
    Guard g(logp);
    Tensor loss = torch::empty({}, logp.options());
    dpa::nll_fwd(logp.data_ptr<float>(), t.data_ptr<int64_t>(), loss.data_ptr<float>(), logp.size(0), logp.size(1), cur_stream());
    return loss;
}
Tensor nll_bwd(const Tensor& gout, const Tensor& t, int64_t rows, int64_t V) {
    CHECK_F32(gout); CHECK_I64(t); TORCH_CHECK(gout.numel() == 1 && t.numel() == rows, "nll bwd: shapes");
    Guard g(gout);
    Tensor d = torch::empty({rows, V}, gout.options());
    dpa::nll_bwd(gout.data_ptr<float>(), t.data_ptr<int64_t>(), d.data_ptr<float>(), rows, V, cur_stream());
    return d;
}
Tensor dropout_mask(int64_t n, double p, int64_t seed, int64_t offset, const Tensor& like, const c10::optional<Tensor>& step) {
    CHECK_DEV(like);
    Guard g(like);
    Tensor m = torch::empty({n}, like.options().dtype(at::kFloat));
    const int64_t* sp = nullptr;
    if (step.has_value() && step->defined()) {
        TORCH_CHECK(step->is_cuda() && step->scalar_type() == at::kLong && step->numel() == 1, "dropout_mask: step must be a device int64 scalar");
        sp = step->data_ptr<int64_t>();
    }
    dpa::dropout_mask(m.data_ptr<float>(), n, (float)p, (uint64_t)seed, (uint64_t)offset, sp, cur_stream());
    return m;
}
void counter_add(Tensor& c, int64_t v) {
    TORCH_CHECK(c.is_cuda() && c.scalar_type() == at::kLong && c.numel() == 1, "counter_add: device int64 scalar");
    Guard g(c);
    dpa::counter_add(c.data_ptr<int64_t>(), v, cur_stream());
}
Tensor mul_f32(const Tensor& a, const Tensor& b) {
    CHECK_F32(a); CHECK_F32(b); TORCH_CHECK(a.numel() == b.numel(), "mul: sizes");
    Guard g(a);
    Tensor y = torch::empty_like(a);
    dpa::mul_f32(a.data_ptr<float>(), b.data_ptr<float>(), y.data_ptr<float>(), a.numel(), cur_stream());
    return y;
}

}  // namespace

void bind_transformer(pybind11::module_& m);  // bindings_tfm.cpp
void bind_rccl(pybind11::module_& m);         // rccl_comm.cpp
void bind_peer(pybind11::module_& m);         // peer_comm.cpp

// Launcher temporaries (gemm_wide's split-K slabs) come from torch's caching allocator: visible to torch.cuda.memory_*,
// shared with the step's other temporaries, re-issued only to later work of the stream they were taken for.
static void* torch_workspace_alloc(size_t bytes, hipStream_t stream) {
    return c10::hip::HIPCachingAllocator::raw_alloc_with_stream(bytes, stream);
}
static void torch_workspace_free(void* ptr) { c10::hip::HIPCachingAllocator::raw_delete(ptr); }

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.doc() = "distributedpytorch_amd gfx950 kernels";
    dpa::set_workspace_allocator(&torch_workspace_alloc, &torch_workspace_free);
    m.def("conv_fprop", &conv_fprop, py::arg("x"), py::arg("w"), py::arg("bias"), py::arg("stride"), py::arg("pad"), py::arg("act"),
          py::arg("want_stats"), py::arg("stats_buf"), py::arg("addend") = py::none(), py::arg("gelu_twin") = false,
          py::arg("stride_w") = -1, py::arg("pad_w") = -1, py::arg("out_w") = -1);
    m.def("conv_dgrad", &conv_dgrad, py::arg("dy"), py::arg("wt"), py::arg("H"), py::arg("W"), py::arg("stride"), py::arg("pad"),
          py::arg("addend"), py::arg("addend_mask"), py::arg("bn_x") = py::none(), py::arg("bn_mask") = py::none(),
          py::arg("bn_stats") = py::none(), py::arg("gelu_x") = py::none());
    m.def("bn_bwd_apply", &bn_bwd_apply);
    m.def("set_deterministic", [](bool on) { dpa::set_deterministic(on); });
    m.def("deterministic", []() { return dpa::deterministic(); });
    m.def("bn_backward_coefs", &bn_backward_coefs);
    m.def("tail_bwd_ok", &tail_bwd_ok);
    m.def("tail_bwd", &tail_bwd, py::arg("dout"), py::arg("kout"), py::arg("craw"), py::arg("co"), py::arg("xin"), py::arg("wt"),
          py::arg("dw"), py::arg("bn_x") = py::none(), py::arg("bn_mask") = py::none(), py::arg("bn_stats") = py::none());
    m.def("conv_fprop_stats", &conv_fprop_stats);
    m.def("conv_bn_res_relu", &conv_bn_res_relu);
    m.def("conv_bnb_stats", &conv_bnb_stats);
    m.def("conv_bnb_apply", &conv_bnb_apply);
    m.def("bn_bwd_coefs", &bn_bwd_coefs);
    m.def("conv_wgrad", &conv_wgrad, py::arg("x"), py::arg("dy"), py::arg("dw"), py::arg("stride"), py::arg("pad"),
          py::arg("dbias") = py::none(), py::arg("stride_w") = -1, py::arg("pad_w") = -1);
    m.def("transpose_filter", &transpose_filter);
    m.def("transpose_filters_batched", &transpose_filters_batched);
    m.def("transpose_tile", [] { return dpa::transpose_tile(); });
    m.def("colsum", &colsum);
    m.def("wgrad_wide_group", &wgrad_wide_group, py::arg("xs"), py::arg("dys"), py::arg("dws"), py::arg("dbiases"));
    m.def("wgrad_wide_query", &wgrad_wide_query);
    m.def("set_wgrad_wide_splitk", [](int64_t n) { dpa::set_wgrad_wide_splitk((int)n); });
    m.def("bn_stats", &bn_stats);
    m.def("bn_finalize", &bn_finalize);
    m.def("bn_apply", &bn_apply, py::arg("x"), py::arg("residual"), py::arg("scale"), py::arg("shift"), py::arg("relu"),
          py::arg("want_mask"), py::arg("res_scale") = py::none(), py::arg("res_shift") = py::none());
    m.def("relu_mask_mul", &relu_mask_mul);
    m.def("bn_backward", &bn_backward);
    m.def("bn_relu_maxpool_fwd", &bn_relu_maxpool_fwd);
    m.def("pool_bn_backward", &pool_bn_backward);
    m.def("maxpool_fwd", &maxpool_fwd);
    m.def("maxpool_bwd", &maxpool_bwd);
    m.def("avgpool_fwd", &avgpool_fwd);
    m.def("avgpool_bwd", &avgpool_bwd);
    m.def("nchw_to_nhwc_pad", &nchw_to_nhwc_pad);
    m.def("nchw_to_pairs", &nchw_to_pairs);
    m.def("stem_pack_filter", &stem_pack_filter);
    m.def("stem_unpack_dw", &stem_unpack_dw);
    m.def("ew2", &ew2);
    m.def("softmax_ce_fwd", &softmax_ce_fwd);
    m.def("softmax_ce_bwd", &softmax_ce_bwd);
    m.def("softmax_ce_fused_ok", &softmax_ce_fused_ok);
    m.def("softmax_ce_fused", &softmax_ce_fused);
    m.def("scale_unless_one", &scale_unless_one);
    m.def("sgd_momentum", &sgd_momentum, py::arg("p"), py::arg("grad"), py::arg("m"), py::arg("p_lp"), py::arg("lr"), py::arg("mu"),
          py::arg("wd"), py::arg("grad_scale"), py::arg("zero_grad"), py::arg("nesterov"), py::arg("seg_ends"), py::arg("seg_wd"),
          py::arg("scale_dev") = py::none());
    m.def("adamw", &adamw, py::arg("p"), py::arg("grad"), py::arg("m"), py::arg("v"), py::arg("p_lp"), py::arg("lr"), py::arg("b1"), py::arg("b2"),
          py::arg("eps"), py::arg("wd"), py::arg("grad_scale"), py::arg("step"), py::arg("zero_grad"), py::arg("scale_dev") = py::none());
    m.def("cast_f32_to_bf16", &cast_f32_to_bf16);
    m.def("cast_bf16_to_f32", &cast_bf16_to_f32);
    m.def("conv5_pool_relu_fwd", &conv5_pool_relu_fwd);
    m.def("linear_log_softmax_fwd", &linear_log_softmax_fwd);
    m.def("conv5_pool_relu_bwd", &conv5_pool_relu_bwd);
    m.def("linear_f32_fwd", &linear_f32_fwd);
    m.def("linear_f32_bwd", &linear_f32_bwd);
    m.def("linear_lsm_nll_fwd", &linear_lsm_nll_fwd);
    m.def("linear_lsm_nll_bwd", &linear_lsm_nll_bwd);
    m.def("log_softmax_fwd", &log_softmax_fwd);
    m.def("log_softmax_bwd", &log_softmax_bwd);
    m.def("nll_fwd", &nll_fwd);
    m.def("nll_bwd", &nll_bwd);
    m.def("dropout_mask", &dropout_mask);
    m.def("counter_add", &counter_add);
    m.def("axpy_dev", [](Tensor& dst, const Tensor& src, const Tensor& g) {
        CHECK_F32(dst); CHECK_F32(src); CHECK_F32(g);
        TORCH_CHECK(src.numel() == dst.numel() && g.numel() == 1, "axpy_dev: sizes");
        Guard gd(dst);
        dpa::axpy_dev(dst.data_ptr<float>(), src.data_ptr<float>(), dst.numel(), g.data_ptr<float>(), cur_stream());
    });
    m.def("reduce_shards", [](Tensor& dst, const Tensor& src, int64_t nsrc, int64_t stride) {
        CHECK_F32(dst); CHECK_F32(src);
        TORCH_CHECK(src.numel() >= (nsrc - 1) * stride + dst.numel(), "reduce_shards: src too small");
        Guard g(dst);
        dpa::reduce_shards(dst.data_ptr<float>(), src.data_ptr<float>(), (int)nsrc, stride, dst.numel(), cur_stream());
    });
    m.def("mul_f32", &mul_f32);
    m.def("reduce_shards_bf16", [](Tensor& dst, const Tensor& src, int64_t nsrc, int64_t stride, Tensor& out) {
        CHECK_F32(dst); CHECK_BF16(src); CHECK_BF16(out);
        TORCH_CHECK(nsrc >= 0 && src.numel() >= (nsrc > 0 ? (nsrc - 1) * stride + dst.numel() : 0) && out.numel() >= dst.numel(),
                    "reduce_shards_bf16: buffers too small");
        Guard g(dst);
        dpa::reduce_shards_bf16(dst.data_ptr<float>(), src.data_ptr(), (int)nsrc, stride, dst.numel(), out.data_ptr(), cur_stream());
    });
    m.def("set_conv3x3_halo", [](bool on) { dpa::set_conv3x3_halo(on ? 1 : 0); });
    m.def("conv3x3_halo_enabled", []() { return dpa::conv3x3_halo_enabled() != 0; });
    m.def("set_wide_gemm_min_tiles", [](int64_t n) { dpa::set_wide_gemm_min_tiles((int)n); });
    m.def("set_wide_gemm_splitk", [](int64_t n) { dpa::set_wide_gemm_splitk((int)n); });
    m.def("set_wide_gemm_bn", [](int64_t n) { dpa::set_wide_gemm_bn((int)n); });
    m.def("set_attention_fwd_pipe", [](bool on) { dpa::set_attention_fwd_pipe(on ? 1 : 0); });
    m.def("wide_gemm_min_tiles", []() { return (int64_t)dpa::wide_gemm_min_tiles(); });
    bind_transformer(m);
    bind_rccl(m);
    bind_peer(m);
}
