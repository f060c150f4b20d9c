// Bindings for the transformer (GPT-2) kernels: LayerNorm, embedding, causal attention.
#include <torch/extension.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>

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

// x [rows, C] bf16 -> (y bf16, mean fp32 [rows], rstd fp32 [rows])
std::vector<Tensor> layernorm_fwd(const Tensor& x, const Tensor& g, const Tensor& b, double eps) {
    CHECK_BF16(x); CHECK_F32(g); CHECK_F32(b);
    const int64_t C = x.size(-1), rows = x.numel() / C;
    TORCH_CHECK(C % 8 == 0 && C <= 2048 && g.numel() == C && b.numel() == C, "layernorm: C must be a multiple of 8, <= 2048");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(x.device());
    auto fopt = x.options().dtype(at::kFloat);
    Tensor y = torch::empty_like(x), mean = torch::empty({rows}, fopt), rstd = torch::empty({rows}, fopt);
    dpa::layernorm_fwd(x.data_ptr(), g.data_ptr<float>(), b.data_ptr<float>(), y.data_ptr(), mean.data_ptr<float>(),
                       rstd.data_ptr<float>(), rows, C, (float)eps, cur_stream());
    return {y, mean, rstd};
}

// returns (dx bf16, dgamma fp32 [C], dbeta fp32 [C]); with dgamma_acc / dbeta_acc (fp32 [C], e.g. slices of the
// flat gradient buffer) the parameter gradients are ADDED there instead and the two returned tensors are empty
// `addend` (bf16, x's shape): dx += addend - the gradient of the residual connection that forks off in front of the LayerNorm
std::vector<Tensor> layernorm_bwd(const Tensor& dy, const Tensor& x, const Tensor& g, const Tensor& mean, const Tensor& rstd,
                                  const c10::optional<Tensor>& dgamma_acc, const c10::optional<Tensor>& dbeta_acc,
                                  const c10::optional<Tensor>& addend, bool defer_reduce) {
    CHECK_BF16(dy); CHECK_BF16(x); CHECK_F32(g); CHECK_F32(mean); CHECK_F32(rstd);
    const int64_t C = x.size(-1), rows = x.numel() / C;
    TORCH_CHECK(dy.numel() == x.numel() && C % 8 == 0 && C <= 2048 && g.numel() == C && mean.numel() == rows && rstd.numel() == rows,
                "layernorm_bwd: shapes");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(x.device());
    auto fopt = x.options().dtype(at::kFloat);
    const int pr = dpa::layernorm_partial_rows(rows);
    Tensor dx = torch::empty_like(x), part = torch::empty({2, pr, C}, fopt);
    float* pp = part.data_ptr<float>();
    const void* ad = nullptr;
    if (addend.has_value() && addend->defined()) {
        CHECK_BF16((*addend));
        TORCH_CHECK(addend->numel() == x.numel(), "layernorm_bwd: addend shape");
        ad = addend->data_ptr();
    }
    dpa::layernorm_bwd(dy.data_ptr(), x.data_ptr(), g.data_ptr<float>(), mean.data_ptr<float>(), rstd.data_ptr<float>(),
                       dx.data_ptr(), pp, pp + (long)pr * C, pr, rows, C, ad, cur_stream());
    if (defer_reduce) return {dx, part};   // the caller folds the partial sums later (ln_fold_partials), off the critical path
    const bool acc = dgamma_acc.has_value() && dgamma_acc->defined() && dbeta_acc.has_value() && dbeta_acc->defined();
    if (acc) {
        CHECK_F32((*dgamma_acc)); CHECK_F32((*dbeta_acc));
        TORCH_CHECK(dgamma_acc->numel() == C && dbeta_acc->numel() == C, "layernorm_bwd: accumulator shapes");
        dpa::reduce_partials2(pp, pr, C, dgamma_acc->data_ptr<float>(), dbeta_acc->data_ptr<float>(), 1, cur_stream());
        return {dx, Tensor(), Tensor()};
    }
    Tensor dgb = torch::empty({2, C}, fopt);
    float* o = dgb.data_ptr<float>();
    dpa::reduce_partials2(pp, pr, C, o, o + C, 0, cur_stream());
    return {dx, dgb[0], dgb[1]};
}

// part [2, rows, C] (what layernorm_bwd(..., defer_reduce=true) returned): column sums ADDED into dgamma_acc / dbeta_acc.
// Nobody needs a LayerNorm's parameter gradients before the bucket all-reduce, so ops/norm.py runs this on the side
// stream instead of between two kernels of the backward chain.
void ln_fold_partials(const Tensor& part, Tensor& dgamma_acc, Tensor& dbeta_acc) {
    CHECK_F32(part); CHECK_F32(dgamma_acc); CHECK_F32(dbeta_acc);
    TORCH_CHECK(part.dim() == 3 && part.size(0) == 2, "ln_fold_partials: part must be [2, rows, C]");
    const int64_t pr = part.size(1), C = part.size(2);
    TORCH_CHECK(dgamma_acc.numel() == C && dbeta_acc.numel() == C, "ln_fold_partials: accumulator shapes");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(part.device());
    const float* pp = part.data_ptr<float>();
    dpa::reduce_partials2(pp, (int)pr, (int)C, dgamma_acc.data_ptr<float>(), dbeta_acc.data_ptr<float>(), 1, cur_stream());
}

// tokens [B,T] int64, wte [V,C] bf16, wpe [Tmax,C] bf16 -> [B,T,C] bf16
Tensor embedding_fwd(const Tensor& tok, const Tensor& wte, const Tensor& wpe) {
    CHECK_I64(tok); CHECK_BF16(wte); CHECK_BF16(wpe);
    TORCH_CHECK(tok.dim() == 2 && wte.dim() == 2 && wpe.dim() == 2 && wte.size(1) == wpe.size(1), "embedding: shapes");
    const int64_t B = tok.size(0), T = tok.size(1), C = wte.size(1);
    TORCH_CHECK(C % 8 == 0 && T <= wpe.size(0), "embedding: C % 8, T <= positions");
    // token range is validated on the host only in debug runs (it needs a sync); the kernel clamps.
    c10::hip::HIPGuardMasqueradingAsCUDA guard(wte.device());
    Tensor y = torch::empty({B, T, C}, wte.options());
    dpa::embedding_fwd(tok.data_ptr<int64_t>(), wte.data_ptr(), wpe.data_ptr(), y.data_ptr(), B, T, C, wte.size(0), cur_stream());
    return y;
}

// accumulates into fp32 dwte [V,C], dwpe [Tmax,C]
void embedding_bwd(const Tensor& tok, const Tensor& dy, Tensor& dwte, Tensor& dwpe) {
    CHECK_I64(tok); CHECK_BF16(dy); CHECK_F32(dwte); CHECK_F32(dwpe);
    const int64_t B = tok.size(0), T = tok.size(1), C = dy.size(-1);
    TORCH_CHECK(dy.numel() == B * T * C && dwte.size(1) == C && dwpe.size(1) == C && T <= dwpe.size(0) && C % 8 == 0, "embedding_bwd: shapes");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(dy.device());
    dpa::embedding_bwd(tok.data_ptr<int64_t>(), dy.data_ptr(), dwte.data_ptr<float>(), dwpe.data_ptr<float>(), B, T, C, dwte.size(0), cur_stream());
}

// y = (addend +) x * keep / (1 - p), the mask drawn in line from the counter hash of (seed, offset, element index)
Tensor dropout_bf16(const Tensor& x, double p, int64_t seed, int64_t offset, const c10::optional<Tensor>& addend) {
    CHECK_BF16(x);
    TORCH_CHECK(p >= 0.0 && p < 1.0 && x.numel() % 8 == 0, "dropout_bf16: p in [0, 1), numel % 8 == 0");
    const void* ad = nullptr;
    if (addend.has_value() && addend->defined()) {
        CHECK_BF16((*addend));
        TORCH_CHECK(addend->numel() == x.numel(), "dropout_bf16: addend shape");
        ad = addend->data_ptr();
    }
    c10::hip::HIPGuardMasqueradingAsCUDA guard(x.device());
    Tensor y = torch::empty_like(x);
    dpa::dropout_bf16(x.data_ptr(), ad, y.data_ptr(), x.numel(), (float)p, (uint64_t)seed, (uint64_t)offset, cur_stream());
    return y;
}

#ifdef DPA_HAVE_ATTENTION
// qkv [B,T,3,H,D] bf16 -> (out [B,T,H,D] bf16, lse [B,H,T] fp32)
std::vector<Tensor> attention_fwd(const Tensor& qkv, double drop_p, int64_t seed, int64_t offset) {
    CHECK_BF16(qkv);
    TORCH_CHECK(qkv.dim() == 5 && qkv.size(2) == 3 && (qkv.size(4) == 64 || qkv.size(4) == 128), "attention: qkv must be [B,T,3,H,64|128]");
    const int64_t B = qkv.size(0), T = qkv.size(1), H = qkv.size(3), D = qkv.size(4);
    TORCH_CHECK(T >= 1 && B >= 1 && H >= 1 && B <= 65535 && qkv.numel() < (1ll << 31), "attention: empty or oversized problem");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(qkv.device());
    Tensor out = torch::empty({B, T, H, D}, qkv.options());
    Tensor lse = torch::empty({B, H, T}, qkv.options().dtype(at::kFloat));
    TORCH_CHECK(drop_p >= 0.0 && drop_p < 1.0, "attention: dropout probability");
    dpa::attention_fwd(qkv.data_ptr(), out.data_ptr(), lse.data_ptr<float>(), B, T, H, D, (float)drop_p, (uint64_t)seed, (uint64_t)offset, cur_stream());
    return {out, lse};
}

Tensor attention_bwd(const Tensor& qkv, const Tensor& out, const Tensor& dout, const Tensor& lse, double drop_p, int64_t seed, int64_t offset) {
    CHECK_BF16(qkv); CHECK_BF16(out); CHECK_BF16(dout); CHECK_F32(lse);
    TORCH_CHECK(qkv.dim() == 5 && qkv.size(2) == 3 && (qkv.size(4) == 64 || qkv.size(4) == 128), "attention_bwd: qkv must be [B,T,3,H,64|128]");
    const int64_t B = qkv.size(0), T = qkv.size(1), H = qkv.size(3), D = qkv.size(4);
    TORCH_CHECK(T >= 1 && out.numel() == B * T * H * D && dout.numel() == out.numel() && lse.numel() == B * H * T, "attention_bwd: shapes");
    c10::hip::HIPGuardMasqueradingAsCUDA guard(qkv.device());
    Tensor dqkv = torch::empty_like(qkv);
    Tensor delta = torch::empty({B, H, T}, lse.options());
    dpa::attention_bwd(qkv.data_ptr(), out.data_ptr(), dout.data_ptr(), lse.data_ptr<float>(), dqkv.data_ptr(),
                       delta.data_ptr<float>(), B, T, H, D, (float)drop_p, (uint64_t)seed, (uint64_t)offset, cur_stream());
    return dqkv;
}
#endif
}  // namespace

void bind_transformer(pybind11::module_& m) {
    m.def("layernorm_fwd", &layernorm_fwd);
    m.def("layernorm_bwd", &layernorm_bwd, pybind11::arg("dy"), pybind11::arg("x"), pybind11::arg("g"), pybind11::arg("mean"),
          pybind11::arg("rstd"), pybind11::arg("dgamma_acc"), pybind11::arg("dbeta_acc"), pybind11::arg("addend") = pybind11::none(),
          pybind11::arg("defer_reduce") = false);
    m.def("ln_fold_partials", &ln_fold_partials);
    m.def("embedding_fwd", &embedding_fwd);
    m.def("embedding_bwd", &embedding_bwd);
    m.def("dropout_bf16", &dropout_bf16, pybind11::arg("x"), pybind11::arg("p"), pybind11::arg("seed"), pybind11::arg("offset"),
          pybind11::arg("addend") = pybind11::none());
#ifdef DPA_HAVE_ATTENTION
    m.def("attention_fwd", &attention_fwd, pybind11::arg("qkv"), pybind11::arg("drop_p") = 0.0, pybind11::arg("seed") = 0,
          pybind11::arg("offset") = 0);
    m.def("attention_bwd", &attention_bwd, pybind11::arg("qkv"), pybind11::arg("out"), pybind11::arg("dout"), pybind11::arg("lse"),
          pybind11::arg("drop_p") = 0.0, pybind11::arg("seed") = 0, pybind11::arg("offset") = 0);
#endif
}
