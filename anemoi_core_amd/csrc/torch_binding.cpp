// Thin PyTorch-ROCm layer over the C ABI (include/anemoi_hip.h): the hot forward entry points as TORCH_LIBRARY ops
// (namespace anemoi_hip) that take at::Tensor, enqueue on c10::hip::getCurrentHIPStream() and allocate their outputs - what
// SURVEY.md section 8(b) sketches as the extension boundary of the reference's registered op (triton/gt.py:390-428) and of the
// torch.nn layers it selects through `layer_kernels` (layers/utils.py:87-142).
//
// Why it exists next to the ctypes binding (anemoi_core_amd/_lib.py, ops.py): argument marshalling.  A ctypes call with 20+
// arguments plus the Python-side shape / stride bookkeeping costs ~10 us per launch, which an EAGER forward (~100 launches) or
// training step (~1 500) pays on the host; here the same checks run in C++.  The kernels, the C ABI and the numerics are the same
// - tests/test_torch_ext_gpu.py holds the two paths bit-identical; a captured hipGraph replay never sees either.
//
// No device code in this file; it links against libanemoi_hip.so (same directory, $ORIGIN).
#include <ATen/ATen.h>
#include <c10/hip/HIPStream.h>
#include <torch/library.h>

#include <tuple>

#include "anemoi_hip.h"

namespace {

using at::Tensor;
using OptTensor = std::optional<Tensor>;

void* cur_stream() { return static_cast<void*>(c10::hip::getCurrentHIPStream().stream()); }

anemoi_dtype_t dt_of(const Tensor& t) {
  switch (t.scalar_type()) {
    case at::kFloat: return ANEMOI_F32;
    case at::kBFloat16: return ANEMOI_BF16;
    case at::kHalf: return ANEMOI_F16;
    default: TORCH_CHECK_VALUE(false, "unsupported dtype ", t.scalar_type(), "; supported: float32, bfloat16, float16");
  }
}

void check(int rc, const char* what) {
  if (rc == ANEMOI_OK) return;
  const char* msg = anemoi_hip_last_error();
  if (rc == ANEMOI_E_INVALID) TORCH_CHECK_VALUE(false, what, ": ", msg);
  if (rc == ANEMOI_E_UNSUPPORTED) TORCH_CHECK_NOT_IMPLEMENTED(false, what, ": ", msg);
  TORCH_CHECK(false, what, ": ", msg, " (code ", rc, ")");
}

// a 2-D row-major view whose last dimension is contiguous: (pointer, leading dimension in elements) - ops._rows
struct Rows {
  const void* p = nullptr;
  int64_t ld = 0;
};

Rows rows(const OptTensor& t, const char* name, c10::ScalarType dtype, const c10::Device& dev) {
  if (!t.has_value() || !t->defined()) return {};
  TORCH_CHECK(t->is_cuda(), "anemoi_core_amd kernels run on an MI355X (ROCm) device only; ", name, " is on '", t->device(),
              "'. There is no CPU fallback in the product path.");
  TORCH_CHECK(t->device() == dev, "tensors on different devices: ", dev, " vs ", t->device());
  TORCH_CHECK_VALUE(t->dim() == 2, name, ": expected a 2-D tensor, got ", t->dim(), " dimensions");
  TORCH_CHECK_VALUE(t->size(1) <= 1 || t->stride(1) == 1, name, ": last dimension must be contiguous");
  TORCH_CHECK_VALUE(t->scalar_type() == dtype, name, ": dtype ", t->scalar_type(), " does not match ", dtype);
  const int64_t ld = t->size(0) > 1 ? t->stride(0) : std::max<int64_t>(t->size(1), t->stride(0));
  return {t->data_ptr(), ld};
}

const void* vec(const OptTensor& t, const char* name, int64_t n, c10::ScalarType dtype) {
  if (!t.has_value() || !t->defined()) return nullptr;
  TORCH_CHECK_VALUE(t->dim() == 1 && t->size(0) == n && t->is_contiguous() && t->scalar_type() == dtype, name, ": expected contiguous [", n,
                    "] ", dtype);
  return t->data_ptr();
}

void on_current_device(const Tensor& x) {
  TORCH_CHECK(x.is_cuda(), "anemoi_core_amd kernels run on an MI355X (ROCm) device only; got a tensor on '", x.device(),
              "'. There is no CPU fallback in the product path.");
  TORCH_CHECK(x.device().index() == c10::hip::current_device(), "tensors live on ", x.device(), " but the current device is cuda:",
              (int)c10::hip::current_device(), "; make it current before calling anemoi_core_amd ops");
}

// y = act([x | x2] W^T + bias + g1[idx1] + g2[idx2]) + residual            (anemoi_linear_fwd; act: 0 none, 1 GELU)
Tensor linear_impl(const Tensor& x, const Tensor& weight, const OptTensor& bias, int64_t act, const OptTensor& residual, const OptTensor& x2,
                   const OptTensor& g1, const OptTensor& idx1, const OptTensor& g2, const OptTensor& idx2, const OptTensor& out) {
  on_current_device(x);
  const auto dt = x.scalar_type();
  const auto dev = x.device();
  TORCH_CHECK_VALUE(x.dim() == 2 && weight.dim() == 2, "linear: x and weight must be 2-D");
  const int64_t N = x.size(0), K1 = x.size(1), K2 = (x2.has_value() && x2->defined()) ? x2->size(1) : 0, O = weight.size(0);
  TORCH_CHECK_VALUE(weight.size(1) == K1 + K2, "weight is [", O, ", ", weight.size(1), "], expected [", O, ", ", K1 + K2, "]");
  TORCH_CHECK_VALUE(act == ANEMOI_ACT_NONE || act == ANEMOI_ACT_GELU, "unsupported activation ", act);
  if (x2.has_value() && x2->defined()) TORCH_CHECK_VALUE(x2->size(0) == N, "x2 has ", x2->size(0), " rows, expected ", N);
  if (residual.has_value() && residual->defined())
    TORCH_CHECK_VALUE(residual->size(0) == N && residual->size(1) == O, "residual must be [", N, ", ", O, "]");
  const void* ip[2] = {nullptr, nullptr};
  const OptTensor* gs[2] = {&g1, &g2};
  const OptTensor* is[2] = {&idx1, &idx2};
  for (int k = 0; k < 2; ++k) {
    const bool has_g = gs[k]->has_value() && (*gs[k])->defined(), has_i = is[k]->has_value() && (*is[k])->defined();
    TORCH_CHECK_VALUE(has_g == has_i, "g", k + 1, " and its index must be given together");
    if (has_g) {
      const Tensor& idx = **is[k];
      TORCH_CHECK_VALUE((*gs[k])->size(1) == O && idx.scalar_type() == at::kInt && idx.dim() == 1 && idx.size(0) == N && idx.is_contiguous(),
                        "g", k + 1, ": table must be [*, ", O, "] and index contiguous int32 [", N, "]");
      ip[k] = idx.data_ptr();
    }
  }
  Tensor y = (out.has_value() && out->defined()) ? *out : at::empty({N, O}, x.options());
  const Rows rx = rows(x, "x", dt, dev), rx2 = rows(x2, "x2", dt, dev), rw = rows(weight, "weight", dt, dev), rg1 = rows(g1, "g1", dt, dev),
             rg2 = rows(g2, "g2", dt, dev), rr = rows(residual, "residual", dt, dev), ry = rows(y, "out", dt, dev);
  check(anemoi_linear_fwd(rx.p, rx.ld, (int32_t)K1, rx2.p, rx2.ld, (int32_t)K2, rw.p, rw.ld, vec(bias, "bias", O, dt), rg1.p, rg1.ld,
                          (const int32_t*)ip[0], rg2.p, rg2.ld, (const int32_t*)ip[1], rr.p, rr.ld, const_cast<void*>(ry.p), ry.ld, (int32_t)N,
                          (int32_t)O, (anemoi_act_t)act, dt_of(x), cur_stream()),
        "linear_fwd");
  return y;
}

Tensor linear(const Tensor& x, const Tensor& weight, const OptTensor& bias, int64_t act, const OptTensor& residual, const OptTensor& x2,
              const OptTensor& g1, const OptTensor& idx1, const OptTensor& g2, const OptTensor& idx2) {
  return linear_impl(x, weight, bias, act, residual, x2, g1, idx1, g2, idx2, std::nullopt);
}

void linear_out(const Tensor& x, const Tensor& weight, const OptTensor& bias, int64_t act, const OptTensor& residual, const OptTensor& x2,
                const OptTensor& g1, const OptTensor& idx1, const OptTensor& g2, const OptTensor& idx2, Tensor out) {
  TORCH_CHECK_VALUE(out.dim() == 2 && out.size(0) == x.size(0) && out.size(1) == weight.size(0), "linear: out must be [rows(x), rows(weight)]");
  linear_impl(x, weight, bias, act, residual, x2, g1, idx1, g2, idx2, out);
}

// LayerNorm over the last dimension (+ residual), fp32 statistics                                  (anemoi_layernorm_fwd)
Tensor layer_norm_impl(const Tensor& x, const Tensor& weight, const OptTensor& bias, double eps, const OptTensor& residual, const OptTensor& out) {
  on_current_device(x);
  const auto dt = x.scalar_type();
  const int64_t D = x.size(-1);
  Tensor x2 = x.reshape({-1, D});
  if (x2.size(1) > 1 && x2.stride(1) != 1) x2 = x2.contiguous();
  const int64_t n = x2.size(0);
  if (out.has_value() && out->defined())
    TORCH_CHECK_VALUE(out->dim() == 2 && out->size(0) == n && out->size(1) == D && out->scalar_type() == dt && out->is_contiguous(),
                      "layer_norm: out must be a contiguous [rows, D] tensor of x's dtype");
  Tensor y = (out.has_value() && out->defined()) ? *out : at::empty({n, D}, x.options());
  OptTensor r2;
  if (residual.has_value() && residual->defined()) {
    TORCH_CHECK_VALUE(residual->sizes() == x.sizes(), "residual shape does not match x");
    r2 = residual->reshape({-1, D});
  }
  const Rows rx = rows(x2, "x", dt, x.device()), rr = rows(r2, "residual", dt, x.device());
  check(anemoi_layernorm_fwd(rx.p, rx.ld, vec(weight, "weight", D, dt), vec(bias, "bias", D, dt), rr.p, rr.ld, y.data_ptr(), D, (int32_t)n,
                             (int32_t)D, (float)eps, dt_of(x), cur_stream()),
        "layernorm_fwd");
  return y.view(x.sizes());
}

Tensor layer_norm(const Tensor& x, const Tensor& weight, const OptTensor& bias, double eps, const OptTensor& residual) {
  return layer_norm_impl(x, weight, bias, eps, residual, std::nullopt);
}

void layer_norm_out(const Tensor& x, const Tensor& weight, const OptTensor& bias, double eps, const OptTensor& residual, Tensor out) {
  layer_norm_impl(x, weight, bias, eps, residual, out);
}

// lin_edge fused into the edge attention (+ self term)                                 (anemoi_gt_attention_fused_edge_fwd)
std::tuple<Tensor, Tensor> gt_attention_fused_edge(const Tensor& q, const Tensor& k, const Tensor& v, const Tensor& edge_feat, const Tensor& w_packed,
                                                   const Tensor& row, const Tensor& colptr, const OptTensor& order, int64_t n_src, int64_t num_heads,
                                                   const OptTensor& addend, bool return_lse) {
  on_current_device(q);
  const auto dt = q.scalar_type();
  const auto dev = q.device();
  const int64_t n_dst = q.size(0), D = q.size(1), M = row.size(0), fe_pad = edge_feat.dim() == 2 ? edge_feat.size(1) : 0;
  TORCH_CHECK_VALUE(D % num_heads == 0, "channels ", D, " not divisible by heads ", num_heads);
  TORCH_CHECK_VALUE(edge_feat.scalar_type() == at::kFloat && edge_feat.dim() == 2 && edge_feat.size(0) == M && edge_feat.is_contiguous(),
                    "edge_feat must be contiguous fp32 [", M, ", fe_pad]");
  TORCH_CHECK_VALUE(w_packed.scalar_type() == at::kFloat && w_packed.dim() == 2 && w_packed.size(0) == D && w_packed.size(1) == fe_pad &&
                        w_packed.is_contiguous(),
                    "w_packed must be contiguous fp32 [", D, ", ", fe_pad, "]");
  TORCH_CHECK_VALUE(colptr.size(0) == n_dst + 1 && k.size(0) == n_src && v.size(0) == n_src, "node counts do not match the graph");
  // a CPU csc.row or a narrower k would end in a GPU memory fault instead of an error (the ctypes path checks through _dev())
  TORCH_CHECK_VALUE(row.device() == dev && colptr.device() == dev && edge_feat.device() == dev && w_packed.device() == dev &&
                        (!order.has_value() || !order->defined() || order->device() == dev),
                    "row / colptr / order / edge_feat / w_packed must be on q's device");
  TORCH_CHECK_VALUE(k.dim() == 2 && v.dim() == 2 && k.size(1) == D && v.size(1) == D, "k and v must have ", D, " columns like q");
  TORCH_CHECK_VALUE(!addend.has_value() || !addend->defined() || (addend->dim() == 2 && addend->size(0) == n_dst && addend->size(1) == D),
                    "addend must be [", n_dst, ", ", D, "]");
  TORCH_CHECK_VALUE(row.scalar_type() == at::kInt && colptr.scalar_type() == at::kInt && row.is_contiguous() && colptr.is_contiguous(),
                    "row / colptr must be contiguous int32");
  const int32_t* ord = nullptr;
  if (order.has_value() && order->defined()) {
    TORCH_CHECK_VALUE(order->scalar_type() == at::kInt && order->dim() == 1 && order->size(0) == n_dst && order->is_contiguous(),
                      "order must be contiguous int32 [n_dst]");
    ord = order->data_ptr<int32_t>();
  }
  Tensor out = at::empty({n_dst, D}, q.options());
  Tensor lse = at::empty({return_lse ? n_dst : 0, num_heads}, q.options().dtype(at::kFloat));
  const Rows rq = rows(q, "q", dt, dev), rk = rows(k, "k", dt, dev), rv = rows(v, "v", dt, dev), ra = rows(addend, "addend", dt, dev);
  check(anemoi_gt_attention_fused_edge_fwd(rq.p, rq.ld, rk.p, rk.ld, rv.p, rv.ld, edge_feat.data_ptr<float>(), (int32_t)fe_pad,
                                           w_packed.data_ptr<float>(), row.data_ptr<int32_t>(), colptr.data_ptr<int32_t>(), ord, ra.p, ra.ld,
                                           out.data_ptr(), D, return_lse ? lse.data_ptr<float>() : nullptr, (int32_t)n_dst, (int32_t)n_src,
                                           (int32_t)num_heads, (int32_t)(D / num_heads), dt_of(q), cur_stream()),
        "gt_attention_fused_edge_fwd");
  return {out, lse};
}

// y = x W^T + bias [+ residual] and the per-strip row statistics of y; 1-D empty tensors = shape not eligible  (anemoi_linear_stats_fwd)
std::tuple<Tensor, Tensor> linear_with_row_stats(const Tensor& x, const Tensor& weight, const OptTensor& bias, const OptTensor& residual) {
  on_current_device(x);
  const auto dt = x.scalar_type();
  const int64_t N = x.size(0), K = x.size(1), O = weight.size(0);
  const auto not_eligible = [&] { return std::make_tuple(at::empty({0}, x.options()), at::empty({0}, x.options().dtype(at::kFloat))); };
  if (dt == at::kFloat || O % 64 || K % 64) return not_eligible();
  Tensor y = at::empty({N, O}, x.options());
  Tensor stats = at::empty({N, O / 64, 2}, x.options().dtype(at::kFloat));
  const Rows rx = rows(x, "x", dt, x.device()), rw = rows(weight, "weight", dt, x.device()), rr = rows(residual, "residual", dt, x.device());
  const int rc = anemoi_linear_stats_fwd(rx.p, rx.ld, (int32_t)K, rw.p, rw.ld, vec(bias, "bias", O, dt), rr.p, rr.ld, y.data_ptr(), O,
                                         stats.data_ptr<float>(), (int32_t)N, (int32_t)O, dt_of(x), cur_stream());
  if (rc == ANEMOI_E_UNSUPPORTED) return not_eligible();
  check(rc, "linear_stats_fwd");
  return {y, stats};
}

// act(LayerNorm(x) W^T + b) from raw x and the producer's statistics; a 1-D empty tensor = shape not eligible             (anemoi_linear_lnfold_fwd)
Tensor linear_ln_folded(const Tensor& x, const Tensor& w_scaled, const Tensor& c, const Tensor& d, const Tensor& stats, double eps, int64_t act) {
  on_current_device(x);
  const auto dt = x.scalar_type();
  const int64_t N = x.size(0), K = x.size(1), O = w_scaled.size(0);
  TORCH_CHECK_VALUE(stats.scalar_type() == at::kFloat && stats.dim() == 3 && stats.size(0) == N && stats.size(1) == K / 64 && stats.size(2) == 2 &&
                        stats.is_contiguous(),
                    "stats must be contiguous fp32 [N, K/64, 2]");
  TORCH_CHECK_VALUE(c.scalar_type() == at::kFloat && d.scalar_type() == at::kFloat && c.numel() == O && d.numel() == O && c.is_contiguous() &&
                        d.is_contiguous(),
                    "c and d must be contiguous fp32 [O]");
  Tensor y = at::empty({N, O}, x.options());
  const Rows rx = rows(x, "x", dt, x.device()), rw = rows(w_scaled, "w_scaled", dt, x.device());
  const int rc = anemoi_linear_lnfold_fwd(rx.p, rx.ld, (int32_t)K, rw.p, rw.ld, c.data_ptr<float>(), d.data_ptr<float>(), stats.data_ptr<float>(),
                                          (int32_t)(K / 64), (float)eps, (anemoi_act_t)act, y.data_ptr(), O, (int32_t)N, (int32_t)O, dt_of(x),
                                          cur_stream());
  if (rc == ANEMOI_E_UNSUPPORTED) return at::empty({0}, x.options());
  check(rc, "linear_lnfold_fwd");
  return y;
}

[[noreturn]] void no_cpu() {
  TORCH_CHECK(false, "anemoi_core_amd kernels run on an MI355X (ROCm) device only; got a CPU tensor. There is no CPU fallback in the product path.");
}

}  // namespace

TORCH_LIBRARY(anemoi_hip, m) {
  m.def("linear(Tensor x, Tensor weight, Tensor? bias, int act, Tensor? residual, Tensor? x2, Tensor? g1, Tensor? idx1, Tensor? g2, Tensor? idx2) -> Tensor");
  m.def("linear_out(Tensor x, Tensor weight, Tensor? bias, int act, Tensor? residual, Tensor? x2, Tensor? g1, Tensor? idx1, Tensor? g2, Tensor? idx2, "
        "Tensor(a!) out) -> ()");
  m.def("layer_norm(Tensor x, Tensor weight, Tensor? bias, float eps, Tensor? residual) -> Tensor");
  m.def("layer_norm_out(Tensor x, Tensor weight, Tensor? bias, float eps, Tensor? residual, Tensor(a!) out) -> ()");
  m.def("gt_attention_fused_edge(Tensor q, Tensor k, Tensor v, Tensor edge_feat, Tensor w_packed, Tensor row, Tensor colptr, Tensor? order, "
        "int n_src, int num_heads, Tensor? addend, bool return_lse) -> (Tensor, Tensor)");
  m.def("linear_with_row_stats(Tensor x, Tensor weight, Tensor? bias, Tensor? residual) -> (Tensor, Tensor)");
  m.def("linear_ln_folded(Tensor x, Tensor w_scaled, Tensor c, Tensor d, Tensor stats, float eps, int act) -> Tensor");
}

TORCH_LIBRARY_IMPL(anemoi_hip, CUDA, m) {  // the ROCm build of PyTorch dispatches HIP tensors under the CUDA key
  m.impl("linear", &linear);
  m.impl("linear_out", &linear_out);
  m.impl("layer_norm", &layer_norm);
  m.impl("layer_norm_out", &layer_norm_out);
  m.impl("gt_attention_fused_edge", &gt_attention_fused_edge);
  m.impl("linear_with_row_stats", &linear_with_row_stats);
  m.impl("linear_ln_folded", &linear_ln_folded);
}

TORCH_LIBRARY_IMPL(anemoi_hip, CPU, m) {  // fail loudly, like the ctypes path
  m.impl("linear", [](const Tensor&, const Tensor&, const OptTensor&, int64_t, const OptTensor&, const OptTensor&, const OptTensor&, const OptTensor&,
                      const OptTensor&, const OptTensor&) -> Tensor { no_cpu(); });
  m.impl("layer_norm", [](const Tensor&, const Tensor&, const OptTensor&, double, const OptTensor&) -> Tensor { no_cpu(); });
}
