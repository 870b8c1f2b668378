// torch.ops.lrx.*: the PyTorch-ROCm custom-op binding of the C ABI in include/lrx.h (last row).
// A thin layer: tensors in, the current HIP stream, workspaces from PyTorch's caching allocator, errors -> TORCH_CHECK
// (Python RuntimeError).  Every op forwards to exactly one lrx_* entry point of liblrx.so; no arithmetic lives here.
// Built by lightretriever_amd/build.py into lightretriever_amd/liblrx_torch.so (next to liblrx.so, rpath $ORIGIN).
#include <ATen/ATen.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <c10/hip/HIPStream.h>
#include <torch/library.h>

#include "../../include/lrx.h"

namespace {

// every op first makes the device of its first tensor current (getCurrentHIPStream() belongs to the CURRENT device, which need
// not be the tensors' device) and then takes that device's current stream
using DevGuard = c10::hip::OptionalHIPGuardMasqueradingAsCUDA;
void* cur_stream() { return (void*)c10::hip::getCurrentHIPStream().stream(); }

void lrx_check(int rc, const char* what) { TORCH_CHECK(rc == LRX_OK, what, ": liblrx error ", rc, ": ", lrx_last_error()); }

void need(const at::Tensor& t, const char* name, at::ScalarType dt, int64_t dim = -1) {
  TORCH_CHECK(t.is_cuda(), name, " must be a device tensor");
  TORCH_CHECK(t.scalar_type() == dt, name, " must be ", dt, ", got ", t.scalar_type());
  TORCH_CHECK(dim < 0 || t.dim() == dim, name, " must have ", dim, " dimensions");
  TORCH_CHECK(t.is_contiguous() || (t.dim() == 2 && t.stride(1) == 1), name, " must be contiguous (rows may be strided)");
}

at::Tensor bytes(int64_t n, const at::Tensor& like) { return at::empty({n}, like.options().dtype(at::kByte)); }

// ---- encoder ---------------------------------------------------------------------------------------------------------------------
// weights: the address of an lrx_encoder_handle {cfg*, weights*} kept alive by its owner (LrxEncoder.handle)
// shadow / row_bounds (optional): `out` = rows [shadow_row0, ...) of an index shard -- the last kernel also writes their tiled shadow rows and
// raises the shard's bounds (lrx_encode_packed_shard), like the ctypes path does for a live FlatIPIndex slot.  Without them a caller that
// encodes into committed shard rows must run shard_commit_rows afterwards.
void encode_packed(const at::Tensor& ids, const at::Tensor& cu_seqlens, int64_t max_seqlen, int64_t weights, at::Tensor out, int64_t mrl_dim,
                   bool normalize, const c10::optional<at::Tensor>& shadow, int64_t shadow_row0, const c10::optional<at::Tensor>& row_bounds) {
  DevGuard guard(ids.device());
  need(ids, "ids", at::kInt, 1);
  need(cu_seqlens, "cu_seqlens", at::kInt, 1);
  need(out, "out", at::kFloat, 2);
  TORCH_CHECK(weights != 0, "encode_packed: null weights handle");
  const lrx_encoder_handle* h = (const lrx_encoder_handle*)weights;
  const int64_t T = ids.numel(), B = cu_seqlens.numel() - 1;
  const int64_t D = mrl_dim > 0 ? mrl_dim : h->cfg->hidden_size;
  TORCH_CHECK(out.size(0) >= B && out.size(1) >= D, "encode_packed: out must be fp32 [>= n_seqs, >= mrl_dim]");
  const size_t need_b = lrx_encode_workspace_bytes(h->cfg, (int32_t)T, (int32_t)B);
  at::Tensor ws = bytes((int64_t)need_b, ids);
  void* sh = nullptr;
  if (shadow.has_value()) {
    TORCH_CHECK(shadow->is_cuda() && shadow->dim() == 1 && shadow->is_contiguous() && shadow->element_size() == 2 && D % 64 == 0 && D == out.size(1) &&
                    shadow_row0 >= 0 && shadow->numel() >= ((shadow_row0 + B + 127) / 128) * 128 * D,
                "encode_packed: shadow must be the shard's 1-D tiled 16-bit shadow (include/lrx.h), out its full-width rows from shadow_row0 on");
    sh = shadow->data_ptr();
  }
  float* rb = nullptr;
  if (row_bounds.has_value()) {
    need(*row_bounds, "row_bounds", at::kFloat, 1);
    TORCH_CHECK(row_bounds->numel() == 2, "encode_packed: row_bounds [2]");
    rb = row_bounds->data_ptr<float>();
  }
  lrx_check(lrx_encode_packed_shard(h->cfg, h->w, ids.data_ptr<int32_t>(), cu_seqlens.data_ptr<int32_t>(), (int32_t)B, (int32_t)T, (int32_t)max_seqlen,
                                    out.data_ptr<float>(), out.stride(0), (int32_t)D, normalize ? 1 : 0, sh, shadow_row0, rb, ws.data_ptr(), need_b,
                                    cur_stream()),
            "encode_packed");
}

// ---- unit kernels -------------------------------------------------------------------------------------------------------------------
at::Tensor rmsnorm(const at::Tensor& x, const at::Tensor& w, double eps) {
  DevGuard guard(x.device());
  need(x, "x", at::kBFloat16, 2);
  need(w, "w", at::kBFloat16, 1);
  TORCH_CHECK(x.is_contiguous() && w.numel() == x.size(1), "rmsnorm: x [rows, H] contiguous, w [H]");
  at::Tensor y = at::empty_like(x);
  lrx_check(lrx_rmsnorm(x.data_ptr(), w.data_ptr(), y.data_ptr(), (int32_t)x.size(0), (int32_t)x.size(1), (float)eps, cur_stream()), "rmsnorm");
  return y;
}

at::Tensor rope_qkv_gemm(const at::Tensor& a, const at::Tensor& wqkv, const c10::optional<at::Tensor>& bias, const at::Tensor& positions,
                         const at::Tensor& cos, const at::Tensor& sin, int64_t num_q_heads, int64_t num_kv_heads, int64_t head_dim) {
  DevGuard guard(a.device());
  need(a, "a", at::kBFloat16, 2);
  need(wqkv, "wqkv", at::kBFloat16, 2);
  need(positions, "positions", at::kInt, 1);
  need(cos, "cos", at::kFloat, 2);
  need(sin, "sin", at::kFloat, 2);
  TORCH_CHECK(a.is_contiguous() && wqkv.is_contiguous() && wqkv.size(1) == a.size(1), "rope_qkv_gemm: a [M,K], wqkv [N,K] contiguous");
  TORCH_CHECK(wqkv.size(0) == (num_q_heads + 2 * num_kv_heads) * head_dim, "rope_qkv_gemm: wqkv rows != (nq + 2 nkv) * d");
  at::Tensor c = at::empty({a.size(0), wqkv.size(0)}, a.options().dtype(at::kHalf));   // fp16, q | k columns in rotary-pair order (include/lrx.h)
  lrx_check(lrx_gemm_qkv_rope(a.data_ptr(), wqkv.data_ptr(), c.data_ptr(), bias.has_value() ? bias->data_ptr() : nullptr, positions.data_ptr<int32_t>(),
                              cos.data_ptr<float>(), sin.data_ptr<float>(), (int32_t)a.size(0), (int32_t)a.size(1), (int32_t)num_q_heads,
                              (int32_t)num_kv_heads, (int32_t)head_dim, cur_stream()),
            "rope_qkv_gemm");
  return c;
}

at::Tensor attn_varlen(const at::Tensor& qkv, const at::Tensor& cu_seqlens, int64_t max_seqlen, int64_t num_q_heads, int64_t num_kv_heads,
                       int64_t head_dim) {
  DevGuard guard(qkv.device());
  need(qkv, "qkv", at::kHalf, 2);
  need(cu_seqlens, "cu_seqlens", at::kInt, 1);
  TORCH_CHECK(qkv.is_contiguous() && qkv.size(1) == (num_q_heads + 2 * num_kv_heads) * head_dim, "attn_varlen: qkv [T, (nq + 2 nkv) * d] contiguous");
  at::Tensor out = at::empty({qkv.size(0), num_q_heads * head_dim}, qkv.options().dtype(at::kBFloat16));
  lrx_check(lrx_attn_varlen_causal(qkv.data_ptr(), cu_seqlens.data_ptr<int32_t>(), (int32_t)cu_seqlens.numel() - 1, (int32_t)qkv.size(0),
                                   (int32_t)max_seqlen, (int32_t)num_q_heads, (int32_t)num_kv_heads, (int32_t)head_dim, out.data_ptr(), 0, cur_stream()),
            "attn_varlen");
  return out;
}

at::Tensor swiglu_gemm(const at::Tensor& a, const at::Tensor& wgu) {
  DevGuard guard(a.device());
  need(a, "a", at::kBFloat16, 2);
  need(wgu, "wgu", at::kBFloat16, 2);
  TORCH_CHECK(a.is_contiguous() && wgu.is_contiguous() && wgu.size(1) == a.size(1) && wgu.size(0) % 2 == 0, "swiglu_gemm: a [M,K], wgu [2I,K] contiguous");
  at::Tensor c = at::empty({a.size(0), wgu.size(0) / 2}, a.options());
  lrx_check(lrx_gemm_bf16_nt(a.data_ptr(), wgu.data_ptr(), c.data_ptr(), nullptr, nullptr, (int32_t)a.size(0), (int32_t)wgu.size(0), (int32_t)a.size(1), 2,
                             cur_stream()),
            "swiglu_gemm");
  return c;
}

// ---- query side -------------------------------------------------------------------------------------------------------------------
at::Tensor embedding_bag_mean(const at::Tensor& table, const at::Tensor& ids, const at::Tensor& offsets, int64_t padding_idx, int64_t out_dim,
                              bool normalize) {
  DevGuard guard(table.device());
  need(table, "table", at::kFloat, 2);
  need(ids, "ids", at::kLong, 1);
  need(offsets, "offsets", at::kLong, 1);
  TORCH_CHECK(table.is_contiguous(), "embedding_bag_mean: table must be contiguous");
  const int64_t D = out_dim > 0 ? out_dim : table.size(1);
  at::Tensor out = at::empty({offsets.numel(), D}, table.options());
  lrx_check(lrx_embedding_bag_mean(table.data_ptr<float>(), (int32_t)table.size(0), (int32_t)table.size(1), ids.data_ptr<int64_t>(), ids.numel(),
                                   offsets.data_ptr<int64_t>(), (int32_t)offsets.numel(), padding_idx, out.data_ptr<float>(), out.stride(0), (int32_t)D,
                                   normalize ? 1 : 0, cur_stream()),
            "embedding_bag_mean");
  return out;
}

// ---- index ----------------------------------------------------------------------------------------------------------------------
std::tuple<at::Tensor, at::Tensor> flat_ip_topk(const at::Tensor& q, const at::Tensor& x, int64_t k, int64_t id_base,
                                                const c10::optional<at::Tensor>& row_bounds) {
  DevGuard guard(q.device());
  need(q, "q", at::kFloat, 2);
  need(x, "x", at::kFloat, 2);
  TORCH_CHECK(q.is_contiguous() && q.size(1) == x.size(1), "flat_ip_topk: q [Q,D] contiguous, x [N,D]");
  at::Tensor d = at::empty({q.size(0), k}, q.options()), i = at::empty({q.size(0), k}, q.options().dtype(at::kLong));
  const size_t wsb = lrx_flat_ip_workspace_bytes(x.size(0), (int32_t)x.size(1), (int32_t)q.size(0), (int32_t)k);
  at::Tensor ws = bytes((int64_t)wsb, q);
  const float* rb = nullptr;
  if (row_bounds.has_value()) {
    need(*row_bounds, "row_bounds", at::kFloat, 1);
    TORCH_CHECK(row_bounds->numel() == 2, "flat_ip_topk: row_bounds [2]");
    rb = row_bounds->data_ptr<float>();
  }
  lrx_check(lrx_flat_ip_search(x.data_ptr<float>(), x.size(0), x.size(0) ? x.stride(0) : x.size(1), (int32_t)x.size(1), rb, q.data_ptr<float>(),
                               (int32_t)q.size(0), (int32_t)k, id_base, d.data_ptr<float>(), i.data_ptr<int64_t>(), ws.data_ptr(), wsb, cur_stream()),
            "flat_ip_topk");
  return {d, i};
}

// x_shadow: the shard's 1-D tiled fp16 shadow (include/lrx.h), block 0 row 0 = x row 0; flags: LRX_SEARCH_FILTER_*
std::tuple<at::Tensor, at::Tensor> flat_ip_topk_bounded(const at::Tensor& q, const at::Tensor& x, const c10::optional<at::Tensor>& x_shadow,
                                                        const at::Tensor& row_bounds, int64_t k, int64_t id_base, int64_t flags) {
  DevGuard guard(q.device());
  need(q, "q", at::kFloat, 2);
  need(x, "x", at::kFloat, 2);
  need(row_bounds, "row_bounds", at::kFloat, 1);
  TORCH_CHECK(q.is_contiguous() && q.size(1) == x.size(1) && row_bounds.numel() == 2, "flat_ip_topk_bounded: q [Q,D] contiguous, x [N,D], row_bounds [2]");
  if (x_shadow.has_value()) {
    TORCH_CHECK(x_shadow->is_cuda() && x_shadow->scalar_type() == at::kHalf && x_shadow->dim() == 1 && x_shadow->is_contiguous() &&
                    x_shadow->numel() >= ((x.size(0) + 127) / 128) * 128 * x.size(1),
                "flat_ip_topk_bounded: x_shadow must be the 1-D tiled fp16 shadow of x (whole 128-row blocks)");
  }
  at::Tensor d = at::empty({q.size(0), k}, q.options()), i = at::empty({q.size(0), k}, q.options().dtype(at::kLong));
  const size_t wsb = lrx_flat_ip_bounded_workspace_bytes(x.size(0), (int32_t)x.size(1), (int32_t)q.size(0), (int32_t)k, (int32_t)flags);
  at::Tensor ws = bytes((int64_t)wsb, q);
  lrx_check(lrx_flat_ip_search_bounded(x.data_ptr<float>(), x.size(0), x.size(0) ? x.stride(0) : x.size(1), (int32_t)x.size(1),
                                       x_shadow.has_value() ? x_shadow->data_ptr() : nullptr, row_bounds.data_ptr<float>(), q.data_ptr<float>(),
                                       (int32_t)q.size(0), (int32_t)k, id_base, d.data_ptr<float>(), i.data_ptr<int64_t>(), ws.data_ptr(), wsb, (int32_t)flags,
                                       cur_stream()),
            "flat_ip_topk_bounded");
  return {d, i};
}

// The same search for one rank of a row-sharded index: also returns the [Q, k] int64 wire words of the exchange (lrx_pack_topk's format,
// row_map applied), written by the search's own last kernel -- all_gather_into_tensor them and hand the result to merge_topk_packed.
std::tuple<at::Tensor, at::Tensor, at::Tensor> flat_ip_topk_bounded_wire(const at::Tensor& q, const at::Tensor& x, const c10::optional<at::Tensor>& x_shadow,
                                                                         const at::Tensor& row_bounds, int64_t k, int64_t id_base,
                                                                         const c10::optional<at::Tensor>& row_map, int64_t flags) {
  DevGuard guard(q.device());
  need(q, "q", at::kFloat, 2);
  need(x, "x", at::kFloat, 2);
  need(row_bounds, "row_bounds", at::kFloat, 1);
  TORCH_CHECK(q.is_contiguous() && q.size(1) == x.size(1) && row_bounds.numel() == 2, "flat_ip_topk_bounded_wire: q [Q,D] contiguous, x [N,D], row_bounds [2]");
  if (x_shadow.has_value()) {
    TORCH_CHECK(x_shadow->is_cuda() && x_shadow->scalar_type() == at::kHalf && x_shadow->dim() == 1 && x_shadow->is_contiguous() &&
                    x_shadow->numel() >= ((x.size(0) + 127) / 128) * 128 * x.size(1),
                "flat_ip_topk_bounded_wire: x_shadow must be the 1-D tiled fp16 shadow of x (whole 128-row blocks)");
  }
  if (row_map.has_value()) {
    need(*row_map, "row_map", at::kLong, 1);
    TORCH_CHECK(row_map->is_contiguous() && row_map->numel() >= x.size(0), "flat_ip_topk_bounded_wire: row_map int64 [>= N] contiguous");
  }
  at::Tensor d = at::empty({q.size(0), k}, q.options()), i = at::empty({q.size(0), k}, q.options().dtype(at::kLong));
  at::Tensor w = at::empty({q.size(0), k}, q.options().dtype(at::kLong));
  const size_t wsb = lrx_flat_ip_bounded_workspace_bytes(x.size(0), (int32_t)x.size(1), (int32_t)q.size(0), (int32_t)k, (int32_t)flags);
  at::Tensor ws = bytes((int64_t)wsb, q);
  lrx_check(lrx_flat_ip_search_bounded_wire(x.data_ptr<float>(), x.size(0), x.size(0) ? x.stride(0) : x.size(1), (int32_t)x.size(1),
                                            x_shadow.has_value() ? x_shadow->data_ptr() : nullptr, row_bounds.data_ptr<float>(), q.data_ptr<float>(),
                                            (int32_t)q.size(0), (int32_t)k, id_base, d.data_ptr<float>(), i.data_ptr<int64_t>(),
                                            row_map.has_value() ? row_map->data_ptr<int64_t>() : nullptr, (uint64_t*)w.data_ptr<int64_t>(), ws.data_ptr(), wsb,
                                            (int32_t)flags, cur_stream()),
            "flat_ip_topk_bounded_wire");
  return {d, i, w};
}

// [R, Q, k] gathered wire words -> the global top-k (lrx_merge_topk_packed)
std::tuple<at::Tensor, at::Tensor> merge_topk_packed(const at::Tensor& words) {
  DevGuard guard(words.device());
  need(words, "words", at::kLong, 3);
  TORCH_CHECK(words.is_contiguous(), "merge_topk_packed: [R,Q,k] contiguous int64 words");
  const int64_t R = words.size(0), Q = words.size(1), k = words.size(2);
  at::Tensor d = at::empty({Q, k}, words.options().dtype(at::kFloat)), i = at::empty({Q, k}, words.options());
  lrx_check(lrx_merge_topk_packed((const uint64_t*)words.data_ptr<int64_t>(), (int32_t)R, (int32_t)Q, (int32_t)k, d.data_ptr<float>(), i.data_ptr<int64_t>(),
                                  cur_stream()),
            "merge_topk_packed");
  return {d, i};
}

// x_shadow: 1-D tiled fp16 shadow (whole 128-row blocks) with row0 = index of x's first row in it, or None (bounds only)
void shard_commit_rows(const at::Tensor& x, const c10::optional<at::Tensor>& x_shadow, at::Tensor row_bounds, int64_t row0) {
  DevGuard guard(x.device());
  need(x, "x", at::kFloat, 2);
  need(row_bounds, "row_bounds", at::kFloat, 1);
  if (x_shadow.has_value()) {
    TORCH_CHECK(x_shadow->is_cuda() && x_shadow->scalar_type() == at::kHalf && x_shadow->is_contiguous() && x_shadow->dim() == 1 &&
                    x_shadow->numel() >= ((row0 + x.size(0) + 127) / 128) * 128 * x.size(1),
                "shard_commit_rows: x_shadow must be the 1-D tiled fp16 shadow, large enough for rows row0 .. row0 + n");
  }
  lrx_check(lrx_shard_commit_rows(x.data_ptr<float>(), x.stride(0), x.size(0), (int32_t)x.size(1), x_shadow.has_value() ? x_shadow->data_ptr() : nullptr, row0,
                                  row_bounds.data_ptr<float>(), cur_stream()),
            "shard_commit_rows");
}

std::tuple<at::Tensor, at::Tensor> merge_topk(const at::Tensor& d_parts, const at::Tensor& i_parts) {
  DevGuard guard(d_parts.device());
  need(d_parts, "d_parts", at::kFloat, 3);
  need(i_parts, "i_parts", at::kLong, 3);
  TORCH_CHECK(d_parts.is_contiguous() && i_parts.is_contiguous() && d_parts.sizes() == i_parts.sizes(), "merge_topk: [R,Q,k] contiguous pairs");
  const int64_t R = d_parts.size(0), Q = d_parts.size(1), k = d_parts.size(2);
  at::Tensor d = at::empty({Q, k}, d_parts.options()), i = at::empty({Q, k}, i_parts.options());
  lrx_check(lrx_merge_topk(d_parts.data_ptr<float>(), i_parts.data_ptr<int64_t>(), (int32_t)R, (int32_t)Q, (int32_t)k, d.data_ptr<float>(),
                           i.data_ptr<int64_t>(), cur_stream()),
            "merge_topk");
  return {d, i};
}

}  // namespace

TORCH_LIBRARY(lrx, m) {
  m.def("encode_packed(Tensor ids, Tensor cu_seqlens, int max_seqlen, int weights, Tensor(a!) out, int mrl_dim=0, bool normalize=True, "
        "Tensor(b!)? shadow=None, int shadow_row0=0, Tensor(c!)? row_bounds=None) -> ()");
  m.def("rmsnorm(Tensor x, Tensor w, float eps) -> Tensor");
  m.def("rope_qkv_gemm(Tensor a, Tensor wqkv, Tensor? bias, Tensor positions, Tensor cos, Tensor sin, int num_q_heads, int num_kv_heads, int head_dim) -> Tensor");
  m.def("attn_varlen(Tensor qkv, Tensor cu_seqlens, int max_seqlen, int num_q_heads, int num_kv_heads, int head_dim) -> Tensor");
  m.def("swiglu_gemm(Tensor a, Tensor wgu) -> Tensor");
  m.def("embedding_bag_mean(Tensor table, Tensor ids, Tensor offsets, int padding_idx=-1, int out_dim=0, bool normalize=True) -> Tensor");
  m.def("flat_ip_topk(Tensor q, Tensor x, int k, int id_base=0, Tensor? row_bounds=None) -> (Tensor, Tensor)");
  m.def("flat_ip_topk_bounded(Tensor q, Tensor x, Tensor? x_shadow, Tensor row_bounds, int k, int id_base=0, int flags=0) -> (Tensor, Tensor)");
  m.def("flat_ip_topk_bounded_wire(Tensor q, Tensor x, Tensor? x_shadow, Tensor row_bounds, int k, int id_base=0, Tensor? row_map=None, int flags=0) -> (Tensor, Tensor, Tensor)");
  m.def("merge_topk_packed(Tensor words) -> (Tensor, Tensor)");
  m.def("shard_commit_rows(Tensor x, Tensor(a!)? x_shadow, Tensor(b!) row_bounds, int row0=0) -> ()");
  m.def("merge_topk(Tensor d_parts, Tensor i_parts) -> (Tensor, Tensor)");
}

TORCH_LIBRARY_IMPL(lrx, CUDA, m) {   // (the ROCm build of PyTorch dispatches HIP tensors under the CUDA key)
  m.impl("encode_packed", &encode_packed);
  m.impl("rmsnorm", &rmsnorm);
  m.impl("rope_qkv_gemm", &rope_qkv_gemm);
  m.impl("attn_varlen", &attn_varlen);
  m.impl("swiglu_gemm", &swiglu_gemm);
  m.impl("embedding_bag_mean", &embedding_bag_mean);
  m.impl("flat_ip_topk", &flat_ip_topk);
  m.impl("flat_ip_topk_bounded", &flat_ip_topk_bounded);
  m.impl("flat_ip_topk_bounded_wire", &flat_ip_topk_bounded_wire);
  m.impl("merge_topk_packed", &merge_topk_packed);
  m.impl("shard_commit_rows", &shard_commit_rows);
  m.impl("merge_topk", &merge_topk);
}
