// C-ABI glue: error state, the fused encoder forward (launch sequence over the kernels in this directory), profiling.
#include "lrx_common.h"
#include <stdarg.h>
#include <atomic>
#include <mutex>
#include <vector>

static thread_local char g_err[512] = "";
void lrx_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* lrx_last_error(void) { return g_err; }
extern "C" int lrx_abi_version(void) { return LRX_ABI_VERSION; }

// ---------------------------------------------------------------------------------------------------------------
// profiling (HIP events on the caller's stream)
// ---------------------------------------------------------------------------------------------------------------
// Process-wide, guarded by g_prof_mu: the per-batch operator may be called from several host threads (the reference calls it from
// TensorPipe RPC server threads).  The unprofiled fast path reads one atomic flag and takes no lock.
static std::mutex g_prof_mu;
static std::atomic<bool> g_prof{false};
static uint32_t g_prof_mask = 0xffffffffu;   // classes that get events
static float g_prof_ms[LRX_PROF_CLASSES];
static double g_prof_flops[LRX_PROF_CLASSES];
static int32_t g_prof_launches[LRX_PROF_CLASSES];
struct ProfRec { hipEvent_t a, b; int cls; };
static std::vector<hipEvent_t> g_ev_pool;
static std::vector<ProfRec> g_recs;
static size_t g_ev_used = 0;

static hipEvent_t prof_event() {
  if (g_ev_used == g_ev_pool.size()) {
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    g_ev_pool.push_back(e);
  }
  return g_ev_pool[g_ev_used++];
}
struct ProfScope {
  hipStream_t s; int cls; hipEvent_t a = nullptr, b = nullptr;
  ProfScope(hipStream_t s_, int cls_, double flops) : s(s_), cls(cls_) {
    if (!g_prof.load(std::memory_order_relaxed)) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (!g_prof || !((g_prof_mask >> cls_) & 1u)) return;
    a = prof_event(); b = prof_event();
    if (a) (void)hipEventRecord(a, s);
    g_prof_flops[cls] += flops;
    g_prof_launches[cls] += 1;
  }
  ~ProfScope() {
    if (!a || !b) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    (void)hipEventRecord(b, s);
    g_recs.push_back({a, b, cls});
  }
};
// Events are only recorded while an lrx_encode_* call runs; nothing synchronises the stream until lrx_get_profile reads them, so
// profiled calls keep the device queue full (a sync per call cost ~2 % of the step in the benchmark's timed region).
static void prof_reset() {
  g_ev_used = 0;
  g_recs.clear();
  for (int i = 0; i < LRX_PROF_CLASSES; ++i) { g_prof_ms[i] = 0.f; g_prof_flops[i] = 0.0; g_prof_launches[i] = 0; }
}
static void prof_begin() {}
static int prof_end(hipStream_t) { return LRX_OK; }
extern "C" void lrx_set_profiling(int32_t enabled) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof = enabled != 0;
  g_prof_mask = enabled > 1 ? ((uint32_t)enabled >> 1) : 0xffffffffu;   // 1 = every class, otherwise bit (c + 1) selects class c
  prof_reset();
}
extern "C" int lrx_get_profile(float* ms, double* flops, int32_t* launches) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (auto& r : g_recs) {
    float t = 0.f;
    if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&t, r.a, r.b) == hipSuccess) g_prof_ms[r.cls] += t;
  }
  g_recs.clear();
  g_ev_used = 0;
  for (int i = 0; i < LRX_PROF_CLASSES; ++i) {
    if (ms) ms[i] = g_prof_ms[i];
    if (flops) flops[i] = g_prof_flops[i];
    if (launches) launches[i] = g_prof_launches[i];
  }
  for (int i = 0; i < LRX_PROF_CLASSES; ++i) { g_prof_ms[i] = 0.f; g_prof_flops[i] = 0.0; g_prof_launches[i] = 0; }
  return LRX_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// encoder forward
// ---------------------------------------------------------------------------------------------------------------
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct EncWs {
  char *x, *h, *qkv, *act;    // x: the bf16 residual stream, or (precise_stream) the bf16 A operand bf16(x32 * gamma_next) of the next projection
  char *xr, *ar, *hr, *actr;  // compact [n_seqs, .] buffers of the pooled tail of the final layer
  float *x32, *xr32;          // precise_stream: the fp32 residual stream (and its pooled-tail rows)
  float* aux32;               // [n_seqs, H] column sums of the other hidden state (LRX_POOL_AVG_FIRST_LAST / AVG_TOP2)
  int32_t* pos;
  float *rsA, *rsB, *ssp;     // folded RMSNorm: row scales for the two norms of a layer, per-n-tile sum-of-squares partials
  int32_t* posr;              // final layer: positions / row scales of the last-token rows (the q projection runs on those rows only)
  float* rsr;
  char *attn_items, *attn_items_tail;   // work lists of the attention launches (all q tiles; each sequence's last q tile), built once per batch
  size_t attn_items_bytes, attn_items_tail_bytes;
  size_t total;
};
static EncWs carve(const lrx_encoder_config* c, int64_t T, int64_t B, char* base) {
  const int64_t H = c->hidden_size, QD = (int64_t)c->num_q_heads * c->head_dim;
  const int64_t QKV = (int64_t)(c->num_q_heads + 2 * c->num_kv_heads) * c->head_dim, I = c->intermediate_size;
  const int64_t HM = H > QD ? H : QD;
  EncWs w;
  size_t off = 0;
  w.x = base + off; off += align_up((size_t)T * H * 2, 1024);
  w.h = base + off; off += align_up((size_t)T * HM * 2, 1024);
  w.qkv = base + off; off += align_up((size_t)T * QKV * 2, 1024);
  w.act = base + off; off += align_up((size_t)T * I * 2, 1024);
  w.pos = (int32_t*)(base + off); off += align_up((size_t)T * 4, 1024);
  w.rsA = (float*)(base + off); off += align_up((size_t)T * 4, 1024);
  w.rsB = (float*)(base + off); off += align_up((size_t)T * 4, 1024);
  w.ssp = (float*)(base + off); off += align_up((size_t)((H + 255) / 256) * T * 4, 1024);
  w.xr = base + off; off += align_up((size_t)B * H * 2, 1024);
  w.ar = base + off; off += align_up((size_t)B * HM * 2, 1024);
  w.hr = base + off; off += align_up((size_t)B * H * 2, 1024);
  w.actr = base + off; off += align_up((size_t)B * I * 2, 1024);
  w.posr = (int32_t*)(base + off); off += align_up((size_t)B * 4 + 32, 1024);
  w.rsr = (float*)(base + off); off += align_up((size_t)B * 4 + 32, 1024);
  w.aux32 = (float*)(base + off); off += align_up((size_t)B * H * 4, 1024);
  {
    // sized for the longest sequence a call may name (check_call: max_seqlen <= max_positions; the list size is non-decreasing in
    // max_seqlen and bounded by (T / 64 + B) x kv heads items of 16 bytes: ~300 KiB for 256 x 512 tokens and 8 kv heads)
    const int64_t smax = c->max_positions;
    w.attn_items_bytes = lrx_attn_items_bytes((int32_t)B, (int32_t)T, (int32_t)smax, c->num_q_heads, c->num_kv_heads, c->head_dim, 0);
    w.attn_items_tail_bytes = lrx_attn_items_bytes((int32_t)B, (int32_t)T, (int32_t)smax, c->num_q_heads, c->num_kv_heads, c->head_dim, 1);
    w.attn_items = base + off; off += align_up(w.attn_items_bytes + 16, 1024);
    w.attn_items_tail = base + off; off += align_up(w.attn_items_tail_bytes + 16, 1024);
  }
  w.x32 = w.xr32 = nullptr;
  if (c->precise_stream) {
    w.x32 = (float*)(base + off); off += align_up((size_t)T * H * 4, 1024);
    w.xr32 = (float*)(base + off); off += align_up((size_t)B * H * 4, 1024);
  }
  w.total = off;
  return w;
}

extern "C" size_t lrx_encode_workspace_bytes(const lrx_encoder_config* cfg, int32_t total_tokens, int32_t n_seqs) {
  if (!cfg) return 0;
  return carve(cfg, total_tokens > 0 ? total_tokens : 1, n_seqs > 0 ? n_seqs : 1, nullptr).total;
}

static int check_cfg(const lrx_encoder_config* c) {
  LRX_CHECK_ARG(c != nullptr, "encode: null config");
  LRX_CHECK_ARG(c->hidden_size > 0 && c->hidden_size % 64 == 0, "encode: hidden_size=%d must be a multiple of 64", c->hidden_size);
  LRX_CHECK_ARG(c->intermediate_size > 0 && c->intermediate_size % 64 == 0, "encode: intermediate_size=%d must be a multiple of 64",
                c->intermediate_size);
  LRX_CHECK_ARG(c->head_dim == 64 || c->head_dim == 128, "encode: head_dim=%d unsupported", c->head_dim);
  LRX_CHECK_ARG(c->num_kv_heads > 0 && c->num_q_heads % c->num_kv_heads == 0, "encode: bad head counts %d/%d", c->num_q_heads, c->num_kv_heads);
  LRX_CHECK_ARG(c->num_layers > 0 && c->vocab_size > 0 && c->max_positions > 0, "encode: bad config");
  LRX_CHECK_ARG(c->precise_stream >= 0 && c->precise_stream <= 3, "encode: precise_stream=%d (0: bf16 stream, 1: fp32 stream, 2: + fp16 GEMM operands, 3: + fp16 QKV operands only)", c->precise_stream);
  LRX_CHECK_ARG(!(c->precise_stream && c->norm_folded), "encode: precise_stream keeps the norm weight on the activation operand -- pass the unfolded weights (norm_folded = 0)");
  return LRX_OK;
}

// What differs between the callers of the layer loop: the attention step (varlen causal over the batch; the same restricted to each sequence's
// last q tile; suffix queries over a captured prefix) and an optional hook between the QKV projection and the attention (the shared-prefix
// pass copies K|V out there; returning 1 stops the loop: nothing after the last layer's K/V is needed from the prefix).
struct LayerHooks {
  // attention of layer l over ws.qkv -> ws.h; last_tile: only each sequence's last 64-row q tile is needed
  int (*attn)(void* ctx, int l, bool last_tile, hipStream_t s);
  int (*after_qkv)(void* ctx, int l, hipStream_t s);   // may be NULL
  void* ctx;
  double attn_flops;            // per layer, for the profile
  int (*before_layer)(void* ctx, int l, hipStream_t s) = nullptr;   // may be NULL: sees the stream as layer l is about to read it (hidden_states[l])
};

// Embedding + all layers over T packed tokens.  Leaves the residual stream BEFORE the final norm in ws.x (bf16) or ws.x32 (precise_stream) --
// or, with tail_cu != NULL, only the n_tail last-token rows of it, compacted, in ws.xr / ws.xr32: after the final layer's attention nothing
// but those rows reaches the pooled output, so its O-projection and MLP run on [n_tail, H] instead of [T, H] (bit-identical rows).
//   norm_folded:    RMSNorm weights pre-multiplied into wqkv / wgu; the row statistic travels as a [T] fp32 scale that the consuming GEMM
//                   applies to its fp32 accumulator, the residual GEMMs emit the sum of squares of the rows they produce.
//   precise_stream: the residual stream is fp32 (x32), every residual GEMM also writes the next projection's bf16 A operand
//                   bf16(x32 * gamma_next) into ws.x; weights exact, row scale on the accumulator as above.  tools/exp/rounding_budget.py:
//                   at 32 layers the bf16 stream + folded weights cost ~1.1e-3 of the 1e-3 cosine budget against the fp32 model.
//   neither:        HF's bf16 arithmetic (normalised activations materialised by lrx_rmsnorm).
static int run_layers(const lrx_encoder_config* c, const lrx_encoder_weights* w, const int32_t* ids, int T, EncWs& ws, const LayerHooks& hk,
                      const int32_t* tail_cu, int n_tail, hipStream_t s) {
  const int H = c->hidden_size, d = c->head_dim, nq = c->num_q_heads, nkv = c->num_kv_heads, I = c->intermediate_size;
  const int QKV = (nq + 2 * nkv) * d, QD = nq * d;
  const bool fold = c->norm_folded != 0, precise = c->precise_stream != 0;
  // fp16 GEMM operands: precise_stream = 2 -- every projection (activations, attention / SwiGLU outputs, the four weight matrices); 3 -- the QKV
  // projection only (its A operand fp16(x * gamma) and wqkv: 70 % of the pipeline's distance to the fp32 model for < 1 % of a step)
  const int f16 = c->precise_stream == 2 ? 1 : 0, f16q = c->precise_stream >= 2 ? 1 : 0;
  const bool scaled = fold || precise;                 // the consumer GEMM multiplies its accumulator rows by rsqrt(mean(x^2) + eps)
  const int NP = (H + 255) / 256;                      // n-tiles of a residual GEMM (N = H)
  int rc;
  if (precise) { ProfScope p(s, 6, 0); if ((rc = lrx_embed_stream32_ex(w->embed, ids, T, H, c->vocab_size, w->layers[0].ln1, ws.x32, ws.x, ws.rsA, c->rms_eps, f16q, s))) return rc; }
  else {
    { ProfScope p(s, 6, 0); if ((rc = lrx_embedding_gather(w->embed, ids, T, H, c->vocab_size, ws.x, s))) return rc; }
    if (fold) { ProfScope p(s, 4, 0); if ((rc = lrx_row_rscale(ws.x, T, H, c->rms_eps, ws.rsA, s))) return rc; }
  }
  // residual GEMM: stream += A . W^T over `rows` rows (+ what the next consumer needs: the bf16 operand in precise mode, the sum of squares)
  // (next_f16: the operand written for the next projection is fp16 -- the gate-up's under mode 2, the QKV's under modes 2 and 3)
  auto resid = [&](const void* A, const void* W, char* xb, float* x32, const void* gamma_next, bool want_next, int rows, int N, int K, int next_f16) -> int {
    ProfScope p(s, 1, 2.0 * rows * (double)N * K);
    if (precise) return lrx_gemm_nt_resid32_ex(A, W, x32, want_next ? xb : nullptr, want_next ? gamma_next : nullptr, rows, N, K, want_next ? ws.ssp : nullptr, f16, next_f16, s);
    return lrx_gemm_bf16_nt_fused(A, W, xb, nullptr, xb, rows, N, K, 1, nullptr, (fold && want_next) ? ws.ssp : nullptr, s);
  };
  for (int l = 0; l < c->num_layers; ++l) {
    const lrx_layer_weights& L = w->layers[l];
    const bool last = l == c->num_layers - 1;
    if (hk.before_layer && (rc = hk.before_layer(hk.ctx, l, s))) return rc;
    if (!scaled) { ProfScope p(s, 4, 0); if ((rc = lrx_rmsnorm(ws.x, L.ln1, ws.h, T, H, c->rms_eps, s))) return rc; }
    const void* Aqkv = scaled ? ws.x : ws.h;
    const void* bq = c->qkv_bias ? L.bqkv : nullptr;
    if (tail_cu != nullptr && last && hk.after_qkv == nullptr && T > 2 * n_tail && c->num_layers > 1) {
      // FINAL layer of a pooled encode: k|v of every token, q of the n_tail last-token rows only (nothing else of q reaches the output: the
      // attention below is restricted to each sequence's last q tile and only its last row is gathered).  The last rows of the operand, of
      // the row scale and of the positions are compacted (ws.hr, ws.rsr, ws.posr: free here), projected, and the q rows scattered back
      // (ws.ar is free until the attention output is gathered).  Rows of the last q tile other than the last keep stale q values: finite
      // (the previous layer's -- hence num_layers > 1: a one-layer model would leave uninitialised memory there), per-row independent, never read out.
      const int B = n_tail;
      { ProfScope p(s, 0, 2.0 * T * (double)(2 * nkv * d) * H + 2.0 * B * (double)QD * H);
        if ((rc = lrx_gemm_qkv_rope_slice_ex(Aqkv, L.wqkv, ws.qkv, bq, ws.pos, w->rope_cos, w->rope_sin, T, H, nq, nkv, d, scaled ? ws.rsA : nullptr, nq, 2 * nkv, f16q, s))) return rc;
        if ((rc = lrx_gather_last_rows(Aqkv, tail_cu, B, H, ws.hr, s))) return rc;
        if ((rc = lrx_gather_rows_u32(ws.pos, tail_cu, B, ws.posr, s))) return rc;
        if (scaled && (rc = lrx_gather_rows_u32(ws.rsA, tail_cu, B, ws.rsr, s))) return rc;
        // (num_kv_heads = 0 in this call: C = ws.ar is [B, QD] with row stride QD -- the q heads are the whole row; ws.ar holds B x max(H, QD)
        // elements, carve())
        if ((rc = lrx_gemm_qkv_rope_slice_ex(ws.hr, L.wqkv, ws.ar, bq, ws.posr, w->rope_cos, w->rope_sin, B, H, nq, 0, d, scaled ? ws.rsr : nullptr, 0, nq, f16q, s))) return rc;
        if ((rc = lrx_scatter_last_rows(ws.ar, tail_cu, B, QD, ws.qkv, QKV, s))) return rc; }
    } else
    { ProfScope p(s, 0, 2.0 * T * (double)QKV * H);   // QKV projection with (row scale,) bias + RoPE fused into the epilogue
      if ((rc = lrx_gemm_qkv_rope_slice_ex(Aqkv, L.wqkv, ws.qkv, bq, ws.pos, w->rope_cos, w->rope_sin, T, H, nq, nkv, d,
                                           scaled ? ws.rsA : nullptr, 0, nq + 2 * nkv, f16q, s))) return rc; }
    if (hk.after_qkv) { rc = hk.after_qkv(hk.ctx, l, s); if (rc == 1) break; if (rc) return rc; }
    if (tail_cu != nullptr && last) {
      const int B = n_tail;
      const double S = (double)T / (double)(B > 0 ? B : 1);
      { ProfScope p(s, 3, hk.attn_flops * 64.0 / (S > 64.0 ? S : 64.0)); if ((rc = hk.attn(hk.ctx, l, true, s))) return rc; }
      { ProfScope p(s, 6, 0); if ((rc = lrx_gather_last_rows(ws.h, tail_cu, B, QD, ws.ar, s))) return rc; }
      { ProfScope p(s, 6, 0);
        if ((rc = precise ? lrx_gather_last_rows(ws.x32, tail_cu, B, 2 * H, ws.xr32, s) : lrx_gather_last_rows(ws.x, tail_cu, B, H, ws.xr, s))) return rc; }
      if ((rc = resid(ws.ar, L.wo, ws.xr, ws.xr32, L.ln2, true, B, H, QD, f16))) return rc;
      if (scaled) { ProfScope p(s, 4, 0); if ((rc = lrx_finalize_rscale(ws.ssp, NP, B, H, c->rms_eps, ws.rsB, s))) return rc; }
      else { ProfScope p(s, 4, 0); if ((rc = lrx_rmsnorm(ws.xr, L.ln2, ws.hr, B, H, c->rms_eps, s))) return rc; }
      { ProfScope p(s, 2, 2.0 * B * (double)(2 * I) * H);
        if ((rc = lrx_gemm_nt_fused_ex(scaled ? ws.xr : ws.hr, L.wgu, ws.actr, nullptr, nullptr, B, 2 * I, H, 2, scaled ? ws.rsB : nullptr, nullptr, f16, s))) return rc; }
      if ((rc = resid(ws.actr, L.wdown, ws.xr, ws.xr32, nullptr, false, B, H, I, f16q))) return rc;
      break;
    }
    { ProfScope p(s, 3, hk.attn_flops); if ((rc = hk.attn(hk.ctx, l, false, s))) return rc; }
    if ((rc = resid(ws.h, L.wo, ws.x, ws.x32, L.ln2, true, T, H, QD, f16))) return rc;
    if (scaled) { ProfScope p(s, 4, 0); if ((rc = lrx_finalize_rscale(ws.ssp, NP, T, H, c->rms_eps, ws.rsB, s))) return rc; }
    else { ProfScope p(s, 4, 0); if ((rc = lrx_rmsnorm(ws.x, L.ln2, ws.h, T, H, c->rms_eps, s))) return rc; }
    { ProfScope p(s, 2, 2.0 * T * (double)(2 * I) * H);
      if ((rc = lrx_gemm_nt_fused_ex(scaled ? ws.x : ws.h, L.wgu, ws.act, nullptr, nullptr, T, 2 * I, H, 2, scaled ? ws.rsB : nullptr, nullptr, f16, s))) return rc; }
    if ((rc = resid(ws.act, L.wdown, ws.x, ws.x32, last ? nullptr : w->layers[l + 1].ln1, !last, T, H, I, f16q))) return rc;
    if (scaled && !last) { ProfScope p(s, 4, 0); if ((rc = lrx_finalize_rscale(ws.ssp, NP, T, H, c->rms_eps, ws.rsA, s))) return rc; }
  }
  return LRX_OK;
}

// the batch callers: varlen causal attention over (cu, n_seqs)
struct BatchAttn { const lrx_encoder_config* c; EncWs* ws; const int32_t* cu; int n_seqs, T, max_seqlen; bool sum_before_last; };
// LRX_POOL_AVG_TOP2: hidden_states[-2] is the stream as it enters the final layer -- its per-sequence column sums go to ws.aux32
static int batch_before_layer(void* ctx, int l, hipStream_t s) {
  const BatchAttn& a = *(const BatchAttn*)ctx;
  if (!a.sum_before_last || l != a.c->num_layers - 1) return LRX_OK;
  ProfScope p(s, 6, 0);
  const bool pr = a.c->precise_stream != 0;
  return lrx_pool_sum_rows(pr ? (const void*)a.ws->x32 : (const void*)a.ws->x, pr ? 1 : 0, nullptr, 0, a.cu, a.n_seqs, a.c->hidden_size, a.ws->aux32, s);
}
static int batch_attn(void* ctx, int, bool last_tile, hipStream_t s) {
  const BatchAttn& a = *(const BatchAttn*)ctx;
  return lrx_attn_varlen_causal_items_ex(a.ws->qkv, a.cu, last_tile ? a.ws->attn_items_tail : a.ws->attn_items,
                                         last_tile ? a.ws->attn_items_tail_bytes : a.ws->attn_items_bytes, a.n_seqs, a.T, a.max_seqlen, a.c->num_q_heads,
                                         a.c->num_kv_heads, a.c->head_dim, a.ws->h, last_tile ? 1 : 0, a.c->precise_stream == 2, s);
}
static int forward_layers(const lrx_encoder_config* c, const lrx_encoder_weights* w, const int32_t* ids, const int32_t* cu, int n_seqs,
                          int T, int max_seqlen, EncWs& ws, bool pooled_tail, hipStream_t s, bool sum_before_last = false) {
  int rc;
  { ProfScope p(s, 6, 0); if ((rc = lrx_build_positions(cu, n_seqs, T, ws.pos, s))) return rc; }
  // the attention work lists: one per batch, read by every layer's launch (the last layer of a pooled encode has its own: last q tiles only)
  { ProfScope p(s, 6, 0);
    if ((rc = lrx_attn_build_items(cu, n_seqs, T, max_seqlen, c->num_q_heads, c->num_kv_heads, c->head_dim, 0, ws.attn_items, ws.attn_items_bytes, s))) return rc;
    if (pooled_tail && (rc = lrx_attn_build_items(cu, n_seqs, T, max_seqlen, c->num_q_heads, c->num_kv_heads, c->head_dim, 1, ws.attn_items_tail,
                                                   ws.attn_items_tail_bytes, s))) return rc; }
  // causal attention flops: sum over sequences is not known on the host without a sync; use the dense upper bound for
  // equal-length batches: n_seqs * S*(S+1)/2 with S = T / n_seqs (exact when all sequences have the same length)
  const double S = (double)T / (double)(n_seqs > 0 ? n_seqs : 1);
  BatchAttn ba = {c, &ws, cu, n_seqs, T, max_seqlen, sum_before_last};
  LayerHooks hk = {batch_attn, nullptr, &ba, 2.0 * 2.0 * c->head_dim * c->num_q_heads * (double)n_seqs * (S * (S + 1.0) / 2.0)};
  if (sum_before_last) hk.before_layer = batch_before_layer;
  return run_layers(c, w, ids, T, ws, hk, pooled_tail ? cu : nullptr, n_seqs, s);
}

static int check_call(const lrx_encoder_config* cfg, const lrx_encoder_weights* w, const int32_t* ids, const int32_t* cu, int n_seqs, int T,
                      int max_seqlen, void* workspace, size_t workspace_bytes) {
  int rc = check_cfg(cfg);
  if (rc) return rc;
  LRX_CHECK_ARG(w && w->embed && w->final_norm && w->rope_cos && w->rope_sin && w->layers, "encode: null weights");
  LRX_CHECK_ARG(ids && cu && n_seqs > 0 && T > 0, "encode: empty batch (n_seqs=%d, tokens=%d)", n_seqs, T);
  LRX_CHECK_ARG(max_seqlen > 0 && max_seqlen <= cfg->max_positions, "encode: max_seqlen=%d exceeds RoPE table (%d)", max_seqlen,
                cfg->max_positions);
  LRX_CHECK_ARG(workspace != nullptr, "encode: null workspace");
  size_t need = lrx_encode_workspace_bytes(cfg, T, n_seqs);
  if (workspace_bytes < need) {
    lrx_set_error("encode: workspace %zu B < required %zu B", workspace_bytes, need);
    return LRX_ERR_WORKSPACE;
  }
  return LRX_OK;
}

// final RMSNorm of all T rows of the stream -> bf16 (last_hidden_state)
static int final_norm_rows(const lrx_encoder_config* c, const lrx_encoder_weights* w, EncWs& ws, int T, void* out, hipStream_t s) {
  ProfScope p(s, 4, 0);
  if (c->precise_stream) return lrx_rmsnorm_f32(ws.x32, w->final_norm, out, T, c->hidden_size, c->rms_eps, s);
  return lrx_rmsnorm(ws.x, w->final_norm, out, T, c->hidden_size, c->rms_eps, s);
}

extern "C" int lrx_encode_packed_pooled(const lrx_encoder_config* cfg, const lrx_encoder_weights* w, const int32_t* ids, const int32_t* cu_seqlens,
                                        int32_t n_seqs, int32_t total_tokens, int32_t max_seqlen, int32_t pooling, float* out, int64_t out_row_stride,
                                        int32_t out_dim, int32_t normalize, void* shadow_out, int64_t shadow_row0, float* row_bounds, void* workspace,
                                        size_t workspace_bytes, void* stream) {
  int rc = check_call(cfg, w, ids, cu_seqlens, n_seqs, total_tokens, max_seqlen, workspace, workspace_bytes);
  if (rc) return rc;
  LRX_CHECK_ARG(out && out_dim > 0 && out_dim <= cfg->hidden_size && out_row_stride >= out_dim, "encode: bad output spec (dim=%d stride=%lld)",
                out_dim, (long long)out_row_stride);
  LRX_CHECK_ARG(pooling >= LRX_POOL_LASTTOKEN && pooling <= LRX_POOL_AVG_TOP2, "encode: pooling=%d (LRX_POOL_*)", pooling);
  hipStream_t s = (hipStream_t)stream;
  EncWs ws = carve(cfg, total_tokens, n_seqs, (char*)workspace);
  prof_begin();
  const bool pr = cfg->precise_stream != 0;
  if (pooling == LRX_POOL_LASTTOKEN) {
    // the released models' strategy: after the final layer's attention only the last-token rows are computed (run_layers' pooled tail)
    if ((rc = forward_layers(cfg, w, ids, cu_seqlens, n_seqs, total_tokens, max_seqlen, ws, true, s))) return rc;
    ProfScope p(s, 6, 0);  // ws.xr / ws.xr32 holds the compacted last-token rows -> cu_seqlens = NULL
    if ((rc = lrx_pool_norm_mode(pr ? (const void*)ws.xr32 : (const void*)ws.xr, w->final_norm, nullptr, n_seqs, cfg->hidden_size, cfg->rms_eps,
                                 LRX_POOL_LASTTOKEN, out, out_row_stride, out_dim, normalize, shadow_out, shadow_row0, row_bounds, pr ? 1 : 0, s)))
      return rc;
  } else {
    // any other strategy of finetune/dense_pooling.py: every layer over every token, then the final norm + pooling over the stream's rows
    // (the two-layer strategies pool a second hidden state too: hidden_states[0] = the embedding rows, summed from the table; hidden_states[-2] =
    // the stream as it enters the final layer, summed by the layer loop's hook)
    if ((rc = forward_layers(cfg, w, ids, cu_seqlens, n_seqs, total_tokens, max_seqlen, ws, false, s, pooling == LRX_POOL_AVG_TOP2))) return rc;
    ProfScope p(s, 6, 0);
    if (pooling == LRX_POOL_AVG_FIRST_LAST &&
        (rc = lrx_pool_sum_rows(w->embed, 2, ids, cfg->vocab_size, cu_seqlens, n_seqs, cfg->hidden_size, ws.aux32, s))) return rc;
    if ((rc = lrx_pool_norm_aux(pr ? (const void*)ws.x32 : (const void*)ws.x, w->final_norm, cu_seqlens, n_seqs, cfg->hidden_size, cfg->rms_eps,
                                pooling, pooling >= LRX_POOL_AVG_FIRST_LAST ? ws.aux32 : nullptr, out, out_row_stride, out_dim, normalize, shadow_out,
                                shadow_row0, row_bounds, pr ? 1 : 0, s)))
      return rc;
  }
  return prof_end(s);
}

extern "C" int lrx_encode_packed_shard(const lrx_encoder_config* cfg, const lrx_encoder_weights* w, const int32_t* ids, const int32_t* cu_seqlens,
                                       int32_t n_seqs, int32_t total_tokens, int32_t max_seqlen, float* out, int64_t out_row_stride,
                                       int32_t out_dim, int32_t normalize, void* shadow_out, int64_t shadow_row0, float* row_bounds, void* workspace,
                                       size_t workspace_bytes, void* stream) {
  return lrx_encode_packed_pooled(cfg, w, ids, cu_seqlens, n_seqs, total_tokens, max_seqlen, LRX_POOL_LASTTOKEN, out, out_row_stride, out_dim, normalize,
                                  shadow_out, shadow_row0, row_bounds, workspace, workspace_bytes, stream);
}

extern "C" int lrx_encode_packed(const lrx_encoder_config* cfg, const lrx_encoder_weights* w, const int32_t* ids, const int32_t* cu_seqlens,
                                 int32_t n_seqs, int32_t total_tokens, int32_t max_seqlen, float* out, int64_t out_row_stride,
                                 int32_t out_dim, int32_t normalize, void* workspace, size_t workspace_bytes, void* stream) {
  return lrx_encode_packed_shard(cfg, w, ids, cu_seqlens, n_seqs, total_tokens, max_seqlen, out, out_row_stride, out_dim, normalize, nullptr, 0,
                                 nullptr, workspace, workspace_bytes, stream);
}

extern "C" int lrx_encode_hidden(const lrx_encoder_config* cfg, const lrx_encoder_weights* w, const int32_t* ids, const int32_t* cu_seqlens,
                                 int32_t n_seqs, int32_t total_tokens, int32_t max_seqlen, void* hidden_out_bf16, void* workspace,
                                 size_t workspace_bytes, void* stream) {
  int rc = check_call(cfg, w, ids, cu_seqlens, n_seqs, total_tokens, max_seqlen, workspace, workspace_bytes);
  if (rc) return rc;
  LRX_CHECK_ARG(hidden_out_bf16 != nullptr, "encode_hidden: null output");
  hipStream_t s = (hipStream_t)stream;
  EncWs ws = carve(cfg, total_tokens, n_seqs, (char*)workspace);
  prof_begin();
  if ((rc = forward_layers(cfg, w, ids, cu_seqlens, n_seqs, total_tokens, max_seqlen, ws, false, s))) return rc;
  if ((rc = final_norm_rows(cfg, w, ws, total_tokens, hidden_out_bf16, s))) return rc;
  return prof_end(s);
}

// Dense + sparse document vectors in one pass (HybridModel.encode_passage with encode_sparse, modeling_hybrid.py:248-323):
// all layers on all tokens, dense = pooled last token (as lrx_encode_packed), sparse = LM-head max aggregation over the
// final-norm hidden states of the tokens tok_mask selects, then relu / log1p / top-k.
extern "C" int lrx_encode_packed_sparse(const lrx_encoder_config* cfg, const lrx_encoder_weights* w, const void* lm_head, const void* lm_head_bias,
                                        const int32_t* ids, const int32_t* cu_seqlens, const uint8_t* tok_mask, int32_t n_seqs, int32_t total_tokens,
                                        int32_t max_seqlen, float* dense_out, int64_t dense_row_stride, int32_t dense_dim, int32_t normalize,
                                        float* sparse_out, int64_t sparse_row_stride, int32_t relu, int32_t log1p, int32_t round_bf16, int32_t top_k,
                                        int32_t min_tokens_to_keep, void* workspace, size_t workspace_bytes, void* stream) {
  int rc = check_call(cfg, w, ids, cu_seqlens, n_seqs, total_tokens, max_seqlen, workspace, workspace_bytes);
  if (rc) return rc;
  LRX_CHECK_ARG(sparse_out && sparse_row_stride >= cfg->vocab_size, "encode_sparse: bad sparse output spec");
  LRX_CHECK_ARG(!dense_out || (dense_dim > 0 && dense_dim <= cfg->hidden_size && dense_row_stride >= dense_dim), "encode_sparse: bad dense output spec");
  hipStream_t s = (hipStream_t)stream;
  const int H = cfg->hidden_size, V = cfg->vocab_size;
  const bool pr = cfg->precise_stream != 0;
  EncWs ws = carve(cfg, total_tokens, n_seqs, (char*)workspace);
  prof_begin();
  if ((rc = forward_layers(cfg, w, ids, cu_seqlens, n_seqs, total_tokens, max_seqlen, ws, false, s))) return rc;
  if (dense_out) {
    ProfScope p(s, 6, 0);
    if ((rc = lrx_pool_norm_shard(pr ? (const void*)ws.x32 : (const void*)ws.x, w->final_norm, cu_seqlens, n_seqs, H, cfg->rms_eps, dense_out, dense_row_stride,
                                  dense_dim, normalize, nullptr, 0, nullptr, pr ? 1 : 0, s))) return rc;
  }
  if ((rc = final_norm_rows(cfg, w, ws, total_tokens, ws.h, s))) return rc;
  { ProfScope p(s, 7, 2.0 * total_tokens * (double)V * H);   // positions are dead after the layers: ws.pos holds the row -> sequence map
    if ((rc = lrx_sparse_max_aggregate(ws.h, lm_head ? lm_head : w->embed, lm_head_bias, cu_seqlens, tok_mask, n_seqs, total_tokens, H, V, sparse_out,
                                       sparse_row_stride, ws.pos, s))) return rc; }
  { ProfScope p(s, 6, 0);
    if ((rc = lrx_sparsify(sparse_out, n_seqs, V, sparse_row_stride, relu, log1p, round_bf16, top_k, min_tokens_to_keep, s))) return rc; }
  return prof_end(s);
}

// ---------------------------------------------------------------------------------------------------------------
// Shared-prefix encode (EmbeddingBag construction): n_seqs sequences = prefix (same for all) + suffix_len own tokens.
// The prefix runs ONCE through the encoder while its per-layer K/V (post-RoPE) are captured; then only the
// n_seqs * suffix_len suffix tokens are pushed through the layers, attending to the captured prefix K/V plus their own
// suffix keys.  Same result as encoding every [prefix + suffix] sequence in full, (P + S2) / S2 times fewer FLOPs.  Both passes
// run the layer loop of the batch path (run_layers) with their own attention step.
// ---------------------------------------------------------------------------------------------------------------
struct PrefWs { EncWs e; char* kvcap; int32_t* cu; size_t total; };
static PrefWs carve_prefixed(const lrx_encoder_config* c, int64_t P1, int64_t n_seqs, int64_t S2, char* base) {
  const int64_t T = (n_seqs * S2 > P1 ? n_seqs * S2 : P1);
  PrefWs w;
  w.e = carve(c, T, n_seqs > 0 ? n_seqs : 1, base);
  size_t off = w.e.total;
  w.kvcap = base + off; off += align_up((size_t)c->num_layers * (size_t)(P1 > 0 ? P1 : 1) * 2 * c->num_kv_heads * c->head_dim * 2, 1024);
  w.cu = (int32_t*)(base + off); off += align_up((size_t)(n_seqs + 2) * 4, 1024);
  w.total = off;
  return w;
}
extern "C" size_t lrx_encode_prefixed_workspace_bytes(const lrx_encoder_config* cfg, int32_t prefix_len, int32_t n_seqs, int32_t suffix_len) {
  if (!cfg) return 0;
  return carve_prefixed(cfg, prefix_len, n_seqs, suffix_len, nullptr).total;
}

struct PrefCtx { const lrx_encoder_config* c; PrefWs* pw; int P1, n_seqs, S2; };
static int prefix_capture(void* ctx, int l, hipStream_t s) {          // K|V (post-RoPE) of the prefix tokens of layer l -> the capture buffer
  const PrefCtx& a = *(const PrefCtx*)ctx;
  const int d = a.c->head_dim, nq = a.c->num_q_heads, nkv = a.c->num_kv_heads;
  const size_t KVW = (size_t)2 * nkv * d, QKV = (size_t)(nq + 2 * nkv) * d, QD = (size_t)nq * d;
  LRX_HIP(hipMemcpy2DAsync(a.pw->kvcap + (size_t)l * a.P1 * KVW * 2, KVW * 2, a.pw->e.qkv + QD * 2, QKV * 2, KVW * 2, a.P1, hipMemcpyDeviceToDevice, s));
  return l == a.c->num_layers - 1 ? 1 : 0;                             // nothing after the last layer's K/V is needed from the prefix
}
static int prefix_attn(void* ctx, int, bool, hipStream_t s) {
  const PrefCtx& a = *(const PrefCtx*)ctx;
  return lrx_attn_varlen_causal_ex(a.pw->e.qkv, a.pw->cu, 1, a.P1, a.P1, a.c->num_q_heads, a.c->num_kv_heads, a.c->head_dim, a.pw->e.h, 0, a.c->precise_stream == 2, s);
}
static int suffix_attn(void* ctx, int l, bool, hipStream_t s) {
  const PrefCtx& a = *(const PrefCtx*)ctx;
  const size_t KVW = (size_t)2 * a.c->num_kv_heads * a.c->head_dim;
  return lrx_attn_prefix_suffix_ex(a.pw->e.qkv, a.pw->kvcap + (size_t)l * a.P1 * KVW * 2, a.n_seqs, a.S2, a.P1, a.c->num_q_heads, a.c->num_kv_heads, a.c->head_dim,
                                   a.pw->e.h, a.c->precise_stream == 2, s);
}

extern "C" int lrx_encode_prefixed(const lrx_encoder_config* cfg, const lrx_encoder_weights* w, const int32_t* prefix_ids, int32_t prefix_len,
                                   const int32_t* suffix_ids, int32_t n_seqs, int32_t suffix_len, float* out, int64_t out_row_stride,
                                   int32_t out_dim, int32_t normalize, void* workspace, size_t workspace_bytes, void* stream) {
  int rc = check_cfg(cfg);
  if (rc) return rc;
  LRX_CHECK_ARG(w && w->embed && w->final_norm && w->rope_cos && w->rope_sin && w->layers, "encode_prefixed: null weights");
  LRX_CHECK_ARG(suffix_ids && n_seqs > 0 && suffix_len > 0 && prefix_len >= 0 && (prefix_len == 0 || prefix_ids), "encode_prefixed: bad batch");
  LRX_CHECK_ARG(prefix_len + suffix_len <= cfg->max_positions, "encode_prefixed: prefix+suffix=%d exceeds RoPE table (%d)", prefix_len + suffix_len,
                cfg->max_positions);
  LRX_CHECK_ARG(out && out_dim > 0 && out_dim <= cfg->hidden_size && out_row_stride >= out_dim, "encode_prefixed: bad output spec");
  LRX_CHECK_ARG(workspace != nullptr, "encode_prefixed: null workspace");
  const size_t need = lrx_encode_prefixed_workspace_bytes(cfg, prefix_len, n_seqs, suffix_len);
  if (workspace_bytes < need) { lrx_set_error("encode_prefixed: workspace %zu B < required %zu B", workspace_bytes, need); return LRX_ERR_WORKSPACE; }
  hipStream_t s = (hipStream_t)stream;
  const lrx_encoder_config* c = cfg;
  PrefWs pw = carve_prefixed(c, prefix_len, n_seqs, suffix_len, (char*)workspace);
  EncWs& ws = pw.e;
  PrefCtx pc = {c, &pw, prefix_len, n_seqs, suffix_len};
  prof_begin();
  // ---- pass 1: the prefix alone (one sequence), capturing K|V of every layer
  if (prefix_len > 0) {
    if ((rc = lrx_uniform_layout(pw.cu, ws.pos, 1, prefix_len, 0, s))) return rc;
    LayerHooks hk = {prefix_attn, prefix_capture, &pc, 0.0};
    if ((rc = run_layers(c, w, prefix_ids, prefix_len, ws, hk, nullptr, 0, s))) return rc;
  }
  // ---- pass 2: the suffix tokens of all sequences
  const int T = n_seqs * suffix_len;
  if ((rc = lrx_uniform_layout(pw.cu, ws.pos, n_seqs, suffix_len, prefix_len, s))) return rc;
  LayerHooks hk = {suffix_attn, nullptr, &pc, 0.0};
  if ((rc = run_layers(c, w, suffix_ids, T, ws, hk, nullptr, 0, s))) return rc;
  { ProfScope p(s, 6, 0);
    const bool pr = c->precise_stream != 0;
    if ((rc = lrx_pool_norm_shard(pr ? (const void*)ws.x32 : (const void*)ws.x, w->final_norm, pw.cu, n_seqs, c->hidden_size, c->rms_eps, out, out_row_stride,
                                  out_dim, normalize, nullptr, 0, nullptr, pr ? 1 : 0, s))) return rc; }
  return prof_end(s);
}
