// Flat inner-product search over an HBM-resident fp32 corpus shard (replaces faiss.IndexFlatIP.search).  ONE translation unit, split by role
// (round 6): this file holds the host side -- kernel dispatch (launch_scores), the plain path (lrx_flat_ip_search), plan_chunk and the
// bounded-search driver, shard maintenance, the exchange kernels -- and includes the device code in source order:
//     lrx_search_filter.h   A. score / filter kernels          lrx_search_select.h   B. selection
//     lrx_search_bounded.h  C. error bound, threshold, fused   lrx_search_refine.h   C. band refine, row-grouped rescoring, part merge
// Map of the unit:
//
//  A. Score kernels (what a search streams the shard through)
//     k_flat_ip_scores<QT>            exact-fp32 MFMA (v_mfma_f32_16x16x4_f32 = an fp32 fma chain), <= 32 queries; writes scores[Q, ld] + block maxima.
//                                     The plain path (lrx_flat_ip_search) for few queries.
//     k_flat_ip_scores_split<..NP..>  NP = 3: six bf16 products ~ fp32 (plain path above 32 queries; the gated exact fallback of the bounded
//                                     search).  NP = 1: one fp16 product straight from the fp32 rows (filter for shards without a shadow).
//     k_filter_xreg / _store / _emit  the filter of the bounded search over the tiled fp16 shadow (lrx_shadow_off): corpus fragments go
//                                     HBM -> registers, queries sit in LDS.  k_filter_xreg: the strided SAMPLE (scores + 16-row group maxima,
//                                     or the whole score matrix in LRX_SEARCH_FILTER_MATRIX mode); _store: its persistent form for large
//                                     samples; _emit: the persistent MAIN pass -- rows reaching the query's threshold are appended to
//                                     per-query candidate lists (per-wave LDS lists, one reservation per (wave, query) per flush).
//                                     (More than 128 queries over a shadow of D >= 1024: BOTH passes run on the GEMM kernel, lrx_gemm.hip
//                                     EPI_SAMPLE / EPI_EMIT, in WIDE chunks of up to 1024 queries -- chunk_queries, round 6.)
//     launch_scores(), k_pack_queries_xb, k_round_queries: query planes / fragment order + the chain's zero-fills; kernel dispatch.
//  B. Selection
//     radix_select_kth*, select_topk_sorted, bitonic_sort_desc: exact k-th / top-k of a score row or a candidate list (one workgroup).
//     k_topk_select, k_topk_select_rescore: the plain path's finish -- RIGOROUS: every row within eps6(q) of the k-th matrix score is
//                                     rescored exactly (fp64 accumulation, one rounding), best k kept; streaming form for huge tie bands.
//  C. The bounded (two-pass) search, lrx_flat_ip_search_bounded: plan_chunk (sample stride, list capacity, workspace layout) ->
//     pack -> sample -> k_sample_threshold (T' = k-th best of the sample, thr = T' - 2 eps(q), lists opened with the sample's rows) ->
//     main pass (_emit) -> k_refine_band (k-th filter score of the list, exact rescoring of the band rows from the fp32 shard) ->
//     k_refine_merge (sorted top-k; or the query's fallback flag) -> gated exact fallback (scores_split<NP=3> + select_rescore).
//     k_refine_topk: the refine step of the score-matrix filter (tiny shards, FILTER_MATRIX).  query_eps_block: the per-query error bound.
//  D. Shard maintenance: k_shard_bounds, k_shard_rows_tiled (lrx_shard_commit_rows): fp16 shadow rows + {max |x|, max |x - fp16(x)|}.
//     Round 4: k_sample_threshold tightens T' to the row-exact k-th sample score when the group maxima are clumpy (corpora stored cluster by
//     cluster); lists hold >= 64 Ki entries; lrx_flat_ip_search_bounded_wire: the chain's last kernel (k_topk_select_rescore, idle unless a
//     query was flagged) also writes every query's results as the 64-bit exchange words; lrx_search_fallback_count /
//     lrx_flat_ip_bounded_list_counts: statistics.
//  E. Multi-GPU result exchange: k_pack_topk / lrx_pack_topk (stand-alone form of the wire words), k_merge_topk / lrx_merge_topk[_packed]
//     (after the RCCL all-gather): lists that arrive in order are RANKED (binary search per other list), anything else is sorted -- in
//     registers (bitonic_sort_desc_regs) from 128 padded entries on.
//
// Every path ends with exact rescoring of the selected rows, so the reported scores do not depend on the path, the query batch size or
// the shard layout; ties go to the lower row id.  Dead ends that were measured and dropped are noted where they would have gone.
#include "lrx_common.h"
#include <float.h>
#include <stdlib.h>

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

#include "lrx_search_filter.h"   // score / filter kernels

extern "C" int64_t lrx_flat_ip_score_ld(int64_t n_rows) { return lrx_cdiv(n_rows > 0 ? n_rows : 1, S_ROWS) * S_ROWS; }

#define SPLIT_MIN_QT 3   // Q > 32 -> split-bf16 kernel (fp32-MFMA-bound otherwise); Q <= 32 stays on the exact-fp32 kernel (HBM-bound)
static size_t split_ws_bytes(int32_t dim) { return (size_t)(dim / 32) * 3 * 8 * 1024; }   // one chunk of <=128 queries

// Which corpus blocks a filter launch covers and what it does with the scores (see k_flat_ip_scores_split)
struct FilterMode {
  int bmode = 0;                          // 0 all blocks, 1 sample blocks (compact stores), 2 non-sample blocks
  int ss = 1;                             // sample stride in units
  int unit = 1;                           // workgroup blocks per sample unit (2: the 256-query main pass works on 256-row tiles)
  int64_t nblocks = -1;                   // workgroups to launch (-1: ld / rows-per-workgroup)
  const float* thr = nullptr;             // emit mode: per-query threshold
  unsigned long long* cand = nullptr;     // emit mode: candidate lists [Q, cap]
  unsigned int* cnt = nullptr;            //            and their fill counts
  unsigned int cap = CAND_CAP_MIN;        //            capacity of one list
  int* zero = nullptr;                    // shadow filter: ints the query-packing kernel clears on the way (first launch of a bounded search)
  int nzero = 0;
  PreSplit presplit;                      // shadow filter: fallback planes the packing kernel writes on the way (ngroups > 0)
  bool planes_ready = false;              // six-product pass: `qsplit` already holds the planes (see PreSplit)
  bool group_max = false;                 // shadow kernels, score stores: `blkmax` receives the maxima of the 16-row wave groups
                                          // (8 per block, row stride 8 x nblk_ld) instead of one maximum per 128-row block
  const struct FusedArgs* fused = nullptr; // sample + selection + main pass in ONE launch (k_filter_fused) right after the query packing
};
struct FusedArgs {
  int64_t ld_s;
  int nblk_s, nblk_ld_s, nsamp, nmain, ss, k;
  const float* bounds;
  float *thr, *eps;
  unsigned long long* cand;
  unsigned int* cnt;
  unsigned int cap;
  FusedCtl* ctl;
  int phases;                             // 7 = sample + selection + main pass (dev: LRX_FUSED_PHASES = 1 or 3 leaves the rest to the old kernels)
};

// planes = 3: fp32-grade scores (six bf16 products); planes = 1: one fp16 product (filter pass of the bounded search, error bound
// query_eps_block); gate != NULL: the whole pass is skipped unless *gate != 0.  One call covers at most one query chunk (128 queries, 256
// for the shadow filter); `ld` is the row stride of `scores` / rounded row count.  Xs = the shard's tiled fp16 shadow (planes == 1) or NULL.
static int lrx_cu_count() {
  // cached per process for the device that was current at the first call (include/lrx.h); C++11 static initialisation: thread-safe
  static const int n_cu = []() {
    int dev = 0;
    hipDeviceProp_t prop;
    return (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
  }();
  return n_cu;
}

static int filter_rows_per_wg(bool shadow) { return shadow ? 128 : 16 * SPF_RT * SPF_WV; }
static int launch_filter_fused(const void* Xs, int64_t n_rows, int dim, const __bf16* qsplit, int nq, int qt, float* scores, float* gmax, const float* qf32,
                               const FusedArgs& fa, hipStream_t s);

static int launch_scores(const float* X, int64_t n_rows, int64_t ldx, int32_t dim, const float* q, int32_t n_queries, float* scores,
                         float* blkmax, __bf16* qsplit, void* stream, int planes = 3, const int* gate = nullptr, const void* Xs = nullptr,
                         int64_t ld = 0, const FilterMode& fm = FilterMode()) {
  LRX_CHECK_ARG(dim > 0 && dim % S_BK == 0, "flat_ip: dim=%d must be a multiple of %d", dim, S_BK);
  LRX_CHECK_ARG(ldx >= dim && ldx % 4 == 0, "flat_ip: ldx=%lld must be >= dim and a multiple of 4", (long long)ldx);
  if (n_rows <= 0 || n_queries <= 0) return LRX_OK;
  const int64_t ld_full = lrx_flat_ip_score_ld(n_rows);
  if (ld == 0) ld = ld_full;
  const int nblk_ld = (int)((ld / SP_ROWS + 3) & ~3);   // blkmax row stride (128-row blocks)
  dim3 block(256);
  hipStream_t s = (hipStream_t)stream;
  const bool emit = fm.cand != nullptr;
  // queries per pass over the corpus: 128 (8 MFMA tiles); the shadow filter takes up to 256 (16 tiles) -- a large query batch then
  // streams the shadow half as often
  const bool shadow_pass = qsplit != nullptr && planes == 1 && Xs != nullptr;
  const int chunk = shadow_pass ? 256 : 128;
  LRX_CHECK_ARG(fm.bmode == 0 || (planes == 1 && qsplit != nullptr && n_queries <= chunk), "flat_ip: sampled filter launch outside the bounded search");
  for (int q0 = 0; q0 < n_queries; q0 += chunk) {
    int nq = n_queries - q0 < chunk ? n_queries - q0 : chunk;
    int qt = (nq + 15) / 16;
    const float* qp = q + (int64_t)q0 * dim;
    float* sp = scores ? scores + (int64_t)q0 * ld : nullptr;
    float* bp = blkmax ? blkmax + (int64_t)q0 * nblk_ld : nullptr;
    if (shadow_pass) {   // any query count: the shadow pass beats the exact-fp32 kernel from Q = 1
      // filter pass over the fp16 shadow of the corpus: half the bytes of the fp32 rows, fragments streamed through registers
      const int64_t nwg = fm.nblocks >= 0 ? fm.nblocks : ld_full / 128;
      if (nwg == 0) continue;
      if (fm.bmode != 2) {   // (the main pass of the score-free filter reuses the planes packed for its sample pass)
        int threads = (dim / 64) * 2 * qt * 64;
        PreSplit ps = fm.presplit;
        ps.nb_xb = (threads + 255) / 256;
        hipLaunchKernelGGL(k_pack_queries_xb, dim3(ps.nb_xb + (ps.ngroups > 0 ? ps.blocks[0] + ps.blocks[1] : 0)), dim3(256), 0, s, qp, nq, dim, qt, qsplit,
                           fm.zero, fm.nzero, ps);
      }
      if (fm.fused != nullptr) {   // the whole filter chain of the score-free search in one persistent launch
        const int rc = launch_filter_fused(Xs, n_rows, dim, qsplit, nq, qt, sp, bp, qp, *fm.fused, s);
        if (rc != LRX_OK) return rc;
        continue;
      }
      // Main pass of the score-free filter: persistent workgroups, one per CU (D / 64 a multiple of the ring depth); everything
      // else: one workgroup per 128-row block
      constexpr int XPF = 4;
#ifndef XPF_S
#define XPF_S 4   // ring depth of the one-workgroup-per-block form (8: no faster at D = 2048, 13 % slower at 10M x 256, Q = 1)
#endif
      const int n_cu = lrx_cu_count();
      // (round 4, profiles/r04_emit_persist_ab.txt: persistent equals one workgroup per block at 3.8 blocks per CU and wins from 7.6 on -- the
      // switch that A/B'ed it is gone, the one-workgroup-per-block emit form remains for D / 64 not a multiple of the ring depth)
      const bool persistent = emit && gate == nullptr && (dim / 64) % XPF == 0 && fm.bmode != 1 && nwg < (1ll << 31);
      // sample pass: two blocks per workgroup (the q slice is fetched once per 256 rows) once there are more sample blocks than CUs; a
      // sample that fits the chip in one round runs one block per workgroup -- its time is the time of ONE workgroup's blocks through
      // one CU (~20 us per 512-KiB block), not a throughput question (125 k-row shard: 34 -> ~20 us, 1M x 2048 at ss = 32: 52 -> ~27 us)
      const bool two_blocks = !emit && fm.bmode == 1 && nwg < (1ll << 31) && nwg > lrx_cu_count();
      // sample pass of a large shard (more than two block pairs per CU): persistent workgroups
      const bool persistent_store = two_blocks && fm.group_max && gate == nullptr && sp != nullptr && bp != nullptr && qt <= 8 && (dim / 64) % XPF == 0 &&
                                    (nwg + 1) / 2 > 2 * (int64_t)n_cu;
#define LRX_XP(QQ, RT_)                                                                                                                     \
  {                                                                                                                                         \
    const int64_t groups = (nwg + RT_ - 1) / RT_;                                                                                           \
    hipLaunchKernelGGL((k_filter_xreg_emit<QQ, (QQ == 8 ? 2 : XPF), RT_>), dim3((unsigned)(groups < n_cu ? groups : n_cu)), dim3(576), 0, s, (const __bf16*)Xs, n_rows, \
                       dim, qsplit, nq, (int)nwg, fm.bmode, fm.ss, fm.unit, fm.thr, fm.cand, fm.cnt, fm.cap);                              \
  }
#define LRX_XN(QQ, EM_)                                                                                                                     \
  hipLaunchKernelGGL((k_filter_xreg<QQ, (EM_ ? XPF : XPF_S), EM_>), dim3((unsigned)nwg), dim3(576), 0, s, (const __bf16*)Xs, n_rows, dim, qsplit, nq, sp, ld, bp, \
                     nblk_ld, gate, fm.bmode, fm.ss, fm.unit, fm.thr, fm.cand, fm.cnt, (int)nwg, fm.group_max ? 1 : 0, fm.cap);
#define LRX_XN2(QQ)   /* sample pass: two blocks per workgroup */                                                                           \
  hipLaunchKernelGGL((k_filter_xreg<QQ, XPF, false, 2>), dim3((unsigned)((nwg + 1) / 2)), dim3(576), 0, s, (const __bf16*)Xs, n_rows, dim, qsplit, nq, sp, ld, bp, \
                     nblk_ld, gate, fm.bmode, fm.ss, fm.unit, fm.thr, fm.cand, fm.cnt, (int)nwg, fm.group_max ? 1 : 0, fm.cap);
#define LRX_XS(QQ)   /* sample pass of a large shard: persistent, two blocks at a time (one from seven query tiles on: registers) */                                                  \
  hipLaunchKernelGGL((k_filter_xreg_store<(QQ), XPF, ((QQ) <= 6 ? 2 : 1)>), dim3((unsigned)n_cu), dim3(576), 0, s, (const __bf16*)Xs, n_rows, dim, qsplit, nq, sp, ld, bp, \
                     nblk_ld, (int)nwg, fm.bmode, fm.ss, fm.unit);
#define LRX_XR(QQ, RT_)                                  \
  case QQ:                                               \
    if (persistent) LRX_XP(QQ, RT_)                      \
    else if (emit) { LRX_XN(QQ, true) }                  \
    else if (persistent_store && QQ <= 8) { LRX_XS(QQ <= 8 ? QQ : 1) } \
    else if (two_blocks && QQ <= 8) { LRX_XN2(QQ <= 8 ? QQ : 1) } \
    else { LRX_XN(QQ, false) }                           \
    break;
      switch (qt) { LRX_XR(1, 2) LRX_XR(2, 2) LRX_XR(3, 2) LRX_XR(4, 2) LRX_XR(5, 2) LRX_XR(6, 2) LRX_XR(7, 2) LRX_XR(8, 2)
                   LRX_XR(9, 1) LRX_XR(10, 1) LRX_XR(11, 1) LRX_XR(12, 1) LRX_XR(13, 1) LRX_XR(14, 1) LRX_XR(15, 1) LRX_XR(16, 1) }
#undef LRX_XR
#undef LRX_XN
#undef LRX_XN2
#undef LRX_XS
#undef LRX_XP
      LRX_LAUNCH_CHECK();
      continue;
    }
    if (qsplit != nullptr && qt >= SPLIT_MIN_QT) {
      const int rb = planes == 3 ? 128 : 16 * SPF_RT * SPF_WV;
      const int64_t nwg = fm.nblocks >= 0 ? fm.nblocks : ld_full / rb;
      // grid of a gated (normally idle) six-product launch: its workgroups walk the blocks.  (Round 4: 1 x, 2 x and 8 x CUs idle equally fast --
      // 4.7-5.0 us under rocprofv3, of which ~3 us are the profiler's per-dispatch overhead: an empty kernel costs 1.5 us per dependent launch.)
      const int64_t gated_cap = 8 * (int64_t)lrx_cu_count();
      if (nwg == 0) continue;
      if (fm.bmode != 2 && !fm.planes_ready) {
        int threads = (dim / 32) * qt * 64;
        hipLaunchKernelGGL(k_split_queries, dim3((threads + 255) / 256), dim3(256), 0, s, qp, nq, dim, qt, planes, qsplit, gate);
      }
#define LRX_SF_(QQ, EM_)                                                                                                                          \
    hipLaunchKernelGGL((k_flat_ip_scores_split<QQ, 1, SPF_RT, SPF_NST, SPF_WV, EM_>), dim3((unsigned)nwg), dim3(64 * SPF_WV), 0, s, X, n_rows, ldx, \
                       dim, qsplit, nq, sp, ld, bp, nblk_ld, gate, fm.bmode, fm.ss, fm.unit, fm.thr, fm.cand, fm.cnt, (int64_t)nwg, fm.cap);
#define LRX_SS(QQ)                                                                                                                              \
  case QQ:                                                                                                                                      \
    if (planes == 3) hipLaunchKernelGGL((k_flat_ip_scores_split<QQ, 3, 2, 2, 4>), dim3((unsigned)(gate != nullptr && nwg > gated_cap ? gated_cap : nwg)), block, 0, s, X, n_rows, ldx, dim, qsplit, nq, sp, ld, bp, nblk_ld, gate, 0, 1, 1, (const float*)nullptr, (unsigned long long*)nullptr, (unsigned int*)nullptr, (int64_t)nwg, 0u); \
    else if (emit) { LRX_SF_(QQ, true) }                                                                                                        \
    else { LRX_SF_(QQ, false) }                                                                                                                 \
    break;
      switch (qt) { LRX_SS(3) LRX_SS(4) LRX_SS(5) LRX_SS(6) LRX_SS(7) LRX_SS(8) }
#undef LRX_SS
#undef LRX_SF_
      LRX_LAUNCH_CHECK();
      continue;
    }
    LRX_CHECK_ARG(fm.bmode == 0 && !emit, "flat_ip: the exact-fp32 score kernel has no sampled / emitting mode");
    dim3 grid((unsigned)(ld_full / S_ROWS));
#define LRX_SC(QQ) case QQ: hipLaunchKernelGGL(k_flat_ip_scores<QQ>, grid, block, 0, s, X, n_rows, ldx, dim, qp, nq, sp, ld, bp, nblk_ld, gate); break;
    switch (qt) { LRX_SC(1) LRX_SC(2) LRX_SC(3) LRX_SC(4) LRX_SC(5) LRX_SC(6) LRX_SC(7) LRX_SC(8) }
#undef LRX_SC
    LRX_LAUNCH_CHECK();
  }
  return LRX_OK;
}

extern "C" int lrx_flat_ip_scores(const float* X, int64_t n_rows, int64_t ldx, int32_t dim, const float* q, int32_t n_queries,
                                  float* scores, void* stream) {
  return launch_scores(X, n_rows, ldx, dim, q, n_queries, scores, nullptr, nullptr, stream);
}

#include "lrx_search_select.h"   // selection primitives, k_topk_select[_rescore]

extern "C" size_t lrx_flat_ip_workspace_bytes(int64_t n_rows, int32_t dim, int32_t n_queries, int32_t k) {
  (void)k;
  const size_t ld = (size_t)lrx_flat_ip_score_ld(n_rows), nq = (size_t)(n_queries > 0 ? n_queries : 1);
  // scores [Q, ld] + per-block maxima [Q, ~ld/256] + split-bf16 query planes of one 128-query chunk + the two row bounds of a caller
  // that does not know them
  return (ld * nq + ((ld / SP_ROWS + 3) & ~(size_t)3) * nq) * sizeof(float) + split_ws_bytes(dim) + 256;
}

static int shard_bounds_launch(const float* X, int64_t ldx, int64_t n_rows, int32_t dim, float* row_bounds, hipStream_t s);

extern "C" int lrx_flat_ip_search(const float* X, int64_t n_rows, int64_t ldx, int32_t dim, const float* row_bounds, const float* q, int32_t n_queries,
                                  int32_t k, int64_t id_base, float* out_scores, int64_t* out_ids, void* workspace, size_t workspace_bytes,
                                  void* stream) {
  LRX_CHECK_ARG(k > 0 && k <= SEL_MAXK, "flat_ip_search: k=%d out of range (1..%d)", k, SEL_MAXK);
  LRX_CHECK_ARG(n_rows >= 0 && n_rows < (1ll << 32), "flat_ip_search: shard rows=%lld out of range", (long long)n_rows);
  if (n_queries <= 0) return LRX_OK;
  if (workspace_bytes < lrx_flat_ip_workspace_bytes(n_rows, dim, n_queries, k)) {
    lrx_set_error("flat_ip_search: workspace %zu B < required %zu B", workspace_bytes, lrx_flat_ip_workspace_bytes(n_rows, dim, n_queries, k));
    return LRX_ERR_WORKSPACE;
  }
  float* scores = (float*)workspace;
  const int64_t ld = lrx_flat_ip_score_ld(n_rows);
  float* blkmax = scores + ld * (int64_t)n_queries;
  const int nblk = (int)(ld / SP_ROWS), nblk_ld = (nblk + 3) & ~3;   // block maxima at 128-row granularity on both paths
  __bf16* qsplit = (__bf16*)(blkmax + (int64_t)nblk_ld * n_queries);
  if (n_rows > 0) {
    int rc = launch_scores(X, n_rows, ldx, dim, q, n_queries, scores, blkmax, qsplit, stream);
    if (rc != LRX_OK) return rc;
  }
  if (n_rows > 0 && dim % 4 == 0) {
    // the band of the final selection needs max |row|: a caller that keeps the shard's bounds passes them; otherwise one more read of the rows
    if (row_bounds == nullptr) {
      float* wb = (float*)((char*)qsplit + split_ws_bytes(dim));
      LRX_HIP(hipMemsetAsync(wb, 0, 2 * sizeof(float), (hipStream_t)stream));
      int rc = shard_bounds_launch(X, ldx, n_rows, dim, wb, (hipStream_t)stream);
      if (rc != LRX_OK) return rc;
      row_bounds = wb;
    }
    hipLaunchKernelGGL(k_topk_select_rescore, dim3(n_queries), dim3(SEL_THREADS), 0, (hipStream_t)stream, (const float*)scores, ld, n_rows, k, id_base,
                       (const float*)blkmax, nblk, nblk_ld, X, ldx, dim, q, out_scores, out_ids, (const int*)nullptr, (const int*)nullptr, row_bounds,
                       (unsigned long long*)nullptr, (const int64_t*)nullptr);
  } else {
    hipLaunchKernelGGL(k_topk_select, dim3(n_queries), dim3(SEL_THREADS), 0, (hipStream_t)stream, scores, ld, n_rows, k, id_base, blkmax, nblk,
                       nblk_ld, out_scores, out_ids, (const int*)nullptr, (const int*)nullptr);
  }
  LRX_LAUNCH_CHECK();
  return LRX_OK;
}

#include "lrx_search_bounded.h"  // error bound, k_sample_threshold, k_filter_fused

extern "C" int lrx_probe_fused_timestamps(uint64_t* out, int32_t n_words) {
  LRX_CHECK_ARG(out != nullptr && n_words > 0 && n_words <= 1024 * 8, "probe_fused_timestamps: bad buffer");
  LRX_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fused_ts), (size_t)n_words * 8));
  return LRX_OK;
}

static int launch_filter_fused(const void* Xs, int64_t n_rows, int dim, const __bf16* qsplit, int nq, int qt, float* scores, float* gmax, const float* qf32,
                               const FusedArgs& fa, hipStream_t s) {
  const int n_cu = lrx_cu_count();
  // two blocks at a time (the q slice is read from LDS once for both) once a workgroup has at least four such steps; fewer, finer steps otherwise
  const bool rt2 = fa.nmain >= 8 * n_cu;
#define LRX_FU(QQ)                                                                                                                               \
  case QQ:                                                                                                                                        \
    if (rt2) hipLaunchKernelGGL((k_filter_fused<QQ, (QQ == 8 ? 2 : 4), 2>), dim3(n_cu), dim3(576), 0, s, (const __bf16*)Xs, n_rows, dim, qsplit, nq, scores, fa.ld_s, \
                                gmax, fa.nblk_s, fa.nblk_ld_s, fa.nsamp, fa.nmain, fa.ss, fa.k, qf32, fa.bounds, fa.thr, fa.eps, fa.cand, fa.cnt, fa.cap, fa.ctl, fa.phases);  \
    else hipLaunchKernelGGL((k_filter_fused<QQ, (QQ == 8 ? 2 : 4), 1>), dim3(n_cu), dim3(576), 0, s, (const __bf16*)Xs, n_rows, dim, qsplit, nq, scores, fa.ld_s,     \
                            gmax, fa.nblk_s, fa.nblk_ld_s, fa.nsamp, fa.nmain, fa.ss, fa.k, qf32, fa.bounds, fa.thr, fa.eps, fa.cand, fa.cnt, fa.cap, fa.ctl, fa.phases);      \
    break;
  switch (qt) { LRX_FU(1) LRX_FU(2) LRX_FU(3) LRX_FU(4) LRX_FU(5) LRX_FU(6) LRX_FU(7) LRX_FU(8)
    default: lrx_set_error("filter_fused: %d query tiles", qt); return LRX_ERR_INVALID; }
#undef LRX_FU
  LRX_LAUNCH_CHECK();
  return LRX_OK;
}

#include "lrx_search_refine.h"   // band refine, row-grouped rescoring, merge of the parts

// ---- host side: per query chunk (<= 256 queries with the shadow, <= 128 without) one pipeline over one workspace ----------------
// flags & 3 (lrx.h LRX_SEARCH_FILTER_*): 0 = choose the filter per chunk, 1 = always the score-matrix filter, 2 = the score-free filter
// whenever the shape allows, 3 = like 2 but never the GEMM kernel for the main pass (A/B runs).  Per call: no process-wide state.
struct BoundedPlan {
  bool fused;                           // emit + one persistent launch for sample, selection and main pass (k_filter_fused)
  bool emit;
  bool gemm;                            // emit: the main pass runs on the GEMM kernel (129..256 queries over the tiled shadow)
  int ss, rb;
  unsigned int cap;                     // emit: capacity of one candidate list
  int64_t ld, nblk, nblk_ld;            // full shard: score row stride, 128-row blocks, blkmax row stride
  int64_t nsamp_wg, nmain_wg;           // emit: workgroups of the sample / main launch (rb rows each)
  int64_t ld_s, nblk_s, nblk_ld_s;      // emit: the compact sample matrix
  size_t off_qsplit, off_q16, off_qs3, off_ints, off_parts, off_cand, total;   // byte offsets into the workspace (the score region starts at 0)
};
static size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

// flags[nq], any_flag[8] (one per 128-query group of the chunk: the gated fallback of a group runs when one of ITS queries overflowed), ...,
// the fused kernel's five counters in the last 8
#define ANY_FLAG_GROUPS (LRX_EMIT_MAX_QUERIES / 128)
static size_t ints_before_cnt(int nq) { return ((size_t)nq + ANY_FLAG_GROUPS + 8 + 63) & ~(size_t)63; }

// Candidate-list capacity and sample stride for top-k: ~k * ss rows reach the sample's k-th score, the fp16 band adds ~30 % -- the list
// should end up around a third full (a list that overflows sends its query to the exact fallback).  k <= 256: 16 Ki entries and ss = 20
// (1/20 of the shard goes through the sample pass); larger k: the capacity grows with k (64 k, a power of two) so that the stride stays
// at 20 -- round 2 kept 16 Ki entries for every k, which at the reference's default top_k = 1000 meant ss = 2: half of the shard went
// through the sample pass and its [Q, N/2] score matrix.
#ifndef SAMPLE_SS_MAX
#define SAMPLE_SS_MAX 32   // (20 in round 2: the bf16 band doubled the list entries per sample row)
#endif
static unsigned int cand_cap_for(int32_t k) {
  unsigned int cap = CAND_CAP_MIN;
  while (cap < 64u * (unsigned int)(k > 0 ? k : 1)) cap <<= 1;
  return cap;
}

// LRX_SEARCH_FUSED (dev builds only; the `flags` bits LRX_SEARCH_FUSED_ALWAYS / _NEVER are the per-call form): unset = the measured rule in
// plan_chunk (small query batches over small shards), 0 = never (the three-launch chain of rounds 2-4), 1 = wherever the fused kernel is
// eligible (A/B runs: tools/exp/fused_ab.sh).  Read once, thread-safe.
static int search_fused_mode() {
  static const int v = lrx_dev_knob("LRX_SEARCH_FUSED", -1) < 0 ? -1 : (lrx_dev_knob("LRX_SEARCH_FUSED", -1) == 0 ? 0 : 1);
  return v;
}
// fused preference of a call: -1 = by the rule, 0 = never, 1 = wherever eligible; flag bits LRX_SEARCH_FUSED_NEVER / _ALWAYS win over the environment
static int fused_pref_of(int flags) { return (flags & 8) ? 0 : ((flags & 4) ? 1 : search_fused_mode()); }

static BoundedPlan plan_chunk(int64_t n_rows, int32_t dim, int32_t nq, int32_t k, bool shadow, int mode, int fused_pref) {
  BoundedPlan p;
  memset(&p, 0, sizeof(p));
  // more than 128 queries over the shadow: the main pass is the GEMM kernel on 256-row tiles (the sample then moves in 256-row units)
  p.gemm = shadow && nq > 128 && dim >= 1024 && dim % 64 == 0 && n_rows < (1ll << 31) && mode != 3;   // (D = 256: 4 K-tiles per 256 x 256 tile, 14 % slower than the 128-row kernel)
  p.rb = p.gemm ? 256 : filter_rows_per_wg(shadow);
  p.ld = lrx_flat_ip_score_ld(n_rows);
  p.nblk = p.ld / SP_ROWS;
  p.nblk_ld = (p.nblk + 3) & ~(int64_t)3;
  p.cap = cand_cap_for(k);
  // (LRX_SS_FORCE, dev builds: A/B runs of the sample stride on one box; read once, thread-safe.  Strides beyond 32 were measured in round 4,
  // profiles/r04_ss_max_ab.txt: nothing to gain -- the cap is a constant)
  static const int ss_force = lrx_dev_knob("LRX_SS_FORCE", 0) >= 2 ? lrx_dev_knob("LRX_SS_FORCE", 0) : 0;
  const int ss_max = SAMPLE_SS_MAX;
  const int64_t nwg = lrx_cdiv(n_rows > 0 ? n_rows : 1, p.rb);
  // The sample stride trades the sample pass against the hits of the main pass: T' is the k-th best of the sample, so ~k * ss rows per query
  // reach it (appended, selected from and band-checked in the refine step), while the sample pass scores rows / ss rows per query into a
  // matrix and runs apart from the persistent main pass.  Three measured rules (tools/exp/ss_sweep.sh, same-box A/B, 100 queries unless noted):
  //  (a) hits: cost ~ a * rows / ss + b * k * ss -> ss ~ sqrt(rows / k); factor 0.25 from top-1000: 1M x 2048 0.96 ms at 8 vs 1.03 at 20,
  //      256 queries 1.72 at 4 vs 1.95; 100 k rows 0.34 at 2 vs 0.44 (256 queries 0.67 vs 1.00: a fifth of all scores were hits at 20);
  //  (b) chip fill: a sample of few blocks per CU is a launch that mostly waits -- by blocks per CU of the shard (k = 100): 3.8 (125 k rows)
  //      best at 2-6, 7.6 (250 k) at 4: 0.263 vs 0.276 at 20, 15 (500 k) at 8: 0.42 vs 0.44, 30 (1M) at 32: 0.715 vs 0.721 at 16, 0.733 at 8;
  //  (c) powers of two only: odd strides are slower by 3-8 % at every size (1M: 0.75-0.76 at 15 against 0.72 at 16; 250 k: 0.286 at 3
  //      against 0.263 at 4) -- the main pass then walks runs of an even number of blocks with its even number of workgroups.
  const int64_t bpc = nwg / lrx_cu_count();
  const int ss_fill = bpc < 6 ? 2 : (bpc < 12 ? 4 : (bpc < 24 ? 8 : 32));
  const int ss_hits = (int)(0.25 * 1.41 * sqrt((double)(n_rows > 0 ? n_rows : 1) / (double)(k > 0 ? k : 1)));   // (x sqrt 2: to the NEAREST power of two below)
  const int ss_list = (int)(p.cap / (3u * (unsigned int)(k > 0 ? k : 1)));   // ~k * ss hits per query must fit the list three times over
  int ss_lim = ss_hits < ss_list ? ss_hits : ss_list;
  ss_lim = ss_lim < ss_fill ? ss_lim : ss_fill;
  int ss = 2;
  while (ss * 2 <= ss_lim) ss *= 2;
  ss = ss > ss_max ? ss_max : (ss < 2 ? 2 : ss);
  if (ss_force) ss = ss_force < ss_list ? ss_force : ss_list;
  // Fused launch (<= 128 queries over the shadow, D / 64 a multiple of the ring depth): the sample runs inside the persistent kernel, at the
  // chip's full rate, so chip fill is no concern and a block costs the same in either pass -- the sample is ONE block per workgroup (every CU
  // works through the sample phase together; a larger sample would only lengthen the wait before the selection), unless the hit / list rules ask for more
  // Where it pays (profiles/r05_fused_ab.txt, same box, fused / chain): 125 k x 2048, k = 100: 0.91 / 0.97 / 0.96 at Q = 1 / 16 / 32 but 1.03 / 1.15 at
  // 64 / 100 -- with seven query tiles ONE block per workgroup is bound by its LDS fragment reads and MFMA issue (36 us against 22 us for a block
  // of the two-at-a-time main pass: profiles/r05_fused_timeline.txt), and every selection waits for the slowest sample block; 250 k rows 0.97-0.99
  // up to Q = 32; 1M x 2048 and 1M x 4096 1.00-1.03; D = 256 (q resident, no per-step barrier) 1.07-1.22; k = 1000 1.08-1.20.  Hence the rule
  // (Q = 32 is a tie: 0.149 / 0.155 in separate processes, 0.131 / 0.130 back to back inside bench.py, so the rule stops at one query tile):
  // at most 16 queries, D >= 512, k <= 256, at most 8 blocks per CU.  LRX_SEARCH_FUSED=1 lifts the rule (not the eligibility).
  const bool fused_rule = nq <= 16 && dim >= 512 && k <= 256 && nwg <= 8 * (int64_t)lrx_cu_count();
  bool fused_ok = fused_pref != 0 && (fused_rule || fused_pref == 1) && shadow && !p.gemm && nq <= 128 && dim % 256 == 0 && n_rows < (1ll << 31) - 256;
  if (fused_ok && !ss_force) {
    int lim = ss_hits < ss_list ? ss_hits : ss_list;
    lim = lim < 2 ? 2 : (lim > 64 ? 64 : lim);
    const int one_round = (int)lrx_cdiv(nwg, lrx_cu_count());
    ss = one_round < 2 ? 2 : (one_round > lim ? lim : one_round);
  }
  while (ss > 2 && (lrx_cdiv(nwg, ss) - 1) * p.rb < 2 * (int64_t)k) ss = fused_ok ? ss - 1 : ss >> 1;
  const int64_t nsamp = lrx_cdiv(nwg, ss);
  // (the fused kernel's selection works on the maxima of the sample's 16-row groups alone -- it has no sorted fallback over the scores --
  // and needs k of them: nsamp blocks x 8 groups.  Smaller samples take the three-launch chain.)
  if (nsamp * (p.rb / 16) < (int64_t)k) fused_ok = false;
  const bool feasible = (nsamp - 1) * p.rb >= 2 * (int64_t)k && nwg - nsamp >= 1 && dim % 4 == 0 && (shadow || nq > 16 * (SPLIT_MIN_QT - 1));
  if (!feasible) p.gemm = false;
  p.emit = feasible && mode != 1 && (mode >= 2 || n_rows >= 16384);
  p.fused = p.emit && fused_ok && !p.gemm;
  p.ss = ss;
  p.nsamp_wg = nsamp;
  p.nmain_wg = nwg - nsamp;
  p.ld_s = lrx_flat_ip_score_ld(nsamp * p.rb);
  p.nblk_s = nsamp * p.rb / SP_ROWS;
  p.nblk_ld_s = (p.ld_s / SP_ROWS + 3) & ~(int64_t)3;
  const size_t fb = (size_t)(nq < 128 ? nq : 128) * (size_t)(p.ld + p.nblk_ld);
  const size_t prim = p.emit ? (size_t)nq * (size_t)(p.ld_s + 8 * p.nblk_ld_s) : (size_t)nq * (size_t)(p.ld + p.nblk_ld);   // (sample: group maxima, 8 per block)
  p.off_qsplit = align256((prim > fb ? prim : fb) * sizeof(float));
  p.off_q16 = align256(p.off_qsplit + split_ws_bytes(dim));
  const size_t nq256 = ((size_t)(nq > 256 ? nq : 256) + 255) & ~(size_t)255;   // (a wide chunk: up to LRX_EMIT_MAX_QUERIES queries)
  p.off_qs3 = align256(p.off_q16 + nq256 * dim * 2);                // planes of the gated fallback, groups of <= 128 queries
  p.off_ints = align256(p.off_qs3 + (nq256 / 128) * split_ws_bytes(dim));
  // ints: flags[nq], any_flag, pad to 64 ints, cnt[nq * CNT_STRIDE] (one memset) | part_cnt[nq * REF_SPLIT] | thr[nq] | eps[nq]
  p.off_parts = align256(p.off_ints + sizeof(int) * (ints_before_cnt(nq) + (size_t)nq * (CNT_STRIDE + 2 + REF_SPLIT)));
  p.off_cand = align256(p.off_parts + (size_t)nq * REF_CAND * 8);
  p.total = p.off_cand + (p.emit ? (size_t)nq * p.cap * 8 : 0);
  return p;
}

// Queries per chunk of one call.  256 over a shadow (128 over fp32 rows) -- or, round 6, a WIDE chunk of up to LRX_EMIT_MAX_QUERIES where the
// main pass can run on the GEMM kernel (shadow, D >= 1024, the score-free filter feasible): the query count is split into equal chunks (a
// multiple of 16 each), so that 1000 queries are ONE pass over the shadow instead of four.  LRX_SEARCH_WIDE_MAX (dev builds): A/B of the width.
static int chunk_queries(int64_t n_rows, int32_t dim, int32_t n_queries, int32_t k, bool shadow, int mode) {
  const int base = shadow ? 256 : 128;
  if (!shadow || n_queries <= base) return base;
  static const int wide_env = lrx_dev_knob("LRX_SEARCH_WIDE_MAX", 0);
  const int wide_max = wide_env >= 256 && wide_env <= LRX_EMIT_MAX_QUERIES ? (wide_env & ~255) : LRX_EMIT_MAX_QUERIES;
  if (wide_max <= base) return base;
  const int nchunks = (n_queries + wide_max - 1) / wide_max;
  const int wide = (((n_queries + nchunks - 1) / nchunks) + 15) & ~15;
  const BoundedPlan p = plan_chunk(n_rows, dim, wide < n_queries ? wide : n_queries, k, shadow, mode, 0);
  return (p.gemm && p.emit) ? wide : base;
}

extern "C" int32_t lrx_flat_ip_bounded_chunk_queries(int64_t n_rows, int32_t dim, int32_t n_queries, int32_t k, int32_t flags, int32_t has_shadow) {
  return chunk_queries(n_rows, dim, n_queries > 0 ? n_queries : 1, k, has_shadow && dim % 64 == 0, flags & 3);
}

extern "C" size_t lrx_flat_ip_bounded_workspace_bytes(int64_t n_rows, int32_t dim, int32_t n_queries, int32_t k, int32_t flags) {
  // the search walks the queries in chunks (chunk_queries: 256 over a shadow, up to LRX_EMIT_MAX_QUERIES where the GEMM main pass applies; 128
  // over fp32 rows), each chunk with its own plan over the same buffer; sized for the filter `flags` selects (the score-matrix filter of
  // LRX_SEARCH_FILTER_MATRIX needs [min(Q, 256), rows] floats), with or without a shadow
  const int32_t nq = n_queries > 0 ? n_queries : 1;
  const int mode = flags & 3;
  size_t need = lrx_flat_ip_workspace_bytes(n_rows, dim, nq < 128 ? nq : 128, k);   // tiny shards / few queries without shadow: plain path in chunks of 128
  for (int sh = 0; sh < 2; ++sh) {
    const int chunk = chunk_queries(n_rows, dim, nq, k, sh != 0, mode);
    const int sizes[2] = {nq < chunk ? nq : chunk, nq > chunk ? nq % chunk : 0};
    for (int i = 0; i < 2; ++i)
      if (sizes[i] > 0) {
        size_t t = plan_chunk(n_rows, dim, sizes[i], k, sh != 0, mode, 0).total;
        const size_t tf = plan_chunk(n_rows, dim, sizes[i], k, sh != 0, mode, 1).total;   // (either chain may run: flags, LRX_SEARCH_FUSED)
        t = tf > t ? tf : t;
        need = t > need ? t : need;
      }
  }
  return need + 512;
}

extern "C" int lrx_flat_ip_search_bounded(const float* X, int64_t n_rows, int64_t ldx, int32_t dim, const void* X_shadow, const float* row_bounds,
                                          const float* q, int32_t n_queries, int32_t k, int64_t id_base, float* out_scores, int64_t* out_ids,
                                          void* workspace, size_t workspace_bytes, int32_t flags, void* stream) {
  return lrx_flat_ip_search_bounded_wire(X, n_rows, ldx, dim, X_shadow, row_bounds, q, n_queries, k, id_base, out_scores, out_ids, nullptr, nullptr,
                                         workspace, workspace_bytes, flags, stream);
}

extern "C" int lrx_flat_ip_search_bounded_wire(const float* X, int64_t n_rows, int64_t ldx, int32_t dim, const void* X_shadow, const float* row_bounds,
                                               const float* q, int32_t n_queries, int32_t k, int64_t id_base, float* out_scores, int64_t* out_ids,
                                               const int64_t* row_map, uint64_t* out_wire, void* workspace, size_t workspace_bytes, int32_t flags,
                                               void* stream) {
  LRX_CHECK_ARG(row_bounds != nullptr, "flat_ip_search_bounded: null row_bounds (device pointer to {max |x_row|, max |x_row - fp16(x_row)|})");
  LRX_CHECK_ARG(k > 0 && k <= SEL_MAXK, "flat_ip_search: k=%d out of range (1..%d)", k, SEL_MAXK);
  LRX_CHECK_ARG(n_rows >= 0 && n_rows < (1ll << 32), "flat_ip_search: shard rows=%lld out of range", (long long)n_rows);
  LRX_CHECK_ARG((flags & ~63) == 0 && (flags & 12) != 12 && (flags & 48) != 48, "flat_ip_search_bounded: unknown flags 0x%x", flags);
  if (n_queries <= 0) return LRX_OK;
  if (workspace_bytes < lrx_flat_ip_bounded_workspace_bytes(n_rows, dim, n_queries, k, flags)) {
    lrx_set_error("flat_ip_search_bounded: workspace %zu B < required %zu B", workspace_bytes, lrx_flat_ip_bounded_workspace_bytes(n_rows, dim, n_queries, k, flags));
    return LRX_ERR_WORKSPACE;
  }
  const int mode = flags & 3;
  const bool shadow = X_shadow != nullptr && dim % 64 == 0;
  const int qt_max = ((n_queries < 128 ? n_queries : 128) + 15) / 16;
  // tiny shards and odd widths take the plain path; so do small query batches without a shadow (HBM-bound on the exact-fp32 kernel already)
  if ((qt_max < SPLIT_MIN_QT && !shadow) || n_rows <= REF_CAND || dim % 4 != 0) {
    for (int q0 = 0; q0 < n_queries; q0 += 128) {
      const int nq = n_queries - q0 < 128 ? n_queries - q0 : 128;
      const int rc = lrx_flat_ip_search(X, n_rows, ldx, dim, row_bounds, q + (int64_t)q0 * dim, nq, k, id_base, out_scores + (int64_t)q0 * k,
                                        out_ids + (int64_t)q0 * k, workspace, workspace_bytes, stream);
      if (rc != LRX_OK) return rc;
    }
    // (the plain path of tiny shards has no fused tail: the wire words by the stand-alone packing kernel)
    return out_wire != nullptr ? lrx_pack_topk(out_scores, out_ids, row_map, id_base, (int64_t)n_queries * k, out_wire, stream) : LRX_OK;
  }
  hipStream_t s = (hipStream_t)stream;
  const int chunk = chunk_queries(n_rows, dim, n_queries, k, shadow, mode);
  const int use_fused = fused_pref_of(flags);
  for (int q0 = 0; q0 < n_queries; q0 += chunk) {
    const int nq = n_queries - q0 < chunk ? n_queries - q0 : chunk;
    const BoundedPlan p = plan_chunk(n_rows, dim, nq, k, shadow, mode, use_fused);
    LRX_CHECK_ARG(nq <= (shadow ? 256 : 128) || (p.gemm && p.emit), "flat_ip_search_bounded: a chunk of %d queries without the GEMM main pass", nq);
    if (p.total > workspace_bytes) {      // (cannot happen with a workspace sized by lrx_flat_ip_bounded_workspace_bytes for the same flags)
      lrx_set_error("flat_ip_search_bounded: chunk of %d queries needs %zu B of workspace, %zu given", nq, p.total, workspace_bytes);
      return LRX_ERR_WORKSPACE;
    }
    const float* qc = q + (int64_t)q0 * dim;
    float* osc = out_scores + (int64_t)q0 * k;
    int64_t* oic = out_ids + (int64_t)q0 * k;
    char* ws = (char*)workspace;
    float* scores = (float*)ws;
    __bf16* qsplit = (__bf16*)(ws + p.off_qsplit);
    int* flg = (int*)(ws + p.off_ints);
    int* any_flag = flg + nq;                 // [ANY_FLAG_GROUPS]: one per 128-query group of the chunk
    unsigned int* cnt = (unsigned int*)(flg + ints_before_cnt(nq));
    int* part_cnt = (int*)(cnt + (size_t)nq * CNT_STRIDE);
    float* thr = (float*)(part_cnt + (size_t)nq * REF_SPLIT);
    float* eps = thr + nq;
    unsigned long long* parts = (unsigned long long*)(ws + p.off_parts);
    unsigned long long* cand = (unsigned long long*)(ws + p.off_cand);
    // flags, any_flag, list counts start at zero: cleared by the query-packing kernel of the first filter launch when there is one
    const size_t nclear = ints_before_cnt(nq) + (p.emit ? (size_t)nq * CNT_STRIDE : 0);
    const bool clear_in_pack = shadow && nclear < (1u << 30);
    if (!clear_in_pack) LRX_HIP(hipMemsetAsync(flg, 0, sizeof(int) * nclear, s));
    // ... and so are the hi/mid/lo query planes of the gated six-product fallback (groups of <= 128 queries; <= 32 queries run the
    // exact-fp32 kernel, which needs none)
    __bf16* qs3 = (__bf16*)(ws + p.off_qs3);
    const int64_t qs3_stride = (int64_t)(split_ws_bytes(dim) / sizeof(__bf16));
    // (per 256-query sub-chunk j of the chunk: the plane sets of its two 128-query groups, qs3 + (2 j + g) * stride)
    auto presplit_of = [&](int j) {
      PreSplit ps;
      if (!shadow) return ps;
      const int n0 = j * 256, nj = nq - n0 < 256 ? nq - n0 : 256;
      ps.qs3 = qs3 + (int64_t)(2 * j) * qs3_stride;
      ps.stride = qs3_stride;
      for (int f0 = 0, g = 0; f0 < nj; f0 += 128, ++g) {
        const int nf = nj - f0 < 128 ? nj - f0 : 128, qtf = (nf + 15) / 16;
        ps.nf[g] = nf;
        ps.qt[g] = qtf;
        ps.blocks[g] = qtf >= SPLIT_MIN_QT ? ((dim / 32) * qtf * 64 + 255) / 256 : 0;
        ps.ngroups = g + 1;
      }
      if (ps.blocks[0] + ps.blocks[1] == 0) ps.ngroups = 0;
      return ps;
    };
    const PreSplit presplit = presplit_of(0);
    int rc;
    int nsplit = REF_SPLIT;                   // parts per query of the refine step
    if (p.emit) {
      float* blkmax = scores + p.ld_s * (int64_t)nq;
      FilterMode fs;                                          // (sample pass of the 128-row kernels; chunks on the GEMM kernel sample in 256-row tiles there)
      fs.bmode = 1; fs.ss = p.ss; fs.unit = 1; fs.nblocks = p.nsamp_wg;
      fs.group_max = shadow;                                  // the register-streaming kernels hand over the maxima of their 16-row wave groups
      if (clear_in_pack) { fs.zero = flg; fs.nzero = (int)nclear; }
      fs.presplit = presplit;
      FusedArgs fa;
      if (p.fused && clear_in_pack) {
        fa.ld_s = p.ld_s; fa.nblk_s = (int)p.nblk_s; fa.nblk_ld_s = (int)p.nblk_ld_s; fa.nsamp = (int)p.nsamp_wg; fa.nmain = (int)p.nmain_wg; fa.ss = p.ss; fa.k = k;
        fa.bounds = row_bounds; fa.thr = thr; fa.eps = eps; fa.cand = cand; fa.cnt = cnt; fa.cap = p.cap;
        fa.ctl = (FusedCtl*)(flg + ints_before_cnt(nq) - 8);
        // (sample + selection + main pass, always; LRX_FUSED_PHASES bit 7 in a dev build adds the per-workgroup phase stamps of lrx_probe_fused_timestamps)
        static const int fused_stamps = lrx_dev_knob("LRX_FUSED_PHASES", 0) & 128;
        fa.phases = 7 | fused_stamps;
        fs.fused = &fa;
      }
      __bf16* q16 = (__bf16*)(ws + p.off_q16);
      if (p.gemm) {
        // More than 128 queries over a shadow of D >= 1024: BOTH passes on the GEMM kernel (round 6; the 128-row register-streaming kernel needs
        // 170 us for one sample block at 16 query tiles).  The packing kernel still runs once per 256-query sub-chunk: it clears the ints of the
        // chunk (first launch) and writes the fallback planes of the sub-chunk's two 128-query groups.
        for (int j = 0, n0 = 0; n0 < nq; ++j, n0 += 256) {
          const int nj = nq - n0 < 256 ? nq - n0 : 256, qtj = (nj + 15) / 16;
          PreSplit ps = presplit_of(j);
          ps.nb_xb = ((dim / 64) * 2 * qtj * 64 + 255) / 256;
          hipLaunchKernelGGL(k_pack_queries_xb, dim3(ps.nb_xb + (ps.ngroups > 0 ? ps.blocks[0] + ps.blocks[1] : 0)), dim3(256), 0, s, qc + (int64_t)n0 * dim, nj, dim, qtj,
                             qsplit, j == 0 ? fs.zero : (int*)nullptr, j == 0 ? fs.nzero : 0, ps);
          LRX_LAUNCH_CHECK();
        }
        hipLaunchKernelGGL(k_round_queries, dim3((unsigned)lrx_cdiv((int64_t)nq * dim, 1024)), dim3(256), 0, s, qc, (int64_t)nq * dim, q16);
        LRX_LAUNCH_CHECK();
        rc = lrx_gemm_filter_sample_launch(X_shadow, q16, n_rows, nq, dim, p.ss, p.nsamp_wg, scores, blkmax, p.ld_s, (int)p.nblk_ld_s, s);
      } else {
        rc = launch_scores(X, n_rows, ldx, dim, qc, nq, scores, blkmax, qsplit, stream, 1, nullptr, shadow ? X_shadow : nullptr, p.ld_s, fs);
      }
      if (rc != LRX_OK) return rc;
      if (fs.fused == nullptr)
      hipLaunchKernelGGL(k_sample_threshold, dim3(nq), dim3(SEL_THREADS), 0, s, (const float*)scores, p.ld_s, p.nsamp_wg * p.rb, k, (const float*)blkmax,
                         (int)p.nblk_s, (int)p.nblk_ld_s, qc, dim, row_bounds, p.rb, p.ss, n_rows, thr, eps, cand, cnt, fs.group_max ? 16 : 128, p.cap);
      LRX_LAUNCH_CHECK();
      if (fs.fused != nullptr) {
        rc = LRX_OK;                                            // sample, selection and main pass are done
      } else if (p.gemm) {
        rc = lrx_gemm_filter_emit_launch(X_shadow, q16, n_rows, nq, dim, p.ss, p.nmain_wg, thr, cand, cnt, p.cap, s);
      } else {
        FilterMode fm;
        fm.bmode = 2; fm.ss = p.ss; fm.nblocks = p.nmain_wg; fm.thr = thr; fm.cand = cand; fm.cnt = cnt; fm.cap = p.cap;
        rc = launch_scores(X, n_rows, ldx, dim, qc, nq, nullptr, nullptr, qsplit, stream, 1, nullptr, shadow ? X_shadow : nullptr, p.ld, fm);
      }
      if (rc != LRX_OK) return rc;
      // parts per query: the 1024-thread workgroups of the refine step run one per CU, so REF_SPLIT x nq of them beyond the CU count take
      // a second round of ~40 us each (Q = 100: 400 workgroups, 82 us) -- fewer, larger parts then finish sooner
      // (Round 3, measured and not kept: with the narrow fp16 band (~130 rows at k = 100) ONE workgroup per query that also sorts and writes
      // the result, no merge launch: 0.723 / 0.739 vs 0.715 / 0.731 ms on the same box -- the serial select + rescore of one workgroup costs
      // more than the merge launch saves.)
      // (Round 3, measured and not kept: the number of parts, 1..8, that leaves the fewest CUs idle over whole rounds -- 5 for 100 queries:
      // top-1000 0.959 ms against 0.951 with 2 parts, top-100 0.730 against 0.719, 8 parts 1.00 / 0.747: the gather is bound chip-wide.)
      nsplit = (int64_t)nq * REF_SPLIT <= lrx_cu_count() ? REF_SPLIT : ((int64_t)nq * 2 <= lrx_cu_count() ? 2 : 1);
      // (Round 4, re-measured on the per-rank shard sizes, 100 queries: 125 k x 2048 0.190 / 0.178-0.181 / 0.190 / 0.191-0.192 ms for 1 / 2 / 3 / 4 parts,
      // 1.25M x 256 0.202 / 0.196-0.198 / 0.208 / 0.202-0.208, 1M x 2048 0.741 / 0.722-0.726 / 0.735 / 0.738-0.743: two parts everywhere.)
      // Row-grouped rescoring where the chunk wants every row more than twice on average (~1.2 k band rows per query): the pairs and the
      // group counters live in the score region, which nothing reads between the sample's selection and the gated fallback.
      RowPairs rp = {nullptr, nullptr, nullptr, 0};
      {
        const int gshift = dim <= 512 ? 5 : (dim <= 1024 ? 4 : (dim <= 2048 ? 3 : 2));
        const int64_t ngroups = (n_rows + ((int64_t)1 << gshift) - 1) >> gshift;
        const size_t pair_bytes = align256((size_t)nq * REF_CAND * 8);
        const size_t need = 2 * pair_bytes + align256((size_t)(ngroups + 2) * 4) * 2 + 256;
        const bool fits = need <= p.off_qsplit && ((int64_t)1 << gshift) * dim <= ROWGRP_LDS_FLOATS && dim % 4 == 0 && ldx % 4 == 0 && ngroups < (1 << 30);
        // measured (tools/exp/refine_rows_ab.py, profiles/r05_refine_rows_ab.txt; gather / by row, ms): 1000 queries, k = 1000 over 50 k x 2048 1.98 / 1.47,
        // 100 k x 2048 2.20 / 2.14, 100 k x 4096 4.50 / 4.02, but 100 k x 1024 1.46 / 1.57, 200 k x 2048 2.82 / 3.49, k = 100 0.81 / 1.51: rows of
        // >= 8 KiB that the chunk wants >= 2.5 times on average (~1.25 k band rows per query)
        const bool rule = dim >= 2048 && (int64_t)nq * k * 5 / 4 >= (5 * n_rows) / 2;
        if (fits && !(flags & 32) && (rule || (flags & 16))) {
          char* b = ws;
          rp.pairs = (unsigned long long*)b;
          unsigned long long* sorted = (unsigned long long*)(b + pair_bytes);
          rp.grp_cnt = (unsigned int*)(b + 2 * pair_bytes);
          unsigned int* grp_off = (unsigned int*)(b + 2 * pair_bytes + align256((size_t)(ngroups + 2) * 4));
          rp.total = grp_off + ngroups + 1;
          rp.grp_shift = gshift;
          LRX_HIP(hipMemsetAsync(rp.grp_cnt, 0, (size_t)((char*)(rp.total + 1) - (char*)rp.grp_cnt), s));
          hipLaunchKernelGGL(k_refine_band, dim3(nq, nsplit), dim3(1024), 0, s, X, n_rows, ldx, dim, qc, (const unsigned long long*)cand,
                             (const unsigned int*)cnt, (const float*)eps, k, parts, part_cnt, nsplit, p.cap, rp);
          LRX_LAUNCH_CHECK();
          hipLaunchKernelGGL(k_pairs_scan, dim3(1), dim3(1024), 0, s, (const unsigned int*)rp.grp_cnt, grp_off, (int)ngroups);
          LRX_LAUNCH_CHECK();
          hipLaunchKernelGGL(k_pairs_scatter, dim3((unsigned)(4 * lrx_cu_count())), dim3(256), 0, s, (const unsigned long long*)rp.pairs, (const unsigned int*)rp.total,
                             rp.grp_cnt, (const unsigned int*)grp_off, gshift, sorted);
          LRX_LAUNCH_CHECK();
          hipLaunchKernelGGL(k_rescore_row_groups, dim3((unsigned)ngroups), dim3(ROWGRP_THREADS), 0, s, X, n_rows, ldx, dim, qc, (const unsigned long long*)sorted,
                             (const unsigned int*)grp_off, gshift, parts);
          LRX_LAUNCH_CHECK();
        }
      }
      if (rp.pairs == nullptr) {
        hipLaunchKernelGGL(k_refine_band, dim3(nq, nsplit), dim3(1024), 0, s, X, n_rows, ldx, dim, qc, (const unsigned long long*)cand,
                           (const unsigned int*)cnt, (const float*)eps, k, parts, part_cnt, nsplit, p.cap, rp);
        LRX_LAUNCH_CHECK();
      }
    } else {
      float* blkmax = scores + p.ld * (int64_t)nq;
      FilterMode fa;
      if (clear_in_pack) { fa.zero = flg; fa.nzero = (int)nclear; }
      fa.presplit = presplit;
      rc = launch_scores(X, n_rows, ldx, dim, qc, nq, scores, blkmax, qsplit, stream, 1, nullptr, shadow ? X_shadow : nullptr, 0, fa);
      if (rc != LRX_OK) return rc;
      hipLaunchKernelGGL(k_topk_select, dim3(nq), dim3(SEL_THREADS), 0, s, (const float*)scores, p.ld, n_rows, k, id_base, (const float*)blkmax, (int)p.nblk,
                         (int)p.nblk_ld, osc, oic, (const int*)nullptr, (const int*)nullptr);
      LRX_LAUNCH_CHECK();
      hipLaunchKernelGGL(k_refine_topk, dim3(nq, REF_SPLIT), dim3(1024), 0, s, X, n_rows, ldx, dim, qc, (const float*)scores, p.ld, (const float*)blkmax,
                         (int)p.nblk, (int)p.nblk_ld, row_bounds, k, id_base, (const float*)osc, parts, part_cnt);
      LRX_LAUNCH_CHECK();
    }
    // (1024 threads: with 256 the sort of the typical 300-500 band rows took 20 instead of 14 us)
    hipLaunchKernelGGL(k_refine_merge, dim3(nq), dim3(1024), 0, s, (const unsigned long long*)parts, (const int*)part_cnt, n_rows, k, id_base, osc, oic,
                       flg, any_flag, nsplit);
    LRX_LAUNCH_CHECK();
    // gated fallback for the flagged queries (every kernel returns immediately on the device when nothing overflowed), 128 queries at
    // a time over the same score region
    for (int f0 = 0; f0 < nq; f0 += 128) {
      const int nf = nq - f0 < 128 ? nq - f0 : 128;
      float* blkmax = scores + p.ld * (int64_t)nf;
      FilterMode ff;
      const PreSplit psf = presplit_of(f0 / 256);
      ff.planes_ready = psf.ngroups > 0;                      // written by the packing kernel(s) at the head of the chain
      const int* gate = any_flag + f0 / 128;                  // this group's flag: raised by k_refine_merge when one of its queries overflowed
      rc = launch_scores(X, n_rows, ldx, dim, qc + (int64_t)f0 * dim, nf, scores, blkmax, ff.planes_ready ? qs3 + (f0 / 128) * qs3_stride : qsplit, stream, 3,
                         gate, nullptr, 0, ff);
      if (rc != LRX_OK) return rc;
      hipLaunchKernelGGL(k_topk_select_rescore, dim3(nf), dim3(SEL_THREADS), 0, s, (const float*)scores, p.ld, n_rows, k, id_base, (const float*)blkmax,
                         (int)p.nblk, (int)p.nblk_ld, X, ldx, dim, qc + (int64_t)f0 * dim, osc + (int64_t)f0 * k, oic + (int64_t)f0 * k,
                         gate, (const int*)(flg + f0), row_bounds,
                         out_wire != nullptr ? (unsigned long long*)out_wire + ((int64_t)q0 + f0) * k : (unsigned long long*)nullptr, row_map);
      LRX_LAUNCH_CHECK();
    }
  }
  return LRX_OK;
}

// Statistics of the LAST bounded search that used `workspace` (tools, bench legs, tests): the number of candidate-list entries each query
// of the last chunk ended up with -- the rows that passed the filter threshold and reached the refine step.  Same (n_rows, dim,
// n_queries <= 256 (128 without a shadow), k, flags, has_shadow) as that search; zeros when it ran the score-matrix filter.
__global__ void k_copy_list_counts(const unsigned int* __restrict__ cnt, int nq, unsigned int* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nq) out[i] = cnt == nullptr ? 0u : cnt[(size_t)i * CNT_STRIDE];
}
extern "C" int lrx_flat_ip_bounded_list_counts(const void* workspace, int64_t n_rows, int32_t dim, int32_t n_queries, int32_t k, int32_t flags,
                                               int32_t has_shadow, uint32_t* counts_out, void* stream) {
  const bool shadow = has_shadow && dim % 64 == 0;
  LRX_CHECK_ARG(workspace && counts_out && n_queries > 0 && n_queries <= (shadow ? LRX_EMIT_MAX_QUERIES : 128), "bounded_list_counts: one query chunk only (n_queries=%d)", n_queries);
  const BoundedPlan p = plan_chunk(n_rows, dim, n_queries, k, shadow, flags & 3, fused_pref_of(flags));
  const int* flg = (const int*)((const char*)workspace + p.off_ints);
  // (the same test lrx_flat_ip_search_bounded_wire uses to send a call down the plain path, which keeps no lists)
  const int qt_max = ((n_queries < 128 ? n_queries : 128) + 15) / 16;
  const bool plain = (qt_max < SPLIT_MIN_QT && !shadow) || n_rows <= REF_CAND || dim % 4 != 0;
  const unsigned int* cnt = p.emit && !plain ? (const unsigned int*)(flg + ints_before_cnt(n_queries)) : nullptr;
  hipLaunchKernelGGL(k_copy_list_counts, dim3((n_queries + 255) / 256), dim3(256), 0, (hipStream_t)stream, cnt, n_queries, counts_out);
  LRX_LAUNCH_CHECK();
  return LRX_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Shard maintenance for rows that did not come from the encoder's last kernel (FlatIPIndex.add, loaded index files): ONE read of the
// fp32 rows produces the fp16 shadow (RNE, saturating) and raises the two bounds {max |row|, max |row - fp16(row)|} (integer atomic max on
// the non-negative float patterns: order-independent).
// ---------------------------------------------------------------------------------------------------------------
// bounds only (a shard without a shadow, or a plain search that was not given its bounds): a wave walks 16 rows, one atomic pair per
// 64-row workgroup
__global__ void __launch_bounds__(256)
k_shard_bounds(const float* __restrict__ X, int64_t ldx, int64_t n_rows, int D, float* __restrict__ bounds) {
  __shared__ float s_r[4], s_e[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float rmax = 0.f, emax = 0.f;
  for (int j = 0; j < 16; ++j) {
    const int64_t r = (int64_t)blockIdx.x * 64 + wave * 16 + j;
    if (r >= n_rows) break;
    const float* x = X + r * ldx;
    float r2 = 0.f, e2 = 0.f;
    for (int i = lane * 4; i < D; i += 256) {
      const f32x4 v = __builtin_nontemporal_load((const f32x4*)(x + i));
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float d = v[e] - (float)f2h_sat(v[e]);
        r2 += v[e] * v[e];
        e2 += d * d;
      }
    }
    rmax = fmaxf(rmax, wave_sum(r2));
    emax = fmaxf(emax, wave_sum(e2));
  }
  if (lane == 0) { s_r[wave] = rmax; s_e[wave] = emax; }
  __syncthreads();
  if (threadIdx.x == 0) {
    rmax = fmaxf(fmaxf(s_r[0], s_r[1]), fmaxf(s_r[2], s_r[3]));
    emax = fmaxf(fmaxf(s_e[0], s_e[1]), fmaxf(s_e[2], s_e[3]));
    atomicMax((int*)bounds, __float_as_int(sqrtf(rmax) * (1.0f + 1e-6f)));
    atomicMax((int*)bounds + 1, __float_as_int(sqrtf(emax) * (1.0f + 1e-6f)));
  }
}
static int shard_bounds_launch(const float* X, int64_t ldx, int64_t n_rows, int32_t dim, float* row_bounds, hipStream_t s) {
  if (n_rows <= 0) return LRX_OK;
  hipLaunchKernelGGL(k_shard_bounds, dim3((unsigned)lrx_cdiv(n_rows, 64)), dim3(256), 0, s, X, ldx, n_rows, dim, row_bounds);
  LRX_LAUNCH_CHECK();
  return LRX_OK;
}

// Shadow + bounds: a wave owns a 16-row group of the tile grid, lane (fi, fq) reads row fi's 8 floats of k-step (slice, ks, fq) -- four
// lanes cover one 128-B line of the row -- and writes its 16-B piece of the fragment-major tile: 1 KiB contiguous per wave instruction.
__global__ void __launch_bounds__(256)
k_shard_rows_tiled(const float* __restrict__ X, int64_t ldx, int64_t n_rows, int D, __bf16* __restrict__ Xb, int64_t row0, float* __restrict__ bounds) {
  __shared__ float s_r[4], s_e[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fi = lane & 15, fq = lane >> 4;
  const int64_t g = (row0 >> 4) + (int64_t)blockIdx.x * 4 + wave;       // 16-row group of the tile grid
  const int64_t ra = g * 16 + fi, r = ra - row0;                         // row in the tiled array / in X
  const bool live = r >= 0 && r < n_rows;
  const float* x = X + (live ? r : 0) * ldx + fq * 8;
  __bf16* xb = Xb + ((ra >> 7) * (int64_t)(D / 64)) * 8192 + (((ra >> 4) & 7) * 2) * 512 + lane * 8;   // + slice * 8192 + ks * 512
  float r2 = 0.f, e2 = 0.f;
  const int nst = D / 32;                                                // k-steps of 32
  for (int st0 = 0; st0 < nst; st0 += 4) {
    f32x4 v[4][2];
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (st0 + u < nst && live) {
        v[u][0] = __builtin_nontemporal_load((const f32x4*)(x + (st0 + u) * 32));
        v[u][1] = __builtin_nontemporal_load((const f32x4*)(x + (st0 + u) * 32 + 4));
      }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (st0 + u < nst && live) {
        bf16x8 h;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float f = v[u][e >> 2][e & 3];
          const _Float16 hh = f2h_sat(f);
          h[e] = __builtin_bit_cast(__bf16, hh);
          const float d = f - (float)hh;
          r2 += f * f;
          e2 += d * d;
        }
        const int st = st0 + u;
        *(bf16x8*)(xb + (int64_t)(st >> 1) * 8192 + (st & 1) * 512) = h;
      }
  }
  r2 += __shfl_xor(r2, 16, 64); r2 += __shfl_xor(r2, 32, 64);          // the four lanes of a row
  e2 += __shfl_xor(e2, 16, 64); e2 += __shfl_xor(e2, 32, 64);
  const float rmax = wave_max(live ? r2 : 0.f), emax = wave_max(live ? e2 : 0.f);
  if (lane == 0) { s_r[wave] = rmax; s_e[wave] = emax; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float rm = fmaxf(fmaxf(s_r[0], s_r[1]), fmaxf(s_r[2], s_r[3])), em = fmaxf(fmaxf(s_e[0], s_e[1]), fmaxf(s_e[2], s_e[3]));
    atomicMax((int*)bounds, __float_as_int(sqrtf(rm) * (1.0f + 1e-6f)));
    atomicMax((int*)bounds + 1, __float_as_int(sqrtf(em) * (1.0f + 1e-6f)));
  }
}

extern "C" int lrx_shard_commit_rows(const float* X, int64_t ldx, int64_t n_rows, int32_t dim, void* X_shadow, int64_t shadow_row0, float* row_bounds,
                                     void* stream) {
  LRX_CHECK_ARG(dim > 0 && dim % 4 == 0 && ldx >= dim && ldx % 4 == 0, "shard_commit_rows: dim=%d / ldx=%lld must be multiples of 4", dim, (long long)ldx);
  LRX_CHECK_ARG(X_shadow == nullptr || (dim % 64 == 0 && shadow_row0 >= 0), "shard_commit_rows: the tiled shadow needs dim %% 64 == 0 (dim %d) and row0 >= 0", dim);
  LRX_CHECK_ARG(row_bounds != nullptr, "shard_commit_rows: null row_bounds");
  if (n_rows <= 0) return LRX_OK;
  if (X_shadow == nullptr) return shard_bounds_launch(X, ldx, n_rows, dim, row_bounds, (hipStream_t)stream);
  const int64_t groups = ((shadow_row0 + n_rows + 15) >> 4) - (shadow_row0 >> 4);
  hipLaunchKernelGGL(k_shard_rows_tiled, dim3((unsigned)lrx_cdiv(groups, 4)), dim3(256), 0, (hipStream_t)stream, X, ldx, n_rows, dim, (__bf16*)X_shadow,
                     shadow_row0, row_bounds);
  LRX_LAUNCH_CHECK();
  return LRX_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// merge of R shard-local top-k lists
// ---------------------------------------------------------------------------------------------------------------
#define MERGE_MAX 16384   // R * k: one LDS-resident bitonic merge per query (128 KiB at the maximum: 8 shards x k = 2048)
// in_packed != NULL: the lists arrive as one 64-bit word per hit (fp32 score bits << 32 | row as uint32, row 0xFFFFFFFF = none) --
// the form that crosses the all-gather (lrx_pack_topk)
__global__ void __launch_bounds__(1024)
k_merge_topk(const float* __restrict__ in_scores, const int64_t* __restrict__ in_ids, const unsigned long long* __restrict__ in_packed, int R, int Q,
             int k, float* __restrict__ out_scores, int64_t* __restrict__ out_ids) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  unsigned long long* buf = (unsigned long long*)smem_raw;
  const int qi = blockIdx.x, n = R * k;
  int P = 1;
  while (P < n) P <<= 1;
  for (int i = threadIdx.x; i < P; i += blockDim.x) {
    unsigned long long c = 0ull;
    if (i < n) {
      int rr = i / k, j = i - rr * k;
      int64_t src = ((int64_t)rr * Q + qi) * k + j;
      if (in_packed != nullptr) {
        const unsigned long long w = in_packed[src];
        if ((uint32_t)w != 0xFFFFFFFFu) c = ((unsigned long long)f2key(__uint_as_float((uint32_t)(w >> 32))) << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)w);
      } else {
        int64_t id = in_ids[src];
        if (id >= 0) c = ((unsigned long long)f2key(in_scores[src]) << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)id);
      }
    }
    buf[i] = c;
  }
  // ---- R lists that are each already in order (what every search of this library returns: score descending, row ascending -- also after a
  //      monotonic row map) need no sort: an entry's rank in the union is its position in its own list plus, for every other list, the number of
  //      that list's entries in front of it -- a binary search each (words are distinct: a row belongs to one shard), no barrier-separated
  //      stages.  8 x top-100: 13.6 -> ~8 us, 8 x top-1000: 96 -> ~25 us.  Checked here, not assumed: one pass over adjacent pairs; any list out
  //      of order (exact ties re-ordered by a non-monotonic row map, a foreign caller) sends the query to the sort below.
  {                                                             // (R = 1: an ordered list is its own merge)
    int bad = 0;
    __syncthreads();                                            // buf is complete: a thread looks at its neighbour's entry next
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
      const int j = i % k;
      if (j + 1 < k && buf[i] < buf[i + 1]) bad = 1;           // (padding words are 0: they sit at a list's end)
      if (j + 1 < k && buf[i] != 0ull && buf[i] == buf[i + 1]) bad = 1;
    }
    if (!__syncthreads_or(bad)) {
      for (int i = threadIdx.x; i < k; i += blockDim.x) { out_scores[(int64_t)qi * k + i] = -FLT_MAX; out_ids[(int64_t)qi * k + i] = -1; }
      __syncthreads();
      for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const unsigned long long x = buf[i];
        if (x == 0ull) continue;
        const int r = i / k;
        int rank = i - r * k;
        for (int o = 0; o < R; ++o) {
          if (o == r) continue;
          const unsigned long long* l = buf + o * k;            // descending; count the entries in front of x: > x, and == x in an earlier list
          int lo = 0, hi = k;                                   // (equal words = the same row handed in twice: both come out, as from the sort)
          while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            const unsigned long long y = l[mid];
            if (y > x || (y == x && o < r)) lo = mid + 1; else hi = mid;
          }
          rank += lo;
          if (rank >= k) break;
        }
        if (rank < k) {
          out_scores[(int64_t)qi * k + rank] = key2f((uint32_t)(x >> 32));
          out_ids[(int64_t)qi * k + rank] = sel_row(x);
        }
      }
      return;
    }
  }
  // (the launch picks blockDim.x = P / E, E = 2 .. 16, for P >= 128: the sort runs in registers; smaller P: one wave's worth, the LDS form)
  switch (P >= 128 ? P / (int)blockDim.x : 0) {
    case 2: bitonic_sort_desc_regs<2>(buf, P); break;
    case 4: bitonic_sort_desc_regs<4>(buf, P); break;
    case 8: bitonic_sort_desc_regs<8>(buf, P); break;
    case 16: bitonic_sort_desc_regs<16>(buf, P); break;
    default: bitonic_sort_desc(buf, P);
  }
  for (int i = threadIdx.x; i < k; i += blockDim.x) {
    unsigned long long c = i < P ? buf[i] : 0ull;
    int64_t o = (int64_t)qi * k + i;
    if (c == 0ull) { out_scores[o] = -FLT_MAX; out_ids[o] = -1; }
    else { out_scores[o] = key2f((uint32_t)(c >> 32)); out_ids[o] = sel_row(c); }
  }
}

static int merge_launch(const float* in_scores, const int64_t* in_ids, const unsigned long long* in_packed, int32_t n_parts, int32_t n_queries, int32_t k,
                        float* out_scores, int64_t* out_ids, void* stream) {
  LRX_CHECK_ARG(n_parts > 0 && k > 0 && (int64_t)n_parts * k <= MERGE_MAX, "merge_topk: parts*k=%lld exceeds %d", (long long)n_parts * k, MERGE_MAX);
  if (n_queries <= 0) return LRX_OK;
  int P = 1;
  while (P < n_parts * k) P <<= 1;
  size_t smem = (size_t)P * 8;
  if (smem > 48 * 1024) {
    LRX_HIP(hipFuncSetAttribute((const void*)k_merge_topk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  }
  // P >= 128: P / E threads with E = 2 (P <= 2048), 4, 8, 16 entries per thread in registers; below: the LDS sort on 64 threads per 64 entries
  const int threads = P >= 128 ? (P <= 2048 ? P / 2 : 1024) : 64;
  hipLaunchKernelGGL(k_merge_topk, dim3(n_queries), dim3(threads), smem, (hipStream_t)stream, in_scores, in_ids, in_packed, n_parts, n_queries, k,
                     out_scores, out_ids);
  LRX_LAUNCH_CHECK();
  return LRX_OK;
}

extern "C" int lrx_merge_topk(const float* in_scores, const int64_t* in_ids, int32_t n_parts, int32_t n_queries, int32_t k,
                              float* out_scores, int64_t* out_ids, void* stream) {
  return merge_launch(in_scores, in_ids, nullptr, n_parts, n_queries, k, out_scores, out_ids, stream);
}

extern "C" int lrx_merge_topk_packed(const uint64_t* in_packed, int32_t n_parts, int32_t n_queries, int32_t k, float* out_scores, int64_t* out_ids,
                                     void* stream) {
  return merge_launch(nullptr, nullptr, (const unsigned long long*)in_packed, n_parts, n_queries, k, out_scores, out_ids, stream);
}

// (score, id) -> the 64-bit wire word of the exchange; row_map (optional): id = row_map[id - id_base] for id >= 0 (local shard row ->
// global row of the sorted corpus).  Global rows must fit 32 bits (< 2^32 - 1): checked by the caller that owns the numbering.
__global__ void k_pack_topk(const float* __restrict__ D, const int64_t* __restrict__ I, const int64_t* __restrict__ row_map, int64_t id_base, int64_t n,
                            unsigned long long* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int64_t id = I[i];
  if (id >= 0 && row_map != nullptr) id = row_map[id - id_base];
  out[i] = ((unsigned long long)__float_as_uint(D[i]) << 32) | (unsigned long long)(id >= 0 ? (uint32_t)id : 0xFFFFFFFFu);
}

extern "C" int lrx_pack_topk(const float* scores, const int64_t* ids, const int64_t* row_map, int64_t id_base, int64_t n, uint64_t* out_packed,
                             void* stream) {
  if (n <= 0) return LRX_OK;
  hipLaunchKernelGGL(k_pack_topk, dim3((unsigned)lrx_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, scores, ids, row_map, id_base, n,
                     (unsigned long long*)out_packed);
  LRX_LAUNCH_CHECK();
  return LRX_OK;
}
