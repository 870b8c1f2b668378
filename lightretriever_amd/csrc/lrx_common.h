// Shared device/host helpers for liblrx (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/lrx.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define LRX_WAVE 64

// bf16 <-> f32.  A plain cast lowers to v_cvt_pk_bf16_f32 (RNE, NaN-preserving) on gfx950.
__device__ __forceinline__ float bf2f(__bf16 v) { return (float)v; }
__device__ __forceinline__ __bf16 f2bf(float v) { return (__bf16)v; }
__device__ __forceinline__ float bfbits2f(uint32_t lo16) { return __uint_as_float(lo16 << 16); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---- candidate lists of the score-free search filter (lrx_search.hip; the 256-query filter pass lives in lrx_gemm.hip)
#define CAND_CAP_MIN 65536   // (16 Ki until round 4: a corpus stored cluster by cluster put single queries at 17 k hits, and ONE overflowing query costs the whole chunk the six-product pass) smallest per-query capacity of the filter pass's candidate list (score-free filter); grows with k (lrx_search.hip: plan_chunk)
#define CNT_STRIDE 64    // list fill counters sit 256 B apart: the reservations of different queries go to different memory channels

__device__ __forceinline__ uint32_t f2key(float f) {  // monotone: larger float -> larger key
  uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key2f(uint32_t k) {
  uint32_t u = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
  return __uint_as_float(u);
}
// (score key, row) packed so that an unsigned sort is (score desc, row asc); rows < 2^32 per shard
__device__ __forceinline__ unsigned long long sel_pack(uint32_t key, int64_t i) {
  return ((unsigned long long)key << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)i);
}
__device__ __forceinline__ int64_t sel_row(unsigned long long c) { return (int64_t)(0xFFFFFFFFu - (uint32_t)(c & 0xFFFFFFFFull)); }

// ---- the tiled fp16 shadow of an index shard (include/lrx.h, lrx_shard_commit_rows): [128-row block][64-wide k-slice] tiles of 16 KiB, each tile
// FRAGMENT-MAJOR: [16-row group w = 0..7][k-step ks = 0..1][lane = fq*16 + fi][8], the MFMA 16x16x32 A operand of rows 16w + fi, k = 32 ks + 8 fq .. + 7,
// so that a wave of the filter pass loads its fragment with one coalesced 1-KiB request.  Element (row r, column k), D % 64 == 0:
__host__ __device__ __forceinline__ int64_t lrx_shadow_off(int64_t r, int k, int D) {
  return ((r >> 7) * (int64_t)(D / 64) + (k >> 6)) * 8192 + ((((r >> 4) & 7) * 2 + ((k >> 5) & 1)) * 64 + ((k >> 3) & 3) * 16 + (r & 15)) * 8 + (k & 7);
}

// Dev knobs: the environment switches of the A/B tools (sample stride, tile-group sizes, fused-launch phases, ...) exist only in
// -DLRX_DEV_KNOBS builds (`python -m lightretriever_amd.build -DLRX_DEV_KNOBS --out=<variant>.so`, run with LRX_LIB_DEV_VARIANT=<variant>.so);
// the shipping liblrx.so reads NO environment variable (tests/test_abi.py checks its imports): its behaviour depends on its arguments alone.
#ifdef LRX_DEV_KNOBS
#include <stdlib.h>
static inline int lrx_dev_knob(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
#else
static inline int lrx_dev_knob(const char*, int dflt) { return dflt; }
#endif

// host-side error plumbing -----------------------------------------------------------------------------------
void lrx_set_error(const char* fmt, ...);
#define LRX_CHECK_ARG(cond, ...)          \
  do {                                    \
    if (!(cond)) {                        \
      lrx_set_error(__VA_ARGS__);         \
      return LRX_ERR_INVALID;             \
    }                                     \
  } while (0)
#define LRX_HIP(call)                                                                     \
  do {                                                                                    \
    hipError_t e_ = (call);                                                               \
    if (e_ != hipSuccess) {                                                               \
      lrx_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
      return LRX_ERR_HIP;                                                                 \
    }                                                                                     \
  } while (0)
#define LRX_LAUNCH_CHECK() LRX_HIP(hipGetLastError())

static inline int64_t lrx_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

#define LRX_EMIT_MAX_QUERIES 1024   // queries of one chunk of the bounded search over a shadow of D >= 1024 (four 256-query n-tiles per A tile)
// lrx_gemm.hip: the GEMM kernel as the search's filter pass for 129..LRX_EMIT_MAX_QUERIES queries: scores = Xb[rows, D] . q16[nq, D]^T (bf16 operands,
// fp32 accumulation), nothing stored, rows reaching thr[query] appended to the query's candidate list.  Covers the 256-row tiles
// that are not in the sample (every ss-th tile): n_tiles of them.
int lrx_gemm_filter_emit_launch(const void* Xb, const void* q16, int64_t n_rows, int nq, int dim, int ss, int64_t n_tiles,
                                const float* thr, unsigned long long* cand, unsigned int* cnt, unsigned int cap, hipStream_t stream);
// lrx_gemm.hip: ... and as its sample pass (n_tiles sample tiles: corpus tiles 0, ss, 2 ss, ...): fp32 scores [nq, ld_s] + 16-row-group maxima
int lrx_gemm_filter_sample_launch(const void* Xb, const void* q16, int64_t n_rows, int nq, int dim, int ss, int64_t n_tiles, float* scores, float* gmax,
                                  int64_t ld_s, int nblk_ld_s, hipStream_t stream);
// lrx_gemm.hip: the GEMM kernel with the segmented-maximum epilogue (used by lrx_sparse_max_aggregate)
int lrx_gemm_max_aggregate_launch(const void* A, const void* B, const void* bias, const int32_t* row_seg, float* out, int64_t ldo, int M, int N,
                                  int K, hipStream_t stream);

// lrx_elementwise.hip: dst[b] = src[cu_seqlens[b + 1] - 1] for 4-byte elements (positions, row scales of the last-token rows)
// fp16-operand forms of the encoder GEMMs (lrx_gemm.hip; lrx_encoder_config.precise_stream = 2).  The exported entry points are these with f16 = 0.
int lrx_gemm_nt_fused_ex(const void* A, const void* B, void* C, const void* bias, const void* resid, int32_t M, int32_t N, int32_t K, int32_t epilogue,
                         const float* rscale, float* ss_part, int f16, void* stream);
int lrx_gemm_qkv_rope_slice_ex(const void* A, const void* Wqkv, void* C, const void* bias, const int32_t* positions, const float* cos, const float* sin,
                               int32_t M, int32_t K, int32_t num_q_heads, int32_t num_kv_heads, int32_t head_dim, const float* rscale, int32_t head0,
                               int32_t n_heads, int f16, void* stream);
int lrx_gemm_nt_resid32_ex(const void* A, const void* B, float* x32, void* a16_out, const void* gamma, int32_t M, int32_t N, int32_t K, float* ss_part,
                           int f16, int out_f16, void* stream);
int lrx_attn_varlen_causal_items_ex(const void* qkv, const int32_t* cu_seqlens, const void* items, size_t items_bytes, int32_t n_seqs, int32_t total_tokens,
                                    int32_t max_seqlen, int32_t num_q_heads, int32_t num_kv_heads, int32_t head_dim, void* out, int32_t last_tile_only,
                                    int out_f16, void* stream);
int lrx_attn_varlen_causal_ex(const void* qkv, const int32_t* cu_seqlens, int32_t n_seqs, int32_t total_tokens, int32_t max_seqlen, int32_t num_q_heads,
                              int32_t num_kv_heads, int32_t head_dim, void* out, int32_t last_tile_only, int out_f16, void* stream);
int lrx_attn_prefix_suffix_ex(const void* qkv, const void* prefix_kv, int32_t n_seqs, int32_t suffix_len, int32_t prefix_len, int32_t num_q_heads,
                              int32_t num_kv_heads, int32_t head_dim, void* out, int out_f16, void* stream);
int lrx_embed_stream32_ex(const void* table, const int32_t* ids, int32_t n_tokens, int32_t hidden, int32_t vocab, const void* gamma, float* x32, void* a16,
                          float* rscale, float eps, int f16, void* stream);
// lrx_elementwise.hip: the two-layer pooling strategies (column sums of the other hidden state; the pooling kernel with them)
int lrx_pool_sum_rows(const void* src, int src_kind, const int32_t* ids, int vocab, const int32_t* cu_seqlens, int n_seqs, int hidden_size, float* aux,
                      hipStream_t stream);
int lrx_pool_norm_aux(const void* hidden, const void* final_norm_w, const int32_t* cu_seqlens, int32_t n_seqs, int32_t hidden_size, float eps,
                      int32_t pooling, const float* aux, float* out, int64_t out_row_stride, int32_t out_dim, int32_t normalize, void* shadow_out,
                      int64_t shadow_row0, float* row_bounds, int32_t hidden_f32, void* stream);
int lrx_gather_rows_u32(const void* src, const int32_t* cu_seqlens, int32_t n_seqs, void* dst, hipStream_t stream);
