/*
 * lrx.h -- C ABI of liblrx.so: the MI355X (gfx950) corpus-embedding + flat-IP search hot path of LightRetriever.
 *
 * The reference (caskcsg/lightretriever) is pure Python and has no FFI of its own; its
 * per-batch operator boundary is HybridModel.encode_passage / encode_query and faiss.IndexFlatIP.  Each entry
 * point below names the reference interface (file:line under /root/reference) it replaces.  All pointers are
 * DEVICE pointers unless the name says `host`; all sizes are element counts unless the name says `bytes`;
 * `stream` is a hipStream_t passed as void* (NULL = the null stream).  No entry point allocates device memory,
 * synchronises the device or touches torch: the caller (PyTorch-ROCm in this build) owns memory and streams.
 * Every function returns LRX_OK (0) or a negative error; lrx_last_error() gives the message (thread-local).
 */
#ifndef LRX_H
#define LRX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LRX_ABI_VERSION 8

enum {
  LRX_OK = 0,
  LRX_ERR_INVALID = -1,   /* bad argument / unsupported shape */
  LRX_ERR_HIP = -2,       /* a HIP runtime call failed */
  LRX_ERR_WORKSPACE = -3, /* workspace too small */
};

int lrx_abi_version(void);
const char* lrx_last_error(void);

/* ------------------------------------------------------------------------------------------------------------
 * Encoder model description: the subset of HF config.json that decides the dense-path arithmetic
 * (transformers LlamaConfig / Qwen2Config; reference loads it at finetune/modeling_encoder.py:602-633).
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct lrx_encoder_config {
  int32_t vocab_size;
  int32_t hidden_size;        /* H; multiple of 64 */
  int32_t num_layers;
  int32_t num_q_heads;
  int32_t num_kv_heads;
  int32_t head_dim;           /* 64 or 128 */
  int32_t intermediate_size;  /* multiple of 64 */
  float rms_eps;
  int32_t qkv_bias;           /* 1 for Qwen2.5 */
  int32_t max_positions;      /* rows of the RoPE table */
  int32_t norm_folded;        /* 1: the caller pre-multiplied the RMSNorm weights into the next projection's columns (wqkv := W_qkv diag(ln1),
                                 wgu := W_gu diag(ln2), fp32 product rounded to bf16); the layers then never materialise the normalised
                                 activations: the row statistic rsqrt(mean(x^2)+eps) is applied to the GEMM's fp32 accumulator and the
                                 residual GEMMs emit the sum of squares of the rows they write.  Same function; the two bf16 roundings of
                                 the normalised activations disappear (closer to the fp32 model).  ln1 / ln2 are ignored.            */
  int32_t precise_stream;     /* 1 (needs norm_folded = 0 and the ORIGINAL wqkv / wgu): the residual stream is kept in fp32; every residual GEMM
                                 adds into it with one rounding (to fp32) and also writes the next projection's bf16 operand bf16(x * gamma_next)
                                 -- the norm weight rides on the activation, the weights stay exact, the row statistic is applied to the consumer's
                                 accumulator as with norm_folded.  For deep backbones: at 32 layers the bf16 stream + folded weights spend ~1.1e-3
                                 of the 1e-3 cosine budget against the fp32 model (tools/exp/rounding_budget.py); +6 B / element of traffic
                                 per residual GEMM (~3 % of a step).  LrxEncoder switches it on for EVERY backbone (since round 5: the mode that
                                 holds 1e-3 on trained-like weights; EncoderConfig(precise_stream=False) selects the bf16 stream).
                                 2: the same with FP16 GEMM operands (round 6; on request): wqkv / wo / wgu / wdown hold fp16 values
                                 (the caller converts the bf16 checkpoint once: exact for every |w| in [6.1e-5, 65504]; biases, norm weights
                                 and the embedding stay bf16), the activations fp16(x * gamma), the attention and SwiGLU outputs travel as fp16
                                 and every projection runs on the f16 MFMA (same rate).  Three more mantissa bits per operand: 1 - cos against
                                 the fp32 model drops 14-42 x (8B, trained-like weights: 3.4e-4 -> 8e-6, tools/exp/rounding_fp16_o_act.py).
                                 Values outside fp16's range saturate and count (lrx_device_saturation_count).  -3.5 % docs/s (f16 MFMA power).
                                 3: the QKV projection only (LrxEncoder's default): wqkv as fp16, its A operand fp16(x * gamma) -- the one rounding
                                 that is 70 % of mode 1's distance to the fp32 model (8B, worst of 2 048 documents: 1.11e-3 -> 2.4e-4) for -0.3 %;
                                 wo / wgu / wdown stay bf16.                                                                             */
} lrx_encoder_config;

/* Per-layer weights, bf16 (precise_stream = 2: wqkv / wo / wgu / wdown as fp16; = 3: wqkv as fp16), nn.Linear layout [out, in] row-major (K contiguous).
 *   wqkv  [(nq + 2 nkv) * d, H]   rows = q_proj | k_proj | v_proj concatenated, the d rows of every q and k head in ROTARY-PAIR order:
 *                                 physical row 32 g + 16 i + t of a head = logical row i * d/2 + 16 g + t  (g < d/32, i < 2, t < 16), so that
 *                                 the fused QKV epilogue finds x_j and x_{j + d/2} in one lane and rotates on the fp32 accumulators; q and k
 *                                 leave the projection in the same order (q . k is invariant), v rows are in logical order
 *   wo    [H, nq * d]
 *   wgu   [2 * I, H]              gate/up interleaved in 16-row groups: rows [32j,32j+16) = gate[16j..], [32j+16,32j+32) = up[16j..]
 *   wdown [H, I]
 *   ln1, ln2 [H] bf16 (input_layernorm, post_attention_layernorm); bqkv [(nq+2nkv)*d] bf16 (same row order as wqkv) or NULL */
typedef struct lrx_layer_weights {
  const void* wqkv;
  const void* bqkv;
  const void* wo;
  const void* wgu;
  const void* wdown;
  const void* ln1;
  const void* ln2;
} lrx_layer_weights;

typedef struct lrx_encoder_weights {
  const void* embed;             /* [V, H] bf16 */
  const void* final_norm;        /* [H] bf16 */
  const float* rope_cos;         /* [max_positions, d/2] fp32: cos(position * inv_freq), the values LlamaRotaryEmbedding computes in fp32 */
  const float* rope_sin;
  const lrx_layer_weights* layers; /* HOST array of num_layers structs (device pointers inside) */
} lrx_encoder_weights;

/* What torch.ops.lrx.encode_packed takes as its `weights` argument: the address of one of these (both structs owned, and kept
 * alive, by whoever built them -- LrxEncoder.handle on the Python side).                                                       */
typedef struct lrx_encoder_handle {
  const lrx_encoder_config* cfg;
  const lrx_encoder_weights* w;
} lrx_encoder_handle;

/* Workspace (device bytes) needed by lrx_encode_packed for `total_tokens` packed tokens in `n_seqs` sequences. */
size_t lrx_encode_workspace_bytes(const lrx_encoder_config* cfg, int32_t total_tokens, int32_t n_seqs);

/* Document-side encoder forward on a packed batch.
 * Replaces: HybridModel.encode_passage dense branch (finetune/modeling_hybrid.py:205-278) including the HF
 * LlamaModel/Qwen2Model forward it calls (:248-260), utils/nested_input.py:114-166 (packing: the packed layout is
 * native here), finetune/dense_pooling.py:48-55 ('lasttoken'), the MRL slice and F.normalize (:272-278).
 *   ids        [total_tokens] int32 token ids, sequences back to back
 *   cu_seqlens [n_seqs + 1] int32, cu_seqlens[0] = 0, cu_seqlens[n_seqs] = total_tokens; every sequence non-empty
 *   out        fp32 rows, row b at out + b * out_row_stride, out_dim (<= H) floats each: the L2-normalised
 *              (if normalize != 0) last-token embedding, first out_dim dims (dense_shrink_dim / MRL).  May point
 *              directly into the index shard (the encoder writes in place; no host round trip).
 * max_seqlen: upper bound on any sequence length (<= cfg->max_positions).                                       */
int lrx_encode_packed(const lrx_encoder_config* cfg, const lrx_encoder_weights* w, const int32_t* ids,
                      const int32_t* cu_seqlens, int32_t n_seqs, int32_t total_tokens, int32_t max_seqlen,
                      float* out, int64_t out_row_stride, int32_t out_dim, int32_t normalize, void* workspace,
                      size_t workspace_bytes, void* stream);
/* The same with `out` = rows of an index shard: the last kernel also writes the rows' fp16 shadow (shadow_out = base of the shard's tiled
 * shadow, `out` row b = its row shadow_row0 + b: layout at lrx_shard_commit_rows; NULL = none; needs out_dim % 64 == 0) and raises the shard's
 * bounds (row_bounds, see lrx_shard_commit_rows; NULL = none), so the index needs no second pass over what the encoder just wrote
 * (FaissIndex.build's add, retriever/faiss_index.py:45-58).                                                                       */
int lrx_encode_packed_shard(const lrx_encoder_config* cfg, const lrx_encoder_weights* w, const int32_t* ids,
                            const int32_t* cu_seqlens, int32_t n_seqs, int32_t total_tokens, int32_t max_seqlen,
                            float* out, int64_t out_row_stride, int32_t out_dim, int32_t normalize, void* shadow_out,
                            int64_t shadow_row0, float* row_bounds, void* workspace, size_t workspace_bytes, void* stream);

/* Same forward, but returns the un-pooled final hidden states (after the final RMSNorm), bf16 [total_tokens, H]:
 * the `last_hidden_state` of lm(...) at finetune/modeling_hybrid.py:260.  Used by EmbeddingBag construction
 * (finetune/nonctx_emb_utils.py:296-306 takes last_hidden_state[:, -1]) and by tests.                           */
int lrx_encode_hidden(const lrx_encoder_config* cfg, const lrx_encoder_weights* w, const int32_t* ids,
                      const int32_t* cu_seqlens, int32_t n_seqs, int32_t total_tokens, int32_t max_seqlen,
                      void* hidden_out_bf16, void* workspace, size_t workspace_bytes, void* stream);

/* Shared-prefix encode.  Replaces the inner loop of construct_embedding_bag (finetune/nonctx_emb_utils.py:262-306: for every
 * vocabulary token run the LM on [bos] + prompt + [tok] + [eos] and keep last_hidden_state[:, -1]): all n_seqs sequences share
 * the same prefix_ids [prefix_len]; sequence i continues with suffix_ids[i*suffix_len .. +suffix_len).  The prefix is encoded
 * once (its per-layer K/V are captured), only the suffix tokens run through the layers.  out: fp32 rows [n_seqs, out_dim], the
 * final-norm hidden state of each sequence's LAST token (normalize = 0 for the EmbeddingBag table).                          */
size_t lrx_encode_prefixed_workspace_bytes(const lrx_encoder_config* cfg, int32_t prefix_len, int32_t n_seqs, int32_t suffix_len);
int lrx_encode_prefixed(const lrx_encoder_config* cfg, const lrx_encoder_weights* w, const int32_t* prefix_ids,
                        int32_t prefix_len, const int32_t* suffix_ids, int32_t n_seqs, int32_t suffix_len, float* out,
                        int64_t out_row_stride, int32_t out_dim, int32_t normalize, void* workspace,
                        size_t workspace_bytes, void* stream);

/* (ABI 8) The same with the pooling strategy of finetune/dense_pooling.py:12-82 as an argument (`--pooling_strategy`; the released models use
 * 'lasttoken', which is what the two entry points above compute): */
#define LRX_POOL_LASTTOKEN 0       /* the sequence's last token (dense_pooling.py:48-55) */
#define LRX_POOL_CLS 1             /* its first token (:32-33) */
#define LRX_POOL_MEAN 2            /* mean of the final-norm rows of all its tokens (:35-36), fp32, tokens added in order */
#define LRX_POOL_SECOND_TO_LAST 3  /* token len - 2 (:57-67) */
#define LRX_POOL_THIRD_TO_LAST 4   /* token len - 3 (:69-79) */
#define LRX_POOL_AVG_FIRST_LAST 5  /* mean over the tokens of (hidden_states[0] + hidden_states[-1]) / 2: the embedding rows and the final-norm rows (:38-41) */
#define LRX_POOL_AVG_TOP2 6        /* ... of (hidden_states[-2] + hidden_states[-1]) / 2: the stream before the final layer and the final-norm rows (:43-46) */
/* LASTTOKEN runs the final layer's O-projection / MLP on the pooled rows only; every other strategy runs all layers over all tokens and pools
 * from the residual stream (final norm inside the pooling kernel, fp32 when the stream is).  A sequence shorter than its strategy needs (the
 * reference asserts there) gets a zero row and raises lrx_device_error_count.  The two-layer strategies add one column-sum pass over the other
 * hidden state (the embedding rows; the stream as it enters the final layer) -- served here only: lrx_pool_norm_mode sees one hidden state. */
int lrx_encode_packed_pooled(const lrx_encoder_config* cfg, const lrx_encoder_weights* w, const int32_t* ids,
                             const int32_t* cu_seqlens, int32_t n_seqs, int32_t total_tokens, int32_t max_seqlen, int32_t pooling,
                             float* out, int64_t out_row_stride, int32_t out_dim, int32_t normalize, void* shadow_out,
                             int64_t shadow_row0, float* row_bounds, void* workspace, size_t workspace_bytes, void* stream);

/* Dense + sparse document vectors in one pass.  Replaces HybridModel.encode_passage with encode_sparse
 * (finetune/modeling_hybrid.py:248-323): LM forward -> dense_reps (as lrx_encode_packed; dense_out may be NULL) and
 * sparse_reps = sparsify(max over the tokens tok_mask selects of hidden_t . lm_head^T)  (aggregate, sparse_pooling.py:244-278;
 * MaxLinearMapperFunction, utils/max_linear_map.py:8-88; get_sparse_emb, modeling_hybrid.py:176-203).
 * lm_head: bf16 [vocab, hidden] (NULL = tied to the embedding matrix), lm_head_bias: bf16 [vocab] or NULL.
 * tok_mask: uint8 [total_tokens] = get_sparse_attention_mask at the valid tokens (NULL = drop each sequence's first and last
 * token).  sparse_out: fp32 [n_seqs, sparse_row_stride >= vocab]; a sequence without any selected token gets finfo(bf16).min
 * before relu, like the reference.  round_bf16 = 1: log1p's result is rounded to bf16 (the tensor is bf16 in a bf16 run).   */
int lrx_encode_packed_sparse(const lrx_encoder_config* cfg, const lrx_encoder_weights* w, const void* lm_head,
                             const void* lm_head_bias, const int32_t* ids, const int32_t* cu_seqlens,
                             const uint8_t* tok_mask, int32_t n_seqs, int32_t total_tokens, int32_t max_seqlen,
                             float* dense_out, int64_t dense_row_stride, int32_t dense_dim, int32_t normalize,
                             float* sparse_out, int64_t sparse_row_stride, int32_t relu, int32_t log1p, int32_t round_bf16,
                             int32_t top_k, int32_t min_tokens_to_keep, void* workspace, size_t workspace_bytes, void* stream);

/* Per-kernel-class timing of the lrx_encode_* calls made since lrx_set_profiling / the previous lrx_get_profile: HIP events
 * are recorded on `stream` around every launch of the selected classes; nothing synchronises until lrx_get_profile reads
 * them (it waits for the recorded events, accumulates, and resets).  lrx_set_profiling(0) off, (1) every class, otherwise
 * bit (c + 1) of the argument selects class c (e.g. 1 << 3 = only the gate-up GEMM).
 * classes: 0 gemm/store (qkv), 1 gemm/residual (o, down), 2 gemm/swiglu (gate-up), 3 attention, 4 rmsnorm, 5 rope,
 * 6 other (embedding gather, positions, pool), 7 gemm_maxagg (LM-head GEMM with the max-aggregation epilogue).  Arrays of LRX_PROF_CLASSES entries; flops are algorithmic
 * (2*M*N*K for GEMMs, 2*2*d*sum_s(s*(s+1)/2)*nq for causal attention), 0 for the memory-bound classes.            */
#define LRX_PROF_CLASSES 8
void lrx_set_profiling(int32_t enabled);
/* (ABI 5) Measurement aids.  lrx_trace_marker: an empty kernel named k_trace_marker<id> (id 0..3) on `stream` -- delimits a region of a
 * rocprofv3 kernel trace (tools/check_trace_clean.py).  lrx_probe_stream_read: n_workgroups workgroups stream `bytes` (a multiple of 16)
 * of `buf` with 16-B loads and xor everything into sink[n_workgroups] (device, uint32): bytes / its duration = the read rate this box
 * reaches, the in-run ceiling bench.py reports next to the 8 TB/s of the spec.                                                    */
int lrx_trace_marker(int32_t id, void* stream);
int lrx_probe_stream_read(const void* buf, size_t bytes, uint32_t* sink, int32_t n_workgroups, void* stream);
int lrx_get_profile(float* ms, double* flops, int32_t* launches);

/* ------------------------------------------------------------------------------------------------------------
 * Individual kernels (unit-tested one by one against the oracle; same arithmetic the fused path uses)
 * ---------------------------------------------------------------------------------------------------------- */

/* out[t,:] = table[ids[t],:]  (nn.Embedding inside LlamaModel.forward)  bf16.  An id outside [0, vocab) never reaches the table: its
 * row is zero-filled and a device-side counter is raised (lrx_device_error_count).                                               */
int lrx_embedding_gather(const void* table, const int32_t* ids, int32_t n_tokens, int32_t hidden, int32_t vocab, void* out,
                         void* stream);
/* Number of out-of-range token ids any embedding gather (stand-alone or inside lrx_encode_*) has met since the last reset, plus the
 * attention work lists whose builder ran out of room (lrx_attn_build_items: their launches then compute nothing); -1 if the read
 * failed.  Non-zero = the rows of those calls are not the model's.  SYNCHRONISES the device (a blocking copy): call it at a point where
 * the caller waits for results anyway (LrxExactSearchModel.encode does, once per encode call, and raises).                         */
int64_t lrx_device_error_count(int32_t reset);
/* (ABI 6) Measurement aid of DEV builds (-DLRX_DEV_KNOBS; the shipping library reads no environment variable and records nothing): with
 * LRX_FUSED_PHASES bit 7 set in the environment the fused filter launch of the bounded search (sample + selection
 * + main pass in one persistent kernel) records per-workgroup phase timestamps (100 MHz clock; 8 words per workgroup: start, sample done,
 * selection start, selection done, first main K loop done, thresholds seen, ..., end); this copies the first n_words of them to the host.     */
int lrx_probe_fused_timestamps(uint64_t* out, int32_t n_words);
/* (ABI 5) fp16 range events since the last reset: q|k|v elements of the fused QKV + RoPE epilogue and fp16-shadow elements of pooled rows
 * that were NaN or beyond +-65504 and were stored as a finite +-65504 (counted once per wave instruction that saw any, so "0 or not" is
 * the meaningful reading).  0 on every checkpoint whose q|k|v stay inside fp16's range -- the precondition of the fp16 attention path;
 * a non-zero count means the embeddings of that call are not the model's.  -1 if the read failed.  SYNCHRONISES like the call above. */
int64_t lrx_device_saturation_count(int32_t reset);

/* LlamaRMSNorm (modeling_llama.py:53-67): y = w * bf16(x * rsqrt(mean(x^2) + eps)); x, w, y bf16; rows x hidden */
int lrx_rmsnorm(const void* x, const void* w, void* y, int32_t rows, int32_t hidden, float eps, void* stream);

/* C[M,N] = A[M,K] * B[N,K]^T, bf16 in, fp32 accumulate, bf16 out.  K % 64 == 0, N % 8 == 0.
 *   epilogue 0: C = acc (+ bias[n] if bias != NULL)          ldc = N
 *   epilogue 1: C = acc + resid[m,n]  (resid may alias C)    ldc = N
 *   epilogue 2: SwiGLU on gate/up-interleaved B (see wgu):   C[M, N/2] = silu(gate) * up, ldc = N/2           */
int lrx_gemm_bf16_nt(const void* A, const void* B, void* C, const void* bias, const void* resid, int32_t M,
                     int32_t N, int32_t K, int32_t epilogue, void* stream);

/* Fused QKV projection + rotary embedding: C[M, (nq+2nkv)*d] = A[M,K] * Wqkv^T (+ bias), then apply_rotary_pos_emb
 * (modeling_llama.py:130-160) on the q and k column blocks, on the fp32 accumulators with the fp32 cos/sin table (one rounding); v columns
 * are stored unrotated.  Wqkv / bias rows of the q and k heads in rotary-pair order (lrx_layer_weights); C is FP16 (saturating at
 * +-65504; 11 significant bits for both operands of q . k), q and k columns in the same pair order.                                */
int lrx_gemm_qkv_rope(const void* A, const void* Wqkv, void* C, const void* bias, const int32_t* positions,
                      const float* cos, const float* sin, int32_t M, int32_t K, int32_t num_q_heads,
                      int32_t num_kv_heads, int32_t head_dim, void* stream);

/* positions[t] = t - cu_seqlens[seq(t)] */
int lrx_build_positions(const int32_t* cu_seqlens, int32_t n_seqs, int32_t total_tokens, int32_t* positions, void* stream);

/* Varlen causal GQA attention (flash_attention_2 varlen call the reference requires for packing,
 * utils/nested_input.py:137-146): out[T, nq*d] bf16 = softmax(q k^T / sqrt(d), causal within each sequence) v.
 * qkv [T, (nq + 2 nkv) * d] FP16 as lrx_gemm_qkv_rope writes it (probabilities are rounded to fp16 too; fp32 accumulation).
 * last_tile_only != 0: only the 64-row q tile that holds each sequence's LAST token is computed (other rows of `out`
 * are left untouched) -- all the pooled path needs from the final layer.                                          */
int lrx_attn_varlen_causal(const void* qkv, const int32_t* cu_seqlens, int32_t n_seqs, int32_t total_tokens,
                           int32_t max_seqlen, int32_t num_q_heads, int32_t num_kv_heads, int32_t head_dim,
                           void* out, int32_t last_tile_only, void* stream);

/* The same attention on a prebuilt WORK LIST (ABI 7): which (sequence, kv head, q tile) each persistent workgroup computes, and in which
 * order, depends on cu_seqlens and the head layout only, so the encoder builds it once per batch and every layer's launch reads it
 * (lrx_encode_* do this inside their workspace).  Results are bit-identical to lrx_attn_varlen_causal; the launch is ~10 % shorter on
 * the tiled path (head_dim 128, or head_dim 64 beyond 512 tokens: no per-item index arithmetic).
 *   lrx_attn_items_bytes   size of the list; bounded by (total_tokens / 64 + n_seqs) x kv heads and non-decreasing in total_tokens and
 *                          max_seqlen (size once with the largest batch / longest sequence the caller will ever pass)
 *   lrx_attn_build_items   fills `items` (device memory, 16-byte aligned) on `stream`; one list per (cu_seqlens, max_seqlen, head
 *                          layout, last_tile_only) -- a list built with other arguments than the launch's gives wrong results.  Launches
 *                          nothing when the launch with the same arguments would not read a list (the K/V-resident geometries below).
 *                          At most 1024 persistent workgroups are planned (the builder is one block), whatever the CU count
 *   lrx_attn_varlen_causal_items   the launch (lrx_attn_varlen_causal's arguments + the list); `items` must stay untouched until it has
 *                          run.  Geometries served by the K/V-resident kernel (head_dim 64, max_seqlen <= 512) do not read the list.          */
size_t lrx_attn_items_bytes(int32_t n_seqs, int32_t total_tokens, int32_t max_seqlen, int32_t num_q_heads, int32_t num_kv_heads,
                            int32_t head_dim, int32_t last_tile_only);
int lrx_attn_build_items(const int32_t* cu_seqlens, int32_t n_seqs, int32_t total_tokens, int32_t max_seqlen, int32_t num_q_heads,
                         int32_t num_kv_heads, int32_t head_dim, int32_t last_tile_only, void* items, size_t items_bytes, void* stream);
int lrx_attn_varlen_causal_items(const void* qkv, const int32_t* cu_seqlens, const void* items, size_t items_bytes, int32_t n_seqs,
                                 int32_t total_tokens, int32_t max_seqlen, int32_t num_q_heads, int32_t num_kv_heads, int32_t head_dim,
                                 void* out, int32_t last_tile_only, void* stream);
/* Diagnostics: how many list builds (since the library was loaded) found their buffer too small -- 0 as long as the buffers come from
 * lrx_attn_items_bytes; such a build leaves an empty list (the launch then computes nothing).  Synchronises the device.                    */
int lrx_debug_attn_items_overflow(int32_t* count);

/* Attention of suffix queries over a shared prefix: qkv [n_seqs*suffix_len, (nq+2nkv)*d] (RoPE applied), prefix_kv
 * [prefix_len, 2*nkv*d] (k block | v block of one layer, RoPE applied); query j of a sequence sees the prefix keys and its own
 * suffix keys 0..j.  qkv and prefix_kv FP16 (lrx_gemm_qkv_rope output), out [n_seqs*suffix_len, nq*d] bf16.                    */
int lrx_attn_prefix_suffix(const void* qkv, const void* prefix_kv, int32_t n_seqs, int32_t suffix_len, int32_t prefix_len,
                           int32_t num_q_heads, int32_t num_kv_heads, int32_t head_dim, void* out, void* stream);

/* cu_seqlens[i] = i*len (i = 0..n_seqs), positions[t] = position_offset + t % len : equal-length batch layout built on device */
int lrx_uniform_layout(int32_t* cu_seqlens, int32_t* positions, int32_t n_seqs, int32_t len, int32_t position_offset, void* stream);

/* out[b, v] = max(finfo(bf16).min, max over tokens t of sequence b with tok_mask[t] of bf16(hidden[t,:] . lm_head[v,:] + bias[v])).
 * hidden bf16 [total_tokens, hidden_size] (hidden_size % 64 == 0), lm_head bf16 [vocab_size, hidden_size], out fp32
 * [n_seqs, out_row_stride]; row_seg_workspace: int32 [total_tokens] scratch.  The [tokens, vocab] logits are never written:
 * the maximum is taken per 256x256 tile in the GEMM epilogue and merged with integer atomics.                               */
int lrx_sparse_max_aggregate(const void* hidden, const void* lm_head, const void* bias, const int32_t* cu_seqlens,
                             const uint8_t* tok_mask, int32_t n_seqs, int32_t total_tokens, int32_t hidden_size,
                             int32_t vocab_size, float* out, int64_t out_row_stride, int32_t* row_seg_workspace, void* stream);

/* In place on fp32 [n_rows, row_stride]: relu -> log1p (-> bf16 rounding) -> top-k threshold (entries below the k-th largest
 * value of the row become 0, ties survive; k = max(top_k, min_tokens_to_keep); top_k = 0 disables)  (sparse_pooling.py:92-109). */
int lrx_sparsify(float* reps, int32_t n_rows, int32_t vocab_size, int64_t row_stride, int32_t relu, int32_t log1p,
                 int32_t round_bf16, int32_t top_k, int32_t min_tokens_to_keep, void* stream);

/* Quantise + compact (sparse_converter_mixin.py:129-137): per row the token ids whose round-half-even(max(x,0) * q) != 0, in
 * ascending id order, with that integer weight: ids_out / weights_out int32 [n_rows, capacity] (first `capacity` kept),
 * counts_out int32 [n_rows] = true number of non-zeros.                                                                    */
int lrx_sparse_compact(const float* reps, int32_t n_rows, int32_t vocab_size, int64_t row_stride, int32_t quantization_factor,
                       int32_t capacity, int32_t* ids_out, int32_t* weights_out, int32_t* counts_out, void* stream);

/* Hit-list fusion (retriever/score_fuse_utils.py:3-91 on arrays; IEEE double like the reference's numpy float64).
 * Stage 1, one retrieval system: scores f64 / ids i64 [n_queries, k] (id < 0 = empty slot) -> contribution per entry:
 *   method 0 (fuse_scores_rrf):    1 / (param0 + rank), rank 1 = highest score of the row;
 *   method 1 (fuse_scores_linear): param0 * (s - min) / (max - min + param1), min/max over the row's valid entries.
 * Stage 2: the systems' (ids, contributions) concatenated per query in system order [n_queries, n_entries] -> union by id,
 * contributions of one id summed in system order, rows sorted by fused score (descending, lower id first among equals):
 * scores_out f64 / ids_out i64 [n_queries, n_entries] (-inf / -1 beyond counts_out[q]).  k, n_entries <= 4096.             */
int lrx_hit_contributions(const double* scores, const int64_t* ids, int32_t n_queries, int32_t k, int64_t row_stride,
                          int32_t method, double param0, double param1, double* contrib_out, int64_t contrib_row_stride,
                          void* stream);
int lrx_hit_union(const int64_t* ids, const double* contrib, int32_t n_queries, int32_t n_entries, int64_t row_stride,
                  double* scores_out, int64_t* ids_out, int32_t* counts_out, void* stream);

/* GEMMs with the folded-RMSNorm hooks: rscale [M] fp32 or NULL -- accumulator row m times rscale[m] before bias / RoPE / SwiGLU
 * (store and SwiGLU epilogues); ss_part [ceil(N/256), M] fp32 or NULL -- residual epilogue: sum of squares of the output row's
 * columns of each 256-wide n-tile, written (not accumulated) to slot [tile, m].                                                  */
int lrx_gemm_bf16_nt_fused(const void* A, const void* B, void* C, const void* bias, const void* resid, int32_t M, int32_t N,
                           int32_t K, int32_t epilogue, const float* rscale, float* ss_part, void* stream);
int lrx_gemm_qkv_rope_fused(const void* A, const void* Wqkv, void* C, const void* bias, const int32_t* positions,
                            const float* cos, const float* sin, int32_t M, int32_t K, int32_t num_q_heads,
                            int32_t num_kv_heads, int32_t head_dim, const float* rscale, void* stream);
/* (ABI 6) heads [head0, head0 + n_heads) of the fused projection -- numbered q_0 .. q_{nq-1}, k_0 .. k_{nkv-1}, v_0 .. v_{nkv-1} -- from that
 * row slice of Wqkv / bias into that column slice of C (rows of C keep the full (nq + 2 nkv) * d width).  lrx_encode_packed uses it in the
 * FINAL layer: k|v for every token, q only for the gathered last-token rows (the only q rows the pooled output depends on).            */
int lrx_gemm_qkv_rope_slice(const void* A, const void* Wqkv, void* C, const void* bias, const int32_t* positions,
                            const float* cos, const float* sin, int32_t M, int32_t K, int32_t num_q_heads,
                            int32_t num_kv_heads, int32_t head_dim, const float* rscale, int32_t head0, int32_t n_heads, void* stream);
/* Precise residual stream (lrx_encoder_config.precise_stream).  x32[M, N] (fp32, in place) += A[M, K] * B[N, K]^T; a16_out (bf16 [M, N],
 * may be NULL) = bf16(x32 * gamma[n]) (gamma bf16 [N], NULL = 1): the next projection's operand; ss_part as above, from the fp32 row.   */
int lrx_gemm_bf16_nt_resid32(const void* A, const void* B, float* x32, void* a16_out, const void* gamma, int32_t M, int32_t N,
                             int32_t K, float* ss_part, void* stream);
/* ... its start: x32[t, :] = table[ids[t], :], a16[t, :] = bf16(x32 * gamma), rscale_out[t] = rsqrt(mean(x32^2) + eps)              */
int lrx_embed_stream32(const void* table, const int32_t* ids, int32_t n_tokens, int32_t hidden, int32_t vocab, const void* gamma,
                       float* x32, void* a16, float* rscale_out, float eps, void* stream);
/* ... and its end: y bf16 = w * x32 * rsqrt(mean(x32^2) + eps), one rounding                                                         */
int lrx_rmsnorm_f32(const float* x, const void* w, void* y, int32_t rows, int32_t hidden, float eps, void* stream);
/* rscale_out[r] = rsqrt(mean(x[r,:]^2) + eps) for bf16 rows (LlamaRMSNorm's statistic, modeling_llama.py:53-67) */
int lrx_row_rscale(const void* x, int32_t rows, int32_t hidden_size, float eps, float* rscale_out, void* stream);
/* rscale_out[r] = rsqrt(sum_p ss_part[p, r] / hidden_size + eps), partials added in index order */
int lrx_finalize_rscale(const float* ss_part, int32_t n_parts, int32_t rows, int32_t hidden_size, float eps, float* rscale_out,
                        void* stream);

/* dst[b, :] = src[cu_seqlens[b+1]-1, :]  (bf16 rows of `width` elements): the last-token rows, compacted. */
int lrx_gather_last_rows(const void* src, const int32_t* cu_seqlens, int32_t n_seqs, int32_t width, void* dst, void* stream);
/* (ABI 6) the inverse: dst[cu_seqlens[b+1]-1, 0:width] = src[b, 0:width]  (16-bit elements; rows of dst are dst_row_stride elements apart) */
int lrx_scatter_last_rows(const void* src, const int32_t* cu_seqlens, int32_t n_seqs, int32_t width, void* dst, int64_t dst_row_stride,
                          void* stream);

/* Last-token pooling + final RMSNorm on the pooled rows only + MRL slice + L2 normalise -> fp32 rows.
 * hidden = residual stream BEFORE the final norm [T, H] bf16; cu_seqlens == NULL means `hidden` already holds the
 * n_seqs pooled rows compacted ([n_seqs, H]).                                                                      */
int lrx_pool_norm(const void* hidden, const void* final_norm_w, const int32_t* cu_seqlens, int32_t n_seqs,
                  int32_t hidden_size, float eps, float* out, int64_t out_row_stride, int32_t out_dim,
                  int32_t normalize, void* stream);
/* ... and the shard maintenance fused into it: shadow_out (the shard's tiled fp16 shadow, row b of `out` = its row shadow_row0 + b; may be
 * NULL) and row_bounds (see lrx_shard_commit_rows; may be NULL).  hidden_f32 != 0: `hidden` holds fp32 rows (precise_stream), the norm
 * runs in fp32.                                                                                                                     */
int lrx_pool_norm_shard(const void* hidden, const void* final_norm_w, const int32_t* cu_seqlens, int32_t n_seqs,
                        int32_t hidden_size, float eps, float* out, int64_t out_row_stride, int32_t out_dim,
                        int32_t normalize, void* shadow_out, int64_t shadow_row0, float* row_bounds, int32_t hidden_f32, void* stream);

/* (ABI 8) ... and with the pooling strategy (LRX_POOL_LASTTOKEN .. LRX_POOL_THIRD_TO_LAST; cu_seqlens required for anything but LASTTOKEN): pooling(last_hidden, mask, strategy)
 * of finetune/dense_pooling.py:12-82 over `hidden` = the residual stream before the final norm, which runs inside (per pooled token). */
int lrx_pool_norm_mode(const void* hidden, const void* final_norm_w, const int32_t* cu_seqlens, int32_t n_seqs,
                       int32_t hidden_size, float eps, int32_t pooling, float* out, int64_t out_row_stride, int32_t out_dim,
                       int32_t normalize, void* shadow_out, int64_t shadow_row0, float* row_bounds, int32_t hidden_f32, void* stream);

/* Query side.  Replaces emb_bag.forward + slice + F.normalize at finetune/modeling_hybrid.py:472-490
 * (torch.nn.EmbeddingBag mode='mean', padding_idx) with inputs from tokenize_nonctx_qry_emb_bag
 * (finetune/nonctx_emb_utils.py:197-219).  table fp32 [V, H]; ids int64 [n_ids]; offsets int64 [n_bags];
 * padding_idx < 0 = none; empty bag -> zero row.  out fp32 [n_bags, out_dim].                                    */
int lrx_embedding_bag_mean(const float* table, int32_t vocab, int32_t hidden, const int64_t* ids, int64_t n_ids,
                           const int64_t* offsets, int32_t n_bags, int64_t padding_idx, float* out,
                           int64_t out_row_stride, int32_t out_dim, int32_t normalize, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Flat inner-product index shard resident in HBM.  Replaces faiss.IndexFlatIP.search as called from
 * FaissIndex.search (retriever/faiss_index.py:27-40): exact fp32 scores, descending top-k, ids = row numbers
 * (+ id_base, so shards can return global rows), ties broken by lower row id, id -1 / score -FLT_MAX when k > N.
 *   X [N, D] fp32 row-major (row stride ldx floats), q [Q, D] fp32, D % 32 == 0, k <= 2048.
 * Memory: a shard that keeps the fp16 shadow below holds 6 bytes per element (fp32 rows + shadow: 1.5x the rows alone) -- BASELINE
 * config 3 (10M x 4096) on ONE 288-GB GPU is 164 + 82 GB + < 6 GB of workspace; row-sharded over 8 GPUs 31 GB each.
 * ---------------------------------------------------------------------------------------------------------- */
size_t lrx_flat_ip_workspace_bytes(int64_t n_rows, int32_t dim, int32_t n_queries, int32_t k);

/* One pass over the fp32 rows (six bf16 products = fp32-grade scores; the exact-fp32 MFMA for <= 32 queries) into a [Q, N] score
 * matrix, selection, exact rescoring (fp64 accumulation, one rounding).  The selection is rigorous: every row whose matrix score lies
 * within 2 eps6(q) = 2 (6 D + 8) 2^-23 |q| R of the k-th largest is rescored before the best k are returned (R = row_bounds[0] >=
 * max |x_row|).  row_bounds: DEVICE pointer to the shard's two bounds (lrx_shard_commit_rows) or NULL = not known: the rows are then
 * read once more to measure R.                                                                                                       */
int lrx_flat_ip_search(const float* X, int64_t n_rows, int64_t ldx, int32_t dim, const float* row_bounds, const float* q,
                       int32_t n_queries, int32_t k, int64_t id_base, float* out_scores, int64_t* out_ids, void* workspace,
                       size_t workspace_bytes, void* stream);

/* Same result as lrx_flat_ip_search (exact top-k, ids ascending among equal scores) at close to ONE pass over a 2-byte copy of the
 * shard and without a [queries, rows] score matrix, for shards whose rows satisfy the two bounds in row_bounds (DEVICE pointer to
 * two floats): row_bounds[0] >= max |x_row| and row_bounds[1] >= max |x_row - fp16(x_row)| (<= 0: unknown, 2^-11 * row_bounds[0] is
 * used).  lrx_shard_commit_rows / lrx_encode_packed_shard maintain both.  A single-product FP16 filter pass (fp16(q) . fp16(x), fp32
 * accumulation) bounds every score to +- eps(q) = |q - fp16(q)| R + |fp16(q)| E + (D+32) 2^-23 |fp16(q)| R (~7e-4 |q| R on normalised
 * rows; the bf16 filter of round 2 had 3.7e-3); a strided sample of the shard (every ss-th 128-row block, scored first) gives a lower
 * bound T' of the k-th largest filter score; the pass over the rest keeps only rows with filter score >= T' - 2 eps (a per-query
 * candidate list, ~1e-3 of the rows; capacity 64 k rounded up to a power of two, at least 64 Ki); the rows within 2 eps of the list's
 * k-th score are rescored exactly from the fp32 rows (fp64 accumulation, one rounding) and sorted.  Queries whose list or band
 * overflows (near-duplicate corpora, rows outside fp16's range) are redone by the six-product path (gated on a device flag, no host
 * sync).  Scores returned are the exactly rescored ones.  Bounds smaller than the true values void the guarantee.  row_bounds[1] <= 0
 * means "E not measured": the library then uses 2^-11 R + sqrt(dim) 2^-25, a bound for every shard with R <= 65504 (elements in fp16's
 * normal or subnormal range); with R > 65504 it uses E = R, i.e. every query takes the six-product path (exact, slower).
 * The CU count that sizes the persistent launches is cached per process for the device that was current at the first call.
 * X_shadow (optional, NULL = convert the fp32 rows on the fly: half the queries/s): the shard's TILED FP16 SHADOW kept by the caller
 * next to the fp32 rows -- round-to-nearest-even per element, saturating at +-65504; dim % 64 == 0; layout at lrx_shard_commit_rows;
 * row 0 of the array is the shard's row 0.
 * flags: LRX_SEARCH_FILTER_* below (A/B runs and tests; the result does not depend on it).  Per call -- there is no process-wide
 * search state, calls from several host threads (the reference's RPC server threads) do not interact.
 * Queries are processed in chunks of 256 (128 without X_shadow) over the same workspace, whose size therefore stops growing at
 * 256 queries: min(Q,128) * rows * 4 bytes for the gated fallback plus Q candidate lists.                                          */
enum {
  LRX_SEARCH_FILTER_AUTO = 0,            /* score-free filter from 16 Ki rows on, score-matrix filter below                         */
  LRX_SEARCH_FILTER_MATRIX = 1,          /* always the score-matrix filter                                                           */
  LRX_SEARCH_FILTER_SCORE_FREE = 2,      /* the score-free (candidate-list) filter whenever the shard is large enough for a sample   */
  LRX_SEARCH_FILTER_SCORE_FREE_NO_GEMM = 3, /* like 2, but chunks of 129..256 queries never take the GEMM kernel for the main pass   */
  /* (ABI 6) OR-ed onto one of the above: the FUSED launch of the score-free filter -- sample pass, threshold selection and main pass in one
   * persistent kernel, for chunks of <= 128 queries over a shadow with dim % 256 == 0 -- is chosen by a measured rule (<= 16 queries, dim >= 512,
   * k <= 256, <= 8 blocks of 128 rows per CU: small query batches over a per-rank shard); these two bits force it on wherever it is eligible, or off */
  LRX_SEARCH_FUSED_ALWAYS = 4,
  LRX_SEARCH_FUSED_NEVER = 8,
  /* Exact rescoring of the band rows: per query (gather of its fp32 rows), or -- many queries x large k over a small shard, where every row
   * is wanted by several queries of a chunk -- grouped by ROW: each 16-row group staged once, the query rows streamed from L2; same bits.
   * Default: by a measured rule (rows of >= 8 KiB that a chunk of <= 256 queries wants >= 2.5 times: queries x k x 1.25 >= 2.5 x rows).
   * OR-able with the bits above; 16 | 32 is rejected.                                                                                     */
  LRX_SEARCH_REFINE_ROWS_ALWAYS = 16,
  LRX_SEARCH_REFINE_ROWS_NEVER = 32
};
size_t lrx_flat_ip_bounded_workspace_bytes(int64_t n_rows, int32_t dim, int32_t n_queries, int32_t k, int32_t flags);
int lrx_flat_ip_search_bounded(const float* X, int64_t n_rows, int64_t ldx, int32_t dim, const void* X_shadow, const float* row_bounds,
                               const float* q, int32_t n_queries, int32_t k, int64_t id_base, float* out_scores, int64_t* out_ids,
                               void* workspace, size_t workspace_bytes, int32_t flags, void* stream);
/* (ABI 5) The same search for a rank of a row-sharded index: besides (out_scores, out_ids) the LAST kernel of the chain also writes the
 * k results of every query as the 64-bit wire words of the exchange (out_wire [n_queries, k]; format and row_map as lrx_pack_topk
 * below), so nothing runs between the local search and the RCCL all-gather.  out_wire == NULL: identical to the call above.       */
int lrx_flat_ip_search_bounded_wire(const float* X, int64_t n_rows, int64_t ldx, int32_t dim, const void* X_shadow, const float* row_bounds,
                                    const float* q, int32_t n_queries, int32_t k, int64_t id_base, float* out_scores, int64_t* out_ids,
                                    const int64_t* row_map, uint64_t* out_wire, void* workspace, size_t workspace_bytes, int32_t flags,
                                    void* stream);
/* (ABI 5) Statistics for tools and bench legs.  lrx_search_fallback_count: queries any bounded search of this process has sent to its
 * exact six-product fallback since the last reset (list / band overflow: a performance event, the results are exact either way);
 * SYNCHRONISES like lrx_device_error_count.  lrx_flat_ip_bounded_list_counts: counts_out[n_queries] (device, uint32) = candidate-list
 * entries of each query of the last chunk of the last search that used `workspace` -- the rows that passed the filter and reached the
 * refine step; pass that search's own (n_rows, dim, n_queries of that chunk, k, flags) and whether it had a shadow; zeros when that search
 * ran the score-matrix filter.  Asynchronous on `stream`.                                                                          */
int64_t lrx_search_fallback_count(int32_t reset);
/* (ABI 8) Queries per chunk the bounded search walks a call of `n_queries` in: 256 over a shadow (128 over fp32 rows), or -- where the main
 * pass runs on the GEMM kernel (shadow, dim >= 1024, score-free filter feasible) -- ONE pass over the shadow per up to 1024 queries (equal
 * chunks, a multiple of 16 each).  Callers that pipeline chunks themselves (FlatIPIndex.search) split at these boundaries.                */
int32_t lrx_flat_ip_bounded_chunk_queries(int64_t n_rows, int32_t dim, int32_t n_queries, int32_t k, int32_t flags, int32_t has_shadow);
int lrx_flat_ip_bounded_list_counts(const void* workspace, int64_t n_rows, int32_t dim, int32_t n_queries, int32_t k, int32_t flags,
                                    int32_t has_shadow, uint32_t* counts_out, void* stream);

/* Shard maintenance (FaissIndex.build / IndexFlatIP.add, retriever/faiss_index.py:45-58, for rows that were not written by
 * lrx_encode_packed_shard): one read of n_rows fp32 rows writes their fp16 shadow (X_shadow may be NULL: bounds only) and raises
 * row_bounds[0] = max |row|, row_bounds[1] = max |row - fp16(row)| (device, two floats, zero-initialised by the caller when the
 * shard is created; integer atomic max, order-independent).  dim % 4 == 0.                                                      */
int lrx_shard_commit_rows(const float* X, int64_t ldx, int64_t n_rows, int32_t dim, void* X_shadow, int64_t shadow_row0,
                          float* row_bounds, void* stream);
/* The shadow layout (dim % 64 == 0): X_shadow = base of an array [ceil(rows / 128)][dim / 64] of 16-KiB tiles (128 rows x 64 columns),
 * each tile fragment-major: [16-row group w = 0..7][k-step ks = 0..1][fq = 0..3][fi = 0..15][8] fp16 holds row 16 w + fi, columns
 * 32 ks + 8 fq .. + 7 of the tile (the MFMA 16x16x32 operand of one wave), i.e. element k of row r sits at
 *   ((r / 128) * (dim / 64) + k / 64) * 8192 + ((((r / 16) % 8) * 2 + (k / 32) % 2) * 64 + ((k / 8) % 4) * 16 + r % 16) * 8 + k % 8
 * -- allocated for whole 128-row blocks; shadow_row0 = index within that array of X's first row (writers); for
 * lrx_flat_ip_search_bounded the shard's row 0 is row 0 of the array.  A wave of the filter pass loads its operand with one
 * coalesced 1-KiB request straight into registers (no LDS staging of the corpus side).  (Round 2 also accepted a row-major bf16
 * shadow; it is gone: the tiled layout was faster at every shape and fp16 gives the narrower band.)                               */

/* Score pass only: scores[Q, ld] fp32 with ld = lrx_flat_ip_score_ld(N); columns >= N hold -FLT_MAX. */
int64_t lrx_flat_ip_score_ld(int64_t n_rows);
int lrx_flat_ip_scores(const float* X, int64_t n_rows, int64_t ldx, int32_t dim, const float* q, int32_t n_queries,
                       float* scores, void* stream);

/* Merge R per-shard result lists (after the RCCL all-gather of [Q,k] pairs; replaces faiss IndexShards' host merge
 * used via index_cpu_to_all_gpus, retriever/faiss_index.py:65-68) : in_scores/in_ids [R, Q, k] -> out [Q, k].
 * Limits: R * k <= 16384 (one LDS-resident bitonic merge per query); ids are carried as 32 bits inside the merge: 0 <= id < 2^32. */
int lrx_merge_topk(const float* in_scores, const int64_t* in_ids, int32_t n_parts, int32_t n_queries, int32_t k,
                   float* out_scores, int64_t* out_ids, void* stream);
/* The exchange form: one 64-bit word per hit (fp32 score bits << 32 | global row as uint32, 0xFFFFFFFF = none), so the
 * all-gather moves a single [Q, k] int64 array per rank.  lrx_pack_topk builds it from a shard's (scores, ids) -- row_map (may be
 * NULL) turns shard-local rows (ids - id_base) into global rows; ids < 0 stay "none"; global rows must be < 2^32 - 1 --
 * lrx_merge_topk_packed merges the gathered [R, Q, k] words (R * k <= 16384) with lrx_merge_topk's ordering rule.               */
int lrx_pack_topk(const float* scores, const int64_t* ids, const int64_t* row_map, int64_t id_base, int64_t n,
                  uint64_t* out_packed, void* stream);
int lrx_merge_topk_packed(const uint64_t* in_packed, int32_t n_parts, int32_t n_queries, int32_t k, float* out_scores,
                          int64_t* out_ids, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LRX_H */
