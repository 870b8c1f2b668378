// Sparse document vectors: LM-head max aggregation without the [B,S,V] logits, sparsify, compaction.
//   reference: finetune/sparse_pooling.py:244-278 (aggregate), utils/max_linear_map.py:8-88 (per-timestep matmul + running max),
//              finetune/modeling_hybrid.py:176-203 (relu / log1p / top-k), finetune/sparse_converter_mixin.py:105-160 (quantise).
// The GEMM itself (bf16 MFMA, segmented column maximum in the epilogue) lives in lrx_gemm.hip; this file holds the row->segment
// map, the element-wise sparsify, an exact per-row radix select for the top-k threshold and the ordered compaction.
#include "lrx_common.h"

#define BF16_MIN_F (-3.3895313892515355e38f)   // torch.finfo(torch.bfloat16).min, the reference's running-max start value

__global__ void k_fill_f32(float* __restrict__ p, int64_t rows, int64_t cols, int64_t ld, float v) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < rows * cols) p[(i / cols) * ld + i % cols] = v;
}

// row_seg[t] = index of the sequence holding token t when the token takes part in the aggregation, else -1.
// tok_mask == NULL: the reference's default rule (sparse_pooling.py:23-41 without prompt removal): not the first, not the last token.
__global__ void k_build_row_seg(const int32_t* __restrict__ cu, int n_seqs, int T, const uint8_t* __restrict__ tok_mask, int32_t* __restrict__ row_seg) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  int lo = 0, hi = n_seqs;               // largest b with cu[b] <= t
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (cu[mid] <= t) lo = mid; else hi = mid;
  }
  const int s = cu[lo], e = cu[lo + 1];
  const bool valid = tok_mask ? tok_mask[t] != 0 : (t != s && t != e - 1);
  row_seg[t] = (t < cu[n_seqs] && valid) ? lo : -1;
}

extern "C" int lrx_sparse_max_aggregate(const void* hidden, const void* lm_head, const void* bias, const int32_t* cu_seqlens,
                                        const uint8_t* tok_mask, int32_t n_seqs, int32_t total_tokens, int32_t hidden_size, int32_t vocab_size,
                                        float* out, int64_t out_row_stride, int32_t* row_seg_workspace, void* stream) {
  LRX_CHECK_ARG(hidden && lm_head && cu_seqlens && out && row_seg_workspace, "sparse_max_aggregate: null operand");
  LRX_CHECK_ARG(n_seqs > 0 && total_tokens > 0 && hidden_size > 0 && vocab_size > 0, "sparse_max_aggregate: bad sizes");
  LRX_CHECK_ARG(out_row_stride >= vocab_size, "sparse_max_aggregate: out_row_stride=%lld < vocab_size=%d", (long long)out_row_stride, vocab_size);
  hipStream_t s = (hipStream_t)stream;
  const int64_t n = (int64_t)n_seqs * vocab_size;
  hipLaunchKernelGGL(k_fill_f32, dim3((unsigned)lrx_cdiv(n, 256)), dim3(256), 0, s, out, (int64_t)n_seqs, (int64_t)vocab_size, out_row_stride, BF16_MIN_F);
  LRX_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_build_row_seg, dim3((unsigned)lrx_cdiv(total_tokens, 256)), dim3(256), 0, s, cu_seqlens, n_seqs, total_tokens, tok_mask, row_seg_workspace);
  LRX_LAUNCH_CHECK();
  return lrx_gemm_max_aggregate_launch(hidden, lm_head, bias, row_seg_workspace, out, out_row_stride, total_tokens, vocab_size, hidden_size, s);
}

// ---------------------------------------------------------------------------------------------------------------
// sparsify: relu -> log1p (optionally rounded to bf16: the tensor is bf16 in the reference's bf16 run) in place
// ---------------------------------------------------------------------------------------------------------------
__global__ void k_sparse_transform(float* __restrict__ x, int64_t rows, int64_t cols, int64_t ld, int relu, int log1p_, int round_bf16) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * cols) return;
  float* p = x + (i / cols) * ld + i % cols;
  float v = *p;
  if (relu) v = fmaxf(v, 0.0f);
  if (log1p_) {
    v = log1pf(v);
    if (round_bf16) v = bf2f(f2bf(v));
  }
  *p = v;
}

__device__ __forceinline__ uint32_t f32_key(float v) {   // ascending-orderable key
  const uint32_t b = __float_as_uint(v);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float key_f32(uint32_t k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }

// One workgroup per row: exact k-th largest value by a 4-pass (8 bits each) radix select on the ordered keys, then everything
// strictly below it becomes filter_value (ties with the k-th value survive: sparse_pooling.py:92-109).
__global__ void __launch_bounds__(1024) k_topk_threshold(float* __restrict__ x, int cols, int64_t ld, int k, float filter_value) {
  __shared__ uint32_t hist[256];
  __shared__ uint32_t s_prefix, s_k;
  float* row = x + (int64_t)blockIdx.x * ld;
  const int tid = threadIdx.x;
  if (tid == 0) { s_prefix = 0; s_k = (uint32_t)k; }
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    const uint32_t prefix = s_prefix;
    for (int i = tid; i < cols; i += 1024) {
      const uint32_t key = f32_key(row[i]);
      if (pass == 0 || (key >> (shift + 8)) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (tid == 0) {
      uint32_t need = s_k, bin = 255;
      for (;; --bin) {
        const uint32_t c = hist[bin];
        if (c >= need || bin == 0) break;
        need -= c;
      }
      s_k = need;
      s_prefix = (prefix << 8) | bin;
    }
    __syncthreads();
  }
  const float thr = key_f32(s_prefix);
  for (int i = tid; i < cols; i += 1024)
    if (row[i] < thr) row[i] = filter_value;
}

extern "C" int lrx_sparsify(float* reps, int32_t n_rows, int32_t vocab_size, int64_t row_stride, int32_t relu, int32_t log1p, int32_t round_bf16,
                            int32_t top_k, int32_t min_tokens_to_keep, void* stream) {
  LRX_CHECK_ARG(reps && n_rows >= 0 && vocab_size > 0 && row_stride >= vocab_size, "sparsify: bad operand");
  LRX_CHECK_ARG(top_k >= 0 && min_tokens_to_keep >= 0, "sparsify: bad top_k=%d / min_tokens_to_keep=%d", top_k, min_tokens_to_keep);
  if (n_rows == 0) return LRX_OK;
  hipStream_t s = (hipStream_t)stream;
  if (relu || log1p) {
    const int64_t n = (int64_t)n_rows * vocab_size;
    hipLaunchKernelGGL(k_sparse_transform, dim3((unsigned)lrx_cdiv(n, 256)), dim3(256), 0, s, reps, (int64_t)n_rows, (int64_t)vocab_size, row_stride, relu, log1p, round_bf16);
    LRX_LAUNCH_CHECK();
  }
  if (top_k > 0) {
    int k = top_k > min_tokens_to_keep ? top_k : min_tokens_to_keep;
    if (k > vocab_size) k = vocab_size;
    hipLaunchKernelGGL(k_topk_threshold, dim3(n_rows), dim3(1024), 0, s, reps, vocab_size, row_stride, k, 0.0f);
    LRX_LAUNCH_CHECK();
  }
  return LRX_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// compaction: per row, the entries whose quantised weight round-half-even(max(x,0) * q) is non-zero, in ascending
// token id order (torch.nonzero order, sparse_converter_mixin.py:129-137); at most `capacity` are stored, counts[b] is
// the true number.  One workgroup per row, ballot + prefix sums.
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) k_sparse_compact(const float* __restrict__ x, int cols, int64_t ld, float quant, int capacity,
                                                         int32_t* __restrict__ ids, int32_t* __restrict__ weights, int32_t* __restrict__ counts) {
  __shared__ int wave_cnt[16];
  __shared__ int s_base;
  const float* row = x + (int64_t)blockIdx.x * ld;
  int32_t* oid = ids + (int64_t)blockIdx.x * capacity;
  int32_t* ow = weights + (int64_t)blockIdx.x * capacity;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) s_base = 0;
  __syncthreads();
  for (int c0 = 0; c0 < cols; c0 += 1024) {
    const int i = c0 + tid;
    int q = 0;
    if (i < cols) q = (int)rintf(fmaxf(row[i], 0.0f) * quant);
    const bool nz = q != 0;
    const unsigned long long bal = __ballot(nz);
    const int before = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wave_cnt[wave] = __popcll(bal);
    __syncthreads();
    int off = s_base;
    for (int w = 0; w < wave; ++w) off += wave_cnt[w];
    if (nz) {
      const int slot = off + before;
      if (slot < capacity) { oid[slot] = i; ow[slot] = q; }
    }
    __syncthreads();
    if (tid == 0) {
      int t = 0;
      for (int w = 0; w < 16; ++w) t += wave_cnt[w];
      s_base += t;
    }
    __syncthreads();
  }
  if (tid == 0) counts[blockIdx.x] = s_base;
}

extern "C" int lrx_sparse_compact(const float* reps, int32_t n_rows, int32_t vocab_size, int64_t row_stride, int32_t quantization_factor,
                                  int32_t capacity, int32_t* ids_out, int32_t* weights_out, int32_t* counts_out, void* stream) {
  LRX_CHECK_ARG(reps && ids_out && weights_out && counts_out, "sparse_compact: null operand");
  LRX_CHECK_ARG(n_rows >= 0 && vocab_size > 0 && row_stride >= vocab_size && capacity > 0 && quantization_factor > 0, "sparse_compact: bad sizes");
  if (n_rows == 0) return LRX_OK;
  hipLaunchKernelGGL(k_sparse_compact, dim3(n_rows), dim3(1024), 0, (hipStream_t)stream, reps, vocab_size, row_stride, (float)quantization_factor, capacity,
                     ids_out, weights_out, counts_out);
  LRX_LAUNCH_CHECK();
  return LRX_OK;
}
