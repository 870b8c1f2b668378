// Memory-bound row kernels of the encoder: embedding gather, RMSNorm, RoPE, positions, last-token pool + final norm +
// L2 normalise, and the query-side EmbeddingBag(mean).  All bf16 traffic is 16 B per lane (Guideline 13).
#include "lrx_common.h"

// ---------------------------------------------------------------------------------------------------------------
// embedding gather: one wave per token row
// ---------------------------------------------------------------------------------------------------------------
// ids outside [0, vocab) never index the table: the row becomes zeros and a device-side counter is raised (read by
// lrx_device_error_count; the tokenizer / checkpoint pairing is wrong when that happens -- the host checks len(tokenizer) <= vocab).
__device__ unsigned int g_bad_token_ids = 0;

__global__ void __launch_bounds__(256) k_embedding_gather(const bf16x8* __restrict__ table, const int32_t* __restrict__ ids,
                                                          int n_tokens, int chunks /* H/8 */, int vocab, bf16x8* __restrict__ out) {
  int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n_tokens) return;
  int lane = threadIdx.x & 63;
  const int id = ids[row];
  int64_t dst = (int64_t)row * chunks;
  if (id < 0 || id >= vocab) {
    if (lane == 0) atomicAdd(&g_bad_token_ids, 1u);
    bf16x8 z;
#pragma unroll
    for (int j = 0; j < 8; ++j) z[j] = (__bf16)0.f;
    for (int c = lane; c < chunks; c += 64) out[dst + c] = z;
    return;
  }
  int64_t src = (int64_t)id * chunks;
  for (int c = lane; c < chunks; c += 64) out[dst + c] = table[src + c];
}

extern "C" int lrx_embedding_gather(const void* table, const int32_t* ids, int32_t n_tokens, int32_t hidden, int32_t vocab, void* out,
                                    void* stream) {
  LRX_CHECK_ARG(hidden % 8 == 0 && n_tokens >= 0 && vocab > 0, "embedding_gather: hidden %% 8 != 0 or vocab <= 0");
  if (n_tokens == 0) return LRX_OK;
  hipLaunchKernelGGL(k_embedding_gather, dim3(lrx_cdiv(n_tokens, 4)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16x8*)table, ids, n_tokens, hidden / 8, vocab, (bf16x8*)out);
  LRX_LAUNCH_CHECK();
  return LRX_OK;
}

__device__ unsigned int g_shadow_fp16_saturations = 0;   // (defined here: the first kernel that counts into it follows; described at lrx_device_saturation_count)

// Precise residual stream (lrx_encoder_config.precise_stream): the embedding row becomes the fp32 stream x32, the first projection's bf16 A
// operand a16 = bf16(x * gamma) (gamma = layer 0's input_layernorm weight: the norm weight rides on the operand, the weights stay exact) and
// the row statistic rs = rsqrt(mean(x^2) + eps).  One wave per token.  F16 (precise_stream = 2): a16 = fp16(x * gamma), saturating and counted.
template <bool F16>
__global__ void __launch_bounds__(256) k_embed_stream32(const bf16x8* __restrict__ table, const int32_t* __restrict__ ids, int n_tokens, int chunks,
                                                        int vocab, const bf16x8* __restrict__ gamma, float* __restrict__ x32, bf16x8* __restrict__ a16,
                                                        float* __restrict__ rs, float inv_h, float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n_tokens) return;
  const int lane = threadIdx.x & 63;
  const int id = ids[row];
  const bool bad = id < 0 || id >= vocab;
  if (bad && lane == 0) atomicAdd(&g_bad_token_ids, 1u);
  const int64_t src = (int64_t)(bad ? 0 : id) * chunks, dst = (int64_t)row * chunks;
  float ss = 0.f;
  for (int c = lane; c < chunks; c += 64) {
    bf16x8 v = table[src + c];
    const bf16x8 g = gamma[c];
    bf16x8 a;
    f32x4 lo, hi;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float f = bad ? 0.f : bf2f(v[j]);
      ss += f * f;
      const float t = f * bf2f(g[j]);
      if constexpr (F16) {
        a[j] = __builtin_bit_cast(__bf16, (_Float16)fminf(fmaxf(t, -65504.f), 65504.f));
        if (!(fabsf(t) <= 65504.f)) atomicAdd(&g_shadow_fp16_saturations, 1u);          // (an embedding row times a norm weight: never on a sane checkpoint)
      } else
        a[j] = f2bf(t);
      if (j < 4) lo[j] = f; else hi[j - 4] = f;
    }
    *(f32x4*)(x32 + (dst + c) * 8) = lo;
    *(f32x4*)(x32 + (dst + c) * 8 + 4) = hi;
    a16[dst + c] = a;
  }
  ss = wave_sum(ss);
  if (lane == 0) rs[row] = rsqrtf(ss * inv_h + eps);
}

extern "C" int lrx_embed_stream32(const void* table, const int32_t* ids, int32_t n_tokens, int32_t hidden, int32_t vocab, const void* gamma,
                                  float* x32, void* a16, float* rscale_out, float eps, void* stream) {
  return lrx_embed_stream32_ex(table, ids, n_tokens, hidden, vocab, gamma, x32, a16, rscale_out, eps, 0, stream);
}
int lrx_embed_stream32_ex(const void* table, const int32_t* ids, int32_t n_tokens, int32_t hidden, int32_t vocab, const void* gamma, float* x32, void* a16,
                          float* rscale_out, float eps, int f16, void* stream) {
  LRX_CHECK_ARG(hidden % 8 == 0 && n_tokens >= 0 && vocab > 0 && gamma && x32 && a16 && rscale_out, "embed_stream32: bad operand");
  if (n_tokens == 0) return LRX_OK;
  if (f16)
    hipLaunchKernelGGL(k_embed_stream32<true>, dim3(lrx_cdiv(n_tokens, 4)), dim3(256), 0, (hipStream_t)stream, (const bf16x8*)table, ids, n_tokens, hidden / 8,
                       vocab, (const bf16x8*)gamma, x32, (bf16x8*)a16, rscale_out, 1.0f / (float)hidden, eps);
  else
    hipLaunchKernelGGL(k_embed_stream32<false>, dim3(lrx_cdiv(n_tokens, 4)), dim3(256), 0, (hipStream_t)stream, (const bf16x8*)table, ids, n_tokens, hidden / 8,
                       vocab, (const bf16x8*)gamma, x32, (bf16x8*)a16, rscale_out, 1.0f / (float)hidden, eps);
  LRX_LAUNCH_CHECK();
  return LRX_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Measurement aids (bench.py, tools/): named no-op kernels that delimit a region of a rocprofv3 kernel trace, and a plain streaming read
// of a buffer -- the HBM read rate this box reaches with nothing but 16-B loads in flight, the in-run ceiling next to the 8 TB/s spec.
// ---------------------------------------------------------------------------------------------------------------
template <int ID>
__global__ void k_trace_marker() {}

extern "C" int lrx_trace_marker(int32_t id, void* stream) {
  LRX_CHECK_ARG(id >= 0 && id < 4, "trace_marker: id %d out of range (0..3)", id);
  hipStream_t s = (hipStream_t)stream;
  switch (id) {
    case 0: hipLaunchKernelGGL(k_trace_marker<0>, dim3(1), dim3(64), 0, s); break;
    case 1: hipLaunchKernelGGL(k_trace_marker<1>, dim3(1), dim3(64), 0, s); break;
    case 2: hipLaunchKernelGGL(k_trace_marker<2>, dim3(1), dim3(64), 0, s); break;
    default: hipLaunchKernelGGL(k_trace_marker<3>, dim3(1), dim3(64), 0, s); break;
  }
  LRX_LAUNCH_CHECK();
  return LRX_OK;
}

// Each workgroup walks its contiguous slice with 8 independent 16-B loads per thread per step (128 KiB per workgroup step, like the
// search filter's fragment stream); the xor of everything goes to sink[workgroup] so that the loads cannot be dropped.
__global__ void __launch_bounds__(256) k_stream_read(const u32x4* __restrict__ buf, int64_t n16, int64_t per_wg, uint32_t* __restrict__ sink) {
  const int64_t a = (int64_t)blockIdx.x * per_wg, b = min(a + per_wg, n16);
  u32x4 acc = {0u, 0u, 0u, 0u};
  for (int64_t i = a + threadIdx.x; i < b; i += 8 * 256) {
    u32x4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = i + j * 256 < b ? __builtin_nontemporal_load(buf + i + j * 256) : u32x4{0u, 0u, 0u, 0u};
#pragma unroll
    for (int j = 0; j < 8; ++j) acc ^= v[j];
  }
  uint32_t x = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x ^= __shfl_xor(x, o, 64);
  if ((threadIdx.x & 63) == 0) atomicXor(sink + blockIdx.x, x);
}

extern "C" int lrx_probe_stream_read(const void* buf, size_t bytes, uint32_t* sink, int32_t n_workgroups, void* stream) {
  LRX_CHECK_ARG(buf && sink && bytes % 16 == 0 && n_workgroups > 0, "probe_stream_read: bad operand");
  const int64_t n16 = (int64_t)(bytes / 16);
  const int64_t per = ((n16 + n_workgroups - 1) / n_workgroups + 2047) / 2048 * 2048;
  hipLaunchKernelGGL(k_stream_read, dim3(n_workgroups), dim3(256), 0, (hipStream_t)stream, (const u32x4*)buf, n16, per, sink);
  LRX_LAUNCH_CHECK();
  return LRX_OK;
}

// fp16 shadow elements k_pool_norm had to clamp (|v| > 65504 or NaN: an unnormalised row of a broken checkpoint); the shard's E bound keeps
// the SEARCH exact in that case, this counter makes the event visible (lrx_device_saturation_count).
unsigned int lrx_gemm_saturations(int* ok);                 // lrx_gemm.hip
int lrx_gemm_saturations_reset();

extern "C" int64_t lrx_device_saturation_count(int32_t reset) {
  // both counters are READ before either is cleared: a failing second read leaves the first one intact (-1, nothing lost)
  unsigned int v = 0;
  if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_shadow_fp16_saturations), sizeof(v)) != hipSuccess) return -1;
  int ok = 0;
  const unsigned int g = lrx_gemm_saturations(&ok);
  if (!ok) return -1;
  if (reset && v) {
    const unsigned int z = 0;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_shadow_fp16_saturations), &z, sizeof(z)) != hipSuccess) return -1;
  }
  if (reset && g && !lrx_gemm_saturations_reset()) return -1;
  return (int64_t)v + (int64_t)g;
}

unsigned int lrx_attn_list_overflows(int* ok, int reset);   // lrx_attn.hip: attention work lists the builder could not fit (their launches compute nothing)
extern "C" int64_t lrx_device_error_count(int32_t reset) {
  unsigned int v = 0;
  if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_bad_token_ids), sizeof(v)) != hipSuccess) return -1;
  int ok = 0;
  const unsigned int a = lrx_attn_list_overflows(&ok, reset);
  if (!ok) return -1;
  if (reset && v) {
    const unsigned int z = 0;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_bad_token_ids), &z, sizeof(z)) != hipSuccess) return -1;
  }
  return (int64_t)v + (int64_t)a;
}

// ---------------------------------------------------------------------------------------------------------------
// RMSNorm: one wave per row, row cached in registers (H <= 8192), HF rounding order:
//   y = bf16( w * bf16( x * rsqrt(mean(x^2) + eps) ) )
// ---------------------------------------------------------------------------------------------------------------
#define RMS_MAXC 16
__global__ void __launch_bounds__(256) k_rmsnorm(const bf16x8* __restrict__ x, const bf16x8* __restrict__ w, bf16x8* __restrict__ y,
                                                 int rows, int chunks, float inv_h, float eps) {
  int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  int lane = threadIdx.x & 63;
  const bf16x8* xr = x + (int64_t)row * chunks;
  bf16x8 v[RMS_MAXC];
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < RMS_MAXC; ++i) {
    int c = lane + i * 64;
    if (c < chunks) {
      v[i] = xr[c];
#pragma unroll
      for (int j = 0; j < 8; ++j) { float f = bf2f(v[i][j]); ss += f * f; }
    }
  }
  ss = wave_sum(ss);
  float rstd = rsqrtf(ss * inv_h + eps);
  bf16x8* yr = y + (int64_t)row * chunks;
#pragma unroll
  for (int i = 0; i < RMS_MAXC; ++i) {
    int c = lane + i * 64;
    if (c < chunks) {
      bf16x8 wv = w[c], o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = f2bf(bf2f(wv[j]) * bf2f(f2bf(bf2f(v[i][j]) * rstd)));
      yr[c] = o;
    }
  }
}

extern "C" int lrx_rmsnorm(const void* x, const void* w, void* y, int32_t rows, int32_t hidden, float eps, void* stream) {
  LRX_CHECK_ARG(hidden % 8 == 0 && hidden / 8 <= 64 * RMS_MAXC, "rmsnorm: hidden=%d unsupported", hidden);
  if (rows == 0) return LRX_OK;
  hipLaunchKernelGGL(k_rmsnorm, dim3(lrx_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, (const bf16x8*)x, (const bf16x8*)w,
                     (bf16x8*)y, rows, hidden / 8, 1.0f / (float)hidden, eps);
  LRX_LAUNCH_CHECK();
  return LRX_OK;
}

// The same statistic on an fp32 row (the precise residual stream), ONE rounding: y = bf16(w * x * rsqrt(mean(x^2) + eps)).
__global__ void __launch_bounds__(256) k_rmsnorm_f32(const float* __restrict__ x, const __bf16* __restrict__ w, __bf16* __restrict__ y, int rows, int H,
                                                     float inv_h, float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const float* xr = x + (int64_t)row * H;
  float ss = 0.f;
  for (int i = lane * 4; i < H; i += 256) {
    const f32x4 v = *(const f32x4*)(xr + i);
    ss += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
  }
  ss = wave_sum(ss);
  const float rstd = rsqrtf(ss * inv_h + eps);
  for (int i = lane * 4; i < H; i += 256) {
    const f32x4 v = *(const f32x4*)(xr + i);
    const bf16x4 wv = *(const bf16x4*)(w + i);
    bf16x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = f2bf(bf2f(wv[e]) * (v[e] * rstd));
    *(bf16x4*)(y + (int64_t)row * H + i) = o;
  }
}

extern "C" int lrx_rmsnorm_f32(const float* x, const void* w, void* y, int32_t rows, int32_t hidden, float eps, void* stream) {
  LRX_CHECK_ARG(hidden % 4 == 0 && x && w && y, "rmsnorm_f32: bad operand (hidden=%d)", hidden);
  if (rows == 0) return LRX_OK;
  hipLaunchKernelGGL(k_rmsnorm_f32, dim3(lrx_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, (const __bf16*)w, (__bf16*)y, rows, hidden,
                     1.0f / (float)hidden, eps);
  LRX_LAUNCH_CHECK();
  return LRX_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// positions[t] = t - cu[seq(t)]  (binary search; n_seqs is small)
// ---------------------------------------------------------------------------------------------------------------
__global__ void k_build_positions(const int32_t* __restrict__ cu, int n_seqs, int total, int32_t* __restrict__ pos) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  int lo = 0, hi = n_seqs;  // find largest b with cu[b] <= t
  while (hi - lo > 1) {
    int mid = (lo + hi) >> 1;
    if (cu[mid] <= t) lo = mid; else hi = mid;
  }
  pos[t] = t - cu[lo];
}

extern "C" int lrx_build_positions(const int32_t* cu_seqlens, int32_t n_seqs, int32_t total_tokens, int32_t* positions,
                                   void* stream) {
  if (total_tokens == 0) return LRX_OK;
  LRX_CHECK_ARG(n_seqs > 0, "build_positions: n_seqs must be > 0");
  hipLaunchKernelGGL(k_build_positions, dim3(lrx_cdiv(total_tokens, 256)), dim3(256), 0, (hipStream_t)stream, cu_seqlens, n_seqs,
                     total_tokens, positions);
  LRX_LAUNCH_CHECK();
  return LRX_OK;
}

// rscale[r] = rsqrt(mean(x[r,:]^2) + eps): the RMSNorm row statistic alone (the folded-norm pipeline applies it in the next GEMM's
// epilogue).  One wave per row, fp32 accumulation of the bf16 values like LlamaRMSNorm (modeling_llama.py:53-67).
__global__ void k_row_rscale(const bf16x8* __restrict__ x, int rows, int H, float eps, float* __restrict__ rscale) {
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const bf16x8* p = x + (int64_t)row * (H >> 3);
  float ss = 0.f;
  for (int i = lane; i < (H >> 3); i += 64) {
    const bf16x8 v = p[i];
#pragma unroll
    for (int e = 0; e < 8; ++e) { const float f = bf2f(v[e]); ss += f * f; }
  }
  ss = wave_sum(ss);
  if (lane == 0) rscale[row] = rsqrtf(ss / (float)H + eps);
}

extern "C" int lrx_row_rscale(const void* x, int32_t rows, int32_t hidden_size, float eps, float* rscale_out, void* stream) {
  LRX_CHECK_ARG(x && rscale_out && rows >= 0 && hidden_size > 0 && hidden_size % 8 == 0, "row_rscale: bad operand");
  if (rows == 0) return LRX_OK;
  hipLaunchKernelGGL(k_row_rscale, dim3(lrx_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, (const bf16x8*)x, rows, hidden_size, eps, rscale_out);
  LRX_LAUNCH_CHECK();
  return LRX_OK;
}

// rscale[r] = rsqrt(sum_p ss_part[p, r] / H + eps), partials added in index order (deterministic)
__global__ void k_finalize_rscale(const float* __restrict__ ss_part, int n_parts, int rows, float inv_h, float eps, float* __restrict__ rscale) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  float ss = 0.f;
  for (int p = 0; p < n_parts; ++p) ss += ss_part[(int64_t)p * rows + r];
  rscale[r] = rsqrtf(ss * inv_h + eps);
}

extern "C" int lrx_finalize_rscale(const float* ss_part, int32_t n_parts, int32_t rows, int32_t hidden_size, float eps, float* rscale_out,
                                   void* stream) {
  LRX_CHECK_ARG(ss_part && rscale_out && n_parts > 0 && rows >= 0 && hidden_size > 0, "finalize_rscale: bad operand");
  if (rows == 0) return LRX_OK;
  hipLaunchKernelGGL(k_finalize_rscale, dim3(lrx_cdiv(rows, 256)), dim3(256), 0, (hipStream_t)stream, ss_part, n_parts, rows, 1.0f / (float)hidden_size, eps,
                     rscale_out);
  LRX_LAUNCH_CHECK();
  return LRX_OK;
}

// cu[i] = i * len (i = 0..n) and positions[t] = offset + t % len : equal-length batches built on the device (no H2D copy)
__global__ void k_uniform_layout(int32_t* __restrict__ cu, int32_t* __restrict__ pos, int n_seqs, int len, int offset) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t <= n_seqs) cu[t] = t * len;
  if (t < n_seqs * len) pos[t] = offset + t % len;
}

extern "C" int lrx_uniform_layout(int32_t* cu_seqlens, int32_t* positions, int32_t n_seqs, int32_t len, int32_t position_offset, void* stream) {
  LRX_CHECK_ARG(n_seqs >= 0 && len > 0, "uniform_layout: bad sizes");
  int n = n_seqs * len + 1;
  hipLaunchKernelGGL(k_uniform_layout, dim3(lrx_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, cu_seqlens, positions, n_seqs, len, position_offset);
  LRX_LAUNCH_CHECK();
  return LRX_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// last-token pool + final RMSNorm (on the pooled rows only) + MRL slice + L2 normalise -> fp32 row.
// One 256-thread block per sequence.  Row cached in LDS as fp32.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float block_sum_256(float v, float* red) {
  v = wave_sum(v);
  int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// shadow != NULL: the row also goes out as fp16 (RNE, saturating) into the shard's tiled shadow -- the filter pass of the search streams that
// copy; bounds != NULL: the two shard bounds {max |row|, max |row - fp16(row)|} are raised by integer atomic max on the (non-negative) float
// patterns: the result does not depend on the order rows arrive in.  This is the index maintenance of FlatIPIndex.commit fused into the
// row's producer.
// F32: `hidden` holds fp32 rows (the precise residual stream): the final norm then runs in fp32 without the two bf16 roundings of HF's
// bf16 LlamaRMSNorm (the reference for the 1e-3 bound is the fp32 model).
// mode (LRX_POOL_*, include/lrx.h; finetune/dense_pooling.py:12-82): which row(s) of the sequence are pooled -- its last token (the
// released models), its first ('cls'), its second / third to last, or the MEAN of the final-norm rows of all its tokens (one token at a
// time through the same norm, fp32 accumulation in token order).  A sequence too short for its strategy (the reference asserts) gets a
// zero row and raises the input-error counter (lrx_device_error_count).
template <bool F32>
__global__ void __launch_bounds__(256) k_pool_norm(const void* __restrict__ hidden_v, const __bf16* __restrict__ w,
                                                   const int32_t* __restrict__ cu, int H, float eps, float* __restrict__ out,
                                                   int64_t out_stride, int out_dim, int normalize, __bf16* __restrict__ shadow,
                                                   int64_t shadow_row0, float* __restrict__ bounds, int mode, const float* __restrict__ aux) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* row = (float*)smem_raw;  // H floats
  float* red = row + H;           // 4 floats
  float* acc = red + 4;           // H floats (LRX_POOL_MEAN only)
  int b = blockIdx.x;
  int64_t t0 = cu ? (int64_t)cu[b + 1] - 1 : (int64_t)b, t1 = t0 + 1;      // rows [t0, t1) are pooled
  bool bad = false;
  if (cu && mode != LRX_POOL_LASTTOKEN) {
    const int64_t s0 = cu[b], s1 = cu[b + 1];
    if (mode == LRX_POOL_MEAN || mode >= LRX_POOL_AVG_FIRST_LAST) { t0 = s0; t1 = s1; bad = s1 <= s0; }
    else {
      t0 = mode == LRX_POOL_CLS ? s0 : s1 - (mode == LRX_POOL_SECOND_TO_LAST ? 2 : 3);
      t1 = t0 + 1;
      bad = t0 < s0 || s1 <= s0;
    }
  }
  if (bad) {
    if (threadIdx.x == 0) atomicAdd(&g_bad_token_ids, 1u);
    t1 = t0;                                                            // nothing pooled: a zero row goes out
  }
  const bool mean = (mode == LRX_POOL_MEAN || mode >= LRX_POOL_AVG_FIRST_LAST) && cu != nullptr;
  if (mean || bad) for (int i = threadIdx.x; i < out_dim; i += 256) { if (mean) acc[i] = 0.f; else row[i] = 0.f; }
  float n2 = 0.f;
  for (int64_t t = t0; t < t1; ++t) {
    float ss = 0.f;
    for (int i = threadIdx.x; i < H; i += 256) {
      const float f = F32 ? ((const float*)hidden_v)[t * H + i] : bf2f(((const __bf16*)hidden_v)[t * H + i]);
      row[i] = f;
      ss += f * f;
    }
    ss = block_sum_256(ss, red);
    float rstd = rsqrtf(ss / (float)H + eps);
    for (int i = threadIdx.x; i < out_dim; i += 256) {
      float y = F32 ? bf2f(w[i]) * (row[i] * rstd)
                    : bf2f(f2bf(bf2f(w[i]) * bf2f(f2bf(row[i] * rstd))));  // HF LlamaRMSNorm rounding order, bf16 result
      if (mean) acc[i] += y; else row[i] = y;
    }
    // (thread i owns columns i, i + 256, ... of row and acc in every loop of this kernel: no barrier needed between the tokens)
  }
  if (mean) {
    const float inv = t1 > t0 ? 1.0f / (float)(t1 - t0) : 0.f;
    // two-layer strategies: aux = the other hidden state's column sums (k_pool_sum) -- mean over tokens of (other + last) / 2
    if (aux != nullptr) for (int i = threadIdx.x; i < out_dim; i += 256) row[i] = (acc[i] + aux[(int64_t)b * H + i]) * (0.5f * inv);
    else for (int i = threadIdx.x; i < out_dim; i += 256) row[i] = acc[i] * inv;
  }
  for (int i = threadIdx.x; i < out_dim; i += 256) n2 += row[i] * row[i];
  n2 = block_sum_256(n2, red);
  float scale = normalize ? 1.0f / fmaxf(sqrtf(n2), 1e-12f) : 1.0f;
  float* o = out + (int64_t)b * out_stride;
  // shadow row: row shadow_row0 + b of the tiled layout (lrx_shadow_off)
  const int64_t ra = shadow_row0 + b;
  float r2 = 0.f, e2 = 0.f;
  for (int i = threadIdx.x; i < out_dim; i += 256) {
    const float v = normalize ? row[i] * scale : row[i];
    o[i] = v;
    const _Float16 h = (_Float16)fminf(fmaxf(v, -65504.f), 65504.f);
    if (shadow) {
      shadow[lrx_shadow_off(ra, i, out_dim)] = __builtin_bit_cast(__bf16, h);
      // (one atomic per wave instruction that saw any; the lanes of a wave that are still in the loop vote)
      if (__any(!(fabsf(v) <= 65504.f))) { if ((int)(threadIdx.x & 63) == __ffsll((long long)__ballot(1)) - 1) atomicAdd(&g_shadow_fp16_saturations, 1u); }
    }
    const float d = v - (float)h;
    r2 += v * v;
    e2 += d * d;
  }
  if (bounds != nullptr) {
    r2 = block_sum_256(r2, red);
    e2 = block_sum_256(e2, red);
    if (threadIdx.x == 0) {
      atomicMax((int*)bounds, __float_as_int(sqrtf(r2) * (1.0f + 1e-6f)));
      atomicMax((int*)bounds + 1, __float_as_int(sqrtf(e2) * (1.0f + 1e-6f)));
    }
  }
}

// Column sums of one sequence's rows of a second hidden state, for the two-layer strategies (finetune/dense_pooling.py:38-46): aux[b, :] =
// sum over the sequence's tokens of src[t, :], fp32, tokens added in order.  SRC 0 / 1: bf16 / fp32 rows (the stream BEFORE the final
// layer = hidden_states[-2], LRX_POOL_AVG_TOP2); SRC 2: the embedding rows of the tokens (= hidden_states[0], LRX_POOL_AVG_FIRST_LAST; an id
// outside the table adds a zero row, as in the embedding kernel that counted it).  grid (n_seqs, ceil(H / 256)).
template <int SRC>
__global__ void __launch_bounds__(256) k_pool_sum(const void* __restrict__ src, const int32_t* __restrict__ ids, int vocab, const int32_t* __restrict__ cu,
                                                  int H, float* __restrict__ aux) {
  const int b = blockIdx.x, i = blockIdx.y * 256 + threadIdx.x;
  if (i >= H) return;
  float acc = 0.f;
  for (int64_t t = cu[b]; t < cu[b + 1]; ++t) {
    if (SRC == 2) {
      const int id = ids[t];
      if (id >= 0 && id < vocab) acc += bf2f(((const __bf16*)src)[(int64_t)id * H + i]);
    } else
      acc += SRC == 1 ? ((const float*)src)[t * H + i] : bf2f(((const __bf16*)src)[t * H + i]);
  }
  aux[(int64_t)b * H + i] = acc;
}
int lrx_pool_sum_rows(const void* src, int src_kind, const int32_t* ids, int vocab, const int32_t* cu_seqlens, int n_seqs, int hidden_size, float* aux,
                      hipStream_t stream) {
  LRX_CHECK_ARG(src && cu_seqlens && aux && src_kind >= 0 && src_kind <= 2 && (src_kind != 2 || ids != nullptr), "pool_sum: bad arguments");
  if (n_seqs == 0) return LRX_OK;
  const dim3 grid(n_seqs, lrx_cdiv(hidden_size, 256));
  if (src_kind == 2) hipLaunchKernelGGL(k_pool_sum<2>, grid, dim3(256), 0, stream, src, ids, vocab, cu_seqlens, hidden_size, aux);
  else if (src_kind == 1) hipLaunchKernelGGL(k_pool_sum<1>, grid, dim3(256), 0, stream, src, ids, vocab, cu_seqlens, hidden_size, aux);
  else hipLaunchKernelGGL(k_pool_sum<0>, grid, dim3(256), 0, stream, src, ids, vocab, cu_seqlens, hidden_size, aux);
  LRX_LAUNCH_CHECK();
  return LRX_OK;
}

// (internal form: `aux` = the [n_seqs, hidden_size] column sums of the other hidden state for LRX_POOL_AVG_FIRST_LAST / AVG_TOP2, else NULL)
int lrx_pool_norm_aux(const void* hidden, const void* final_norm_w, const int32_t* cu_seqlens, int32_t n_seqs, int32_t hidden_size, float eps,
                      int32_t pooling, const float* aux, float* out, int64_t out_row_stride, int32_t out_dim, int32_t normalize, void* shadow_out,
                      int64_t shadow_row0, float* row_bounds, int32_t hidden_f32, void* stream) {
  LRX_CHECK_ARG(out_dim > 0 && out_dim <= hidden_size, "pool_norm: out_dim=%d out of range (H=%d)", out_dim, hidden_size);
  LRX_CHECK_ARG(shadow_out == nullptr || (out_dim % 64 == 0 && shadow_row0 >= 0), "pool_norm: the tiled shadow needs out_dim %% 64 == 0 (out_dim %d)", out_dim);
  LRX_CHECK_ARG(pooling >= LRX_POOL_LASTTOKEN && pooling <= LRX_POOL_AVG_TOP2, "pool_norm: pooling=%d (LRX_POOL_*)", pooling);
  LRX_CHECK_ARG((pooling >= LRX_POOL_AVG_FIRST_LAST) == (aux != nullptr), "pool_norm: pooling %d %s the other hidden state's sums (served through lrx_encode_packed_pooled)",
                pooling, aux ? "does not take" : "needs");
  LRX_CHECK_ARG(pooling == LRX_POOL_LASTTOKEN || cu_seqlens != nullptr, "pool_norm: pooling %d needs cu_seqlens (rows already compacted are last-token rows)", pooling);
  if (n_seqs == 0) return LRX_OK;
  const bool mean = pooling == LRX_POOL_MEAN || pooling >= LRX_POOL_AVG_FIRST_LAST;
  size_t smem = (size_t)(hidden_size + 4 + (mean ? hidden_size : 0)) * sizeof(float);
  LRX_CHECK_ARG(smem <= 160 * 1024, "pool_norm: hidden_size=%d does not fit the LDS", hidden_size);
  if (smem > 64 * 1024)        // (beyond the default dynamic-LDS limit: H > 8188, or mean pooling at H > 8190 / 2)
    LRX_HIP(hipFuncSetAttribute(hidden_f32 ? (const void*)k_pool_norm<true> : (const void*)k_pool_norm<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  if (hidden_f32)
    hipLaunchKernelGGL(k_pool_norm<true>, dim3(n_seqs), dim3(256), smem, (hipStream_t)stream, hidden, (const __bf16*)final_norm_w, cu_seqlens,
                       hidden_size, eps, out, out_row_stride, out_dim, normalize, (__bf16*)shadow_out, shadow_row0, row_bounds, pooling, aux);
  else
    hipLaunchKernelGGL(k_pool_norm<false>, dim3(n_seqs), dim3(256), smem, (hipStream_t)stream, hidden, (const __bf16*)final_norm_w, cu_seqlens,
                       hidden_size, eps, out, out_row_stride, out_dim, normalize, (__bf16*)shadow_out, shadow_row0, row_bounds, pooling, aux);
  LRX_LAUNCH_CHECK();
  return LRX_OK;
}

extern "C" int lrx_pool_norm_mode(const void* hidden, const void* final_norm_w, const int32_t* cu_seqlens, int32_t n_seqs,
                                  int32_t hidden_size, float eps, int32_t pooling, float* out, int64_t out_row_stride, int32_t out_dim,
                                  int32_t normalize, void* shadow_out, int64_t shadow_row0, float* row_bounds, int32_t hidden_f32, void* stream) {
  return lrx_pool_norm_aux(hidden, final_norm_w, cu_seqlens, n_seqs, hidden_size, eps, pooling, nullptr, out, out_row_stride, out_dim, normalize, shadow_out,
                           shadow_row0, row_bounds, hidden_f32, stream);
}

extern "C" int lrx_pool_norm_shard(const void* hidden, const void* final_norm_w, const int32_t* cu_seqlens, int32_t n_seqs,
                                   int32_t hidden_size, float eps, float* out, int64_t out_row_stride, int32_t out_dim, int32_t normalize,
                                   void* shadow_out, int64_t shadow_row0, float* row_bounds, int32_t hidden_f32, void* stream) {
  return lrx_pool_norm_mode(hidden, final_norm_w, cu_seqlens, n_seqs, hidden_size, eps, LRX_POOL_LASTTOKEN, out, out_row_stride, out_dim, normalize,
                            shadow_out, shadow_row0, row_bounds, hidden_f32, stream);
}

extern "C" int lrx_pool_norm(const void* hidden, const void* final_norm_w, const int32_t* cu_seqlens, int32_t n_seqs,
                             int32_t hidden_size, float eps, float* out, int64_t out_row_stride, int32_t out_dim, int32_t normalize,
                             void* stream) {
  return lrx_pool_norm_shard(hidden, final_norm_w, cu_seqlens, n_seqs, hidden_size, eps, out, out_row_stride, out_dim, normalize, nullptr, 0,
                             nullptr, 0, stream);
}

// ---------------------------------------------------------------------------------------------------------------
// gather of the last-token rows (compaction for the pooled tail of the final layer): one wave per sequence
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_gather_last_rows(const bf16x8* __restrict__ src, const int32_t* __restrict__ cu, int n_seqs, int chunks,
                                                          bf16x8* __restrict__ dst) {
  int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= n_seqs) return;
  int lane = threadIdx.x & 63;
  int64_t s = ((int64_t)cu[b + 1] - 1) * chunks, d = (int64_t)b * chunks;
  for (int c = lane; c < chunks; c += 64) dst[d + c] = src[s + c];
}

extern "C" int lrx_gather_last_rows(const void* src, const int32_t* cu_seqlens, int32_t n_seqs, int32_t width, void* dst, void* stream) {
  LRX_CHECK_ARG(width % 8 == 0, "gather_last_rows: width %% 8 != 0");
  if (n_seqs == 0) return LRX_OK;
  hipLaunchKernelGGL(k_gather_last_rows, dim3(lrx_cdiv(n_seqs, 4)), dim3(256), 0, (hipStream_t)stream, (const bf16x8*)src, cu_seqlens, n_seqs,
                     width / 8, (bf16x8*)dst);
  LRX_LAUNCH_CHECK();
  return LRX_OK;
}

__global__ void __launch_bounds__(256) k_gather_rows_u32(const uint32_t* __restrict__ src, const int32_t* __restrict__ cu, int n_seqs, uint32_t* __restrict__ dst) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b < n_seqs) dst[b] = src[cu[b + 1] - 1];
}
int lrx_gather_rows_u32(const void* src, const int32_t* cu_seqlens, int32_t n_seqs, void* dst, hipStream_t stream) {
  if (n_seqs == 0) return LRX_OK;
  hipLaunchKernelGGL(k_gather_rows_u32, dim3(lrx_cdiv(n_seqs, 256)), dim3(256), 0, stream, (const uint32_t*)src, cu_seqlens, n_seqs, (uint32_t*)dst);
  LRX_LAUNCH_CHECK();
  return LRX_OK;
}

// the inverse: compact rows back to the last-token positions of a wider row-major buffer (the final layer's q rows)
__global__ void __launch_bounds__(256) k_scatter_last_rows(const bf16x8* __restrict__ src, const int32_t* __restrict__ cu, int n_seqs, int chunks,
                                                           bf16x8* __restrict__ dst, int64_t dst_chunks) {
  int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= n_seqs) return;
  int lane = threadIdx.x & 63;
  int64_t d = ((int64_t)cu[b + 1] - 1) * dst_chunks, s = (int64_t)b * chunks;
  for (int c = lane; c < chunks; c += 64) dst[d + c] = src[s + c];
}

extern "C" int lrx_scatter_last_rows(const void* src, const int32_t* cu_seqlens, int32_t n_seqs, int32_t width, void* dst, int64_t dst_row_stride,
                                     void* stream) {
  LRX_CHECK_ARG(width % 8 == 0 && dst_row_stride % 8 == 0 && dst_row_stride >= width, "scatter_last_rows: width=%d / stride=%lld must be multiples of 8",
                width, (long long)dst_row_stride);
  if (n_seqs == 0) return LRX_OK;
  hipLaunchKernelGGL(k_scatter_last_rows, dim3(lrx_cdiv(n_seqs, 4)), dim3(256), 0, (hipStream_t)stream, (const bf16x8*)src, cu_seqlens, n_seqs,
                     width / 8, (bf16x8*)dst, dst_row_stride / 8);
  LRX_LAUNCH_CHECK();
  return LRX_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// EmbeddingBag(mode='mean', padding_idx): one block per bag, thread = column, ids walked in order (same fp32 summation
// order as the sequential CPU kernel), then slice + optional L2 normalise.
// ---------------------------------------------------------------------------------------------------------------
// EB_V4: columns in groups of four per thread (16-B table reads), the bag's ids staged once in LDS, ids outer / columns inner: a
// thread has all its column groups' loads of several ids in flight (the scalar version re-read the ids and chained one dependent
// 4-byte load per id per column: 88 us for 100 bags x ~20 ids at D = 2048, 8 % of a search pass).  Per column the rows are still
// added in id order -> the same fp32 sums bit for bit.
// EB_V4 = float4 column groups per thread (1, 2, 4, 8; 0 = scalar path).  The rows of a bag are summed in id order (the bits of
// torch.nn.EmbeddingBag), but their loads do not depend on each other: 8 / EB_V4 rows are requested before the first is added (one row
// at a time, a 20-token bag over H = 2048 took 21 us of serial load latency).
template <int EB_V4>
__global__ void __launch_bounds__(256) k_embedding_bag(const float* __restrict__ table, int vocab, int H, const int64_t* __restrict__ ids,
                                                       int64_t n_ids, const int64_t* __restrict__ offsets, int n_bags, int64_t pad,
                                                       float* __restrict__ out, int64_t out_stride, int out_dim, int normalize) {
  __shared__ float red[4];
  __shared__ int s_row[256];      // table row of each id of the current chunk, -1 = skipped (padding_idx / out of range)
  __shared__ int s_cnt[4];
  int b = blockIdx.x;
  int64_t s = offsets[b], e = (b + 1 < n_bags) ? offsets[b + 1] : n_ids;
  float n2 = 0.f;
  float* o = out + (int64_t)b * out_stride;
  if (EB_V4) {
    constexpr int MAXG = EB_V4 > 0 ? EB_V4 : 1;              // column groups per thread: out_dim <= 1024 * MAXG
    constexpr int UR = 8 / MAXG;                             // rows in flight
    const int ng = out_dim >> 2;                             // float4 groups in the output row
    f32x4 acc[MAXG];
#pragma unroll
    for (int j = 0; j < MAXG; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    int cnt = 0;
    for (int64_t c0 = s; c0 < e; c0 += 256) {
      const int nc = (int)min((int64_t)256, e - c0);
      __syncthreads();
      int mine = 0;
      if ((int)threadIdx.x < nc) {
        const int64_t id = ids[c0 + threadIdx.x];
        mine = id != pad ? 1 : 0;
        s_row[threadIdx.x] = (id != pad && id >= 0 && id < vocab) ? (int)id : -1;
      }
      // number of non-padding ids of the chunk (the mean's divisor counts them even when out of range, like the scalar kernel)
      const unsigned long long bal = __ballot(mine);
      if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = __popcll(bal);
      __syncthreads();
      cnt += s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
      for (int i0 = 0; i0 < nc; i0 += UR) {
        int rows[UR];
        f32x4 v[UR][MAXG];
#pragma unroll
        for (int u = 0; u < UR; ++u) {
          rows[u] = i0 + u < nc ? s_row[i0 + u] : -1;
          const f32x4* tr = (const f32x4*)(table + (int64_t)(rows[u] < 0 ? 0 : rows[u]) * H);
#pragma unroll
          for (int j = 0; j < MAXG; ++j) {
            const int g = threadIdx.x + 256 * j;
            v[u][j] = (rows[u] >= 0 && g < ng) ? tr[g] : f32x4{0.f, 0.f, 0.f, 0.f};
          }
        }
#pragma unroll
        for (int u = 0; u < UR; ++u)
          if (rows[u] >= 0) {                                  // (uniform; a skipped row adds nothing, not even + 0)
#pragma unroll
            for (int j = 0; j < MAXG; ++j) {
              const int g = threadIdx.x + 256 * j;
              if (g < ng) { acc[j][0] += v[u][j][0]; acc[j][1] += v[u][j][1]; acc[j][2] += v[u][j][2]; acc[j][3] += v[u][j][3]; }
            }
          }
      }
    }
#pragma unroll
    for (int j = 0; j < MAXG; ++j) {
      const int g = threadIdx.x + 256 * j;
      if (g < ng) {
        f32x4 m;
#pragma unroll
        for (int c = 0; c < 4; ++c) { m[c] = cnt > 0 ? acc[j][c] / (float)cnt : 0.f; n2 += m[c] * m[c]; }
        acc[j] = m;
      }
    }
    const float scale = normalize ? 1.0f / fmaxf(sqrtf(block_sum_256(n2, red)), 1e-12f) : 1.0f;
#pragma unroll
    for (int j = 0; j < MAXG; ++j) {
      const int g = threadIdx.x + 256 * j;
      if (g < ng) {
        f32x4 m = acc[j];
        if (normalize) { m[0] *= scale; m[1] *= scale; m[2] *= scale; m[3] *= scale; }
        *(f32x4*)(o + 4 * g) = m;
      }
    }
    return;
  }
  int cnt = 0;
  for (int64_t i = s; i < e; ++i) cnt += (ids[i] != pad) ? 1 : 0;
  // out_dim <= H; each thread owns columns threadIdx.x + 256*j
  for (int c = threadIdx.x; c < out_dim; c += 256) {
    float acc = 0.f;
    for (int64_t i = s; i < e; ++i) {
      int64_t id = ids[i];
      if (id != pad && id >= 0 && id < vocab) acc += table[id * (int64_t)H + c];
    }
    float m = cnt > 0 ? acc / (float)cnt : 0.f;
    o[c] = m;
    n2 += m * m;
  }
  if (!normalize) return;
  n2 = block_sum_256(n2, red);
  float scale = 1.0f / fmaxf(sqrtf(n2), 1e-12f);
  for (int c = threadIdx.x; c < out_dim; c += 256) o[c] *= scale;  // same thread wrote o[c]
}

extern "C" int lrx_embedding_bag_mean(const float* table, int32_t vocab, int32_t hidden, const int64_t* ids, int64_t n_ids,
                                      const int64_t* offsets, int32_t n_bags, int64_t padding_idx, float* out, int64_t out_row_stride,
                                      int32_t out_dim, int32_t normalize, void* stream) {
  LRX_CHECK_ARG(out_dim > 0 && out_dim <= hidden, "embedding_bag: out_dim=%d out of range (H=%d)", out_dim, hidden);
  if (n_bags == 0) return LRX_OK;
  // 16-B path: rows of the table and of the output 16-B aligned, <= 8 column groups per thread
  const bool v4 = hidden % 4 == 0 && out_dim % 4 == 0 && out_row_stride % 4 == 0 && out_dim <= 8192 && ((uintptr_t)table & 15) == 0 &&
                  ((uintptr_t)out & 15) == 0;
#define LRX_EB(NJ)                                                                                                                          \
  hipLaunchKernelGGL(k_embedding_bag<NJ>, dim3(n_bags), dim3(256), 0, (hipStream_t)stream, table, vocab, hidden, ids, n_ids, offsets, n_bags, \
                     padding_idx, out, out_row_stride, out_dim, normalize)
  if (!v4) LRX_EB(0);
  else if (out_dim <= 1024) LRX_EB(1);
  else if (out_dim <= 2048) LRX_EB(2);
  else if (out_dim <= 4096) LRX_EB(4);
  else LRX_EB(8);
#undef LRX_EB
  LRX_LAUNCH_CHECK();
  return LRX_OK;
}
