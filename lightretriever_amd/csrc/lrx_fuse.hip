// Hit-list fusion on the device: reciprocal-rank fusion and min-max linear interpolation of the hit lists of
// several retrieval systems, restating retriever/score_fuse_utils.py:3-91 on arrays instead of dict-of-dicts.  All arithmetic in
// IEEE double like the reference's numpy float64, contributions of one document summed in system order -> bit-identical scores.
//   stage 1  lrx_hit_contributions: per (query, system) list -> per-entry contribution (1/(k+rank) | w*(s-min)/(max-min+eps))
//   stage 2  lrx_hit_union:         per query, concatenated lists -> union by document id, summed, sorted by fused score
#include "lrx_common.h"

#define FUSE_MAX 4096      // entries per workgroup-sorted list: three systems x top-1000 or two x top-2000 (80 KiB of LDS in lrx_hit_union)

template <typename Less, typename Swap>
__device__ __forceinline__ void bitonic_sort(int n_pow2, Less less, Swap swp) {
  for (int k = 2; k <= n_pow2; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < n_pow2; i += blockDim.x) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const bool up = (i & k) == 0;
          if (up ? less(ixj, i) : less(i, ixj)) swp(i, ixj);
        }
      }
      __syncthreads();
    }
  }
}

// method 0: RRF (param0 = k constant); method 1: linear (param0 = weight, param1 = eps)
__global__ void __launch_bounds__(1024) k_hit_contrib(const double* __restrict__ scores, const int64_t* __restrict__ ids, int k, int64_t ld,
                                                      int method, double p0, double p1, double* __restrict__ contrib, int64_t ldc) {
  __shared__ double s_sc[FUSE_MAX];
  __shared__ int s_pos[FUSE_MAX];
  __shared__ double s_red[2][16];
  const double* sc = scores + (int64_t)blockIdx.x * ld;
  const int64_t* id = ids + (int64_t)blockIdx.x * ld;
  double* out = contrib + (int64_t)blockIdx.x * ldc;
  const int tid = threadIdx.x;
  const double NEG = -__builtin_inf();
  if (method == 0) {
    int n2 = 1;
    while (n2 < k) n2 <<= 1;
    for (int i = tid; i < n2; i += blockDim.x) {
      s_sc[i] = (i < k && id[i] >= 0) ? sc[i] : NEG;
      s_pos[i] = i;
    }
    __syncthreads();
    // descending score, earlier position first among equals (the reference's argsort(-scores) leaves tie order unspecified)
    bitonic_sort(n2, [&](int a, int b) { return s_sc[a] > s_sc[b] || (s_sc[a] == s_sc[b] && s_pos[a] < s_pos[b]); },
                 [&](int a, int b) { double t = s_sc[a]; s_sc[a] = s_sc[b]; s_sc[b] = t; int u = s_pos[a]; s_pos[a] = s_pos[b]; s_pos[b] = u; });
    for (int r = tid; r < n2; r += blockDim.x) {
      const int p = s_pos[r];
      if (p < k) out[p] = id[p] >= 0 ? 1.0 / (p0 + (double)(r + 1)) : 0.0;
    }
  } else {
    double mn = __builtin_inf(), mx = NEG;
    for (int i = tid; i < k; i += blockDim.x)
      if (id[i] >= 0) { mn = fmin(mn, sc[i]); mx = fmax(mx, sc[i]); }
    for (int o = 32; o > 0; o >>= 1) { mn = fmin(mn, __shfl_xor(mn, o)); mx = fmax(mx, __shfl_xor(mx, o)); }
    if ((tid & 63) == 0) { s_red[0][tid >> 6] = mn; s_red[1][tid >> 6] = mx; }
    __syncthreads();
    mn = __builtin_inf(); mx = NEG;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) { mn = fmin(mn, s_red[0][w]); mx = fmax(mx, s_red[1][w]); }
    const double den = mx - mn + p1;
    for (int i = tid; i < k; i += blockDim.x) out[i] = id[i] >= 0 ? (sc[i] - mn) / den * p0 : 0.0;
  }
}

extern "C" int lrx_hit_contributions(const double* scores, const int64_t* ids, int32_t n_queries, int32_t k, int64_t row_stride, int32_t method,
                                     double param0, double param1, double* contrib_out, int64_t contrib_row_stride, void* stream) {
  LRX_CHECK_ARG(scores && ids && contrib_out, "hit_contributions: null operand");
  LRX_CHECK_ARG(n_queries >= 0 && k > 0 && k <= FUSE_MAX && row_stride >= k && contrib_row_stride >= k, "hit_contributions: bad sizes (k=%d, max %d)", k, FUSE_MAX);
  LRX_CHECK_ARG(method == 0 || method == 1, "hit_contributions: method %d (0 = rrf, 1 = linear)", method);
  if (n_queries == 0) return LRX_OK;
  hipLaunchKernelGGL(k_hit_contrib, dim3(n_queries), dim3(1024), 0, (hipStream_t)stream, scores, ids, k, row_stride, method, param0, param1, contrib_out,
                     contrib_row_stride);
  LRX_LAUNCH_CHECK();
  return LRX_OK;
}

__global__ void __launch_bounds__(1024) k_hit_union(const int64_t* __restrict__ ids, const double* __restrict__ contrib, int n, int64_t ld,
                                                    double* __restrict__ out_scores, int64_t* __restrict__ out_ids, int32_t* __restrict__ counts) {
  __shared__ int64_t s_id[FUSE_MAX];
  __shared__ double s_sc[FUSE_MAX];
  __shared__ int s_pos[FUSE_MAX];
  __shared__ int s_cnt;
  const int tid = threadIdx.x;
  const int64_t INV = 0x7fffffffffffffffll;
  int n2 = 1;
  while (n2 < n) n2 <<= 1;
  if (tid == 0) s_cnt = 0;
  for (int i = tid; i < n2; i += blockDim.x) {
    const int64_t v = i < n ? ids[(int64_t)blockIdx.x * ld + i] : -1;
    s_id[i] = v >= 0 ? v : INV;
    s_sc[i] = i < n ? contrib[(int64_t)blockIdx.x * ld + i] : 0.0;
    s_pos[i] = i;
  }
  __syncthreads();
  auto swp = [&](int a, int b) {
    int64_t t = s_id[a]; s_id[a] = s_id[b]; s_id[b] = t;
    double u = s_sc[a]; s_sc[a] = s_sc[b]; s_sc[b] = u;
    int p = s_pos[a]; s_pos[a] = s_pos[b]; s_pos[b] = p;
  };
  // by document, then by original position = system order (the order the reference accumulates the systems in)
  bitonic_sort(n2, [&](int a, int b) { return s_id[a] < s_id[b] || (s_id[a] == s_id[b] && s_pos[a] < s_pos[b]); }, swp);
  // heads sum their run (n2 <= FUSE_MAX = FUSE_MAX / 1024 rounds of the 1024 threads); two passes so nobody overwrites what a head still reads
  constexpr int ROUNDS = FUSE_MAX / 1024;
  double tot[ROUNDS];
  bool hd[ROUNDS];
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) { tot[r] = 0.0; hd[r] = false; }
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    const int i = tid + r * 1024;
    if (i >= n2) break;
    hd[r] = s_id[i] != INV && (i == 0 || s_id[i - 1] != s_id[i]);
    if (hd[r]) {
      double t = 0.0;
      for (int j = i; j < n2 && s_id[j] == s_id[i]; ++j) t += s_sc[j];
      tot[r] = t;
    }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    const int i = tid + r * 1024;
    if (i >= n2) break;
    if (hd[r]) { s_sc[i] = tot[r]; atomicAdd(&s_cnt, 1); }
    else { s_id[i] = INV; s_sc[i] = -__builtin_inf(); }
  }
  __syncthreads();
  // by fused score, descending; lower document id first among equals; dropped entries last
  bitonic_sort(n2, [&](int a, int b) {
    const bool va = s_id[a] != INV, vb = s_id[b] != INV;
    if (va != vb) return va;
    return s_sc[a] > s_sc[b] || (s_sc[a] == s_sc[b] && s_id[a] < s_id[b]);
  }, swp);
  const int cnt = s_cnt;
  for (int i = tid; i < n; i += blockDim.x) {
    out_scores[(int64_t)blockIdx.x * ld + i] = i < cnt ? s_sc[i] : -__builtin_inf();
    out_ids[(int64_t)blockIdx.x * ld + i] = i < cnt ? s_id[i] : -1;
  }
  if (tid == 0) counts[blockIdx.x] = cnt;
}

extern "C" int lrx_hit_union(const int64_t* ids, const double* contrib, int32_t n_queries, int32_t n_entries, int64_t row_stride, double* scores_out,
                             int64_t* ids_out, int32_t* counts_out, void* stream) {
  LRX_CHECK_ARG(ids && contrib && scores_out && ids_out && counts_out, "hit_union: null operand");
  LRX_CHECK_ARG(n_queries >= 0 && n_entries > 0 && n_entries <= FUSE_MAX && row_stride >= n_entries, "hit_union: bad sizes (entries=%d, max %d)", n_entries,
                FUSE_MAX);
  if (n_queries == 0) return LRX_OK;
  hipLaunchKernelGGL(k_hit_union, dim3(n_queries), dim3(1024), 0, (hipStream_t)stream, ids, contrib, n_entries, row_stride, scores_out, ids_out, counts_out);
  LRX_LAUNCH_CHECK();
  return LRX_OK;
}
