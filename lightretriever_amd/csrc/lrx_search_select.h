// liblrx search, part 2 -- SELECTION: radix / bitonic primitives, k_topk_select, exact_dot, k_topk_select_rescore (map: section B).
// Part of the ONE translation unit lrx_search.hip (included there, in source order: filter kernels -> selection -> bounded-search
// device code -> refine kernels; the host driver, the shard maintenance and the exchange kernels stay in lrx_search.hip).  Not a stand-alone header.
#pragma once

// ---------------------------------------------------------------------------------------------------------------
// top-k select
// ---------------------------------------------------------------------------------------------------------------

// sort buf[0..P) descending (P power of two), all threads of the block participate
// (Round 3, measured and not kept: workgroup barriers only before the 20 of 66 stages of a 2048-element sort that pair elements of different
// waves -- k_refine_merge at top-1000 stayed at 35 us: a stage costs its LDS read -> compare -> write latency, ~0.5 us, not its barrier.)
__device__ void bitonic_sort_desc(unsigned long long* buf, int P) {
  for (int k = 2; k <= P; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      __syncthreads();
      for (int i = threadIdx.x; i < P; i += blockDim.x) {
        int ixj = i ^ j;
        if (ixj > i) {
          unsigned long long a = buf[i], b = buf[ixj];
          bool up = (i & k) == 0;  // descending overall
          if (up ? (a < b) : (a > b)) { buf[i] = b; buf[ixj] = a; }
        }
      }
    }
  __syncthreads();
}

// The same sort for P = E x blockDim.x entries with E = 2 or 4 consecutive entries per thread IN REGISTERS (round 4): of the 66 stages of a
// 2048-entry sort 11 pair entries of one thread, 45 pair threads of one wave (64-bit lane exchange, no LDS, no barrier) and only 10 pair
// different waves (LDS round trip + two barriers).  The LDS version above pays ~0.5 us of read -> compare -> write latency for every one of
// the 66: k_refine_merge at top_k = 1000 (1 200 entries -> P = 2048) 34 us; this one 21.  Also the merge of the gathered per-shard lists
// (k_merge_topk: 8 shards x top-100 = 800 entries -> P = 1024, 55 stages).  blockDim.x a multiple of 64 (or one partial wave), P = E * blockDim.x.
template <int E>
__device__ void bitonic_sort_desc_regs(unsigned long long* buf, int P) {
  static_assert(E == 2 || E == 4 || E == 8 || E == 16, "2 .. 16 entries per thread");
  const int t = threadIdx.x, base = t * E;
  unsigned long long v[E];
  __syncthreads();
#pragma unroll
  for (int r = 0; r < E; ++r) v[r] = buf[base + r];
  for (int k = 2; k <= P; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      if (j < E) {
        // partner inside the thread: entries r and r ^ j
#pragma unroll
        for (int r = 0; r < E; ++r) {
          const int rp = r ^ j;
          if (rp > r && rp < E) {
            const bool upr = ((base + r) & k) == 0;
            const unsigned long long a = v[r], b = v[rp];
            if (upr ? (a < b) : (a > b)) { v[r] = b; v[rp] = a; }
          }
        }
      } else if (j < 64 * E) {
        // partner thread t ^ (j / E) in the same wave; the entry at the lower index keeps the larger word when (index & k) == 0
        const int pl = j / E;
        const bool lower = (t & pl) == 0;
#pragma unroll
        for (int r = 0; r < E; ++r) {
          const unsigned long long o = __shfl_xor(v[r], pl, 64);
          const bool upr = ((base + r) & k) == 0;                 // (the same for both partners: they differ in bit j < k only)
          const bool keep_max = lower == upr;
          v[r] = keep_max ? (v[r] > o ? v[r] : o) : (v[r] < o ? v[r] : o);
        }
      } else {
        // partner in another wave: through LDS
#pragma unroll
        for (int r = 0; r < E; ++r) buf[base + r] = v[r];
        __syncthreads();
        const bool lower = (base & j) == 0;
#pragma unroll
        for (int r = 0; r < E; ++r) {
          const unsigned long long o = buf[(base + r) ^ j];
          const bool upr = ((base + r) & k) == 0;
          const bool keep_max = lower == upr;
          v[r] = keep_max ? (v[r] > o ? v[r] : o) : (v[r] < o ? v[r] : o);
        }
        __syncthreads();
      }
    }
  }
#pragma unroll
  for (int r = 0; r < E; ++r) buf[base + r] = v[r];
  __syncthreads();
}

#define SEL_THREADS 1024
#define SEL_MAXK 2048
#define SEL_CAND 4096   // candidate capacity of the fast path (and of the exact path's output list)
#define SEL_EQCAP 2048

struct SelShared {
  unsigned int hist[16][256];
  unsigned long long cand[SEL_CAND];
  unsigned long long eqs[SEL_EQCAP];
  unsigned int eqidx[SEL_EQCAP];
  unsigned int bucket, kk, cnt, ngt, neq;
};

// One digit of the radix select after the per-wave histograms of that digit are complete: finds the bucket holding the kk-th largest
// key (suffix sums S(b) = count of keys in buckets >= b, by a wave scan per 64 buckets plus the totals of the higher waves -- a serial
// walk over 256 LDS entries by one thread cost ~8 us per pass), updates kk to the rank inside the bucket, neq to the bucket's count.
template <class SH>
__device__ __forceinline__ unsigned int radix_pick(SH& sh, unsigned int& kk, unsigned int& neq) {
  const int tid = threadIdx.x;
  __syncthreads();
  unsigned int cnt_b = 0, suf = 0;
  if (tid < 256) {
#pragma unroll
    for (int w = 0; w < 16; ++w) cnt_b += sh.hist[w][tid];
    suf = cnt_b;                                   // inclusive suffix within the wave: lanes >= lane
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const unsigned int up = __shfl_down(suf, o, 64);
      if ((tid & 63) + o < 64) suf += up;
    }
  }
  __syncthreads();                                 // all per-wave histograms consumed before hist[1] is reused for the wave totals
  if (tid < 256 && (tid & 63) == 0) sh.hist[1][tid >> 6] = suf;
  __syncthreads();
  if (tid < 256) {
    for (int w = (tid >> 6) + 1; w < 4; ++w) suf += sh.hist[1][w];
    const unsigned int above = suf - cnt_b;        // keys in strictly higher buckets
    if (suf >= kk && above < kk) { sh.bucket = tid; sh.kk = kk - above; sh.cnt = cnt_b; }
  }
  __syncthreads();
  const unsigned int bucket = sh.bucket;
  kk = sh.kk;
  neq = sh.cnt;
  __syncthreads();
  return bucket;
}

// exact radix select (4 x 8 bit) of the kk-th largest key of row[0..n): returns the key, the number of elements
// equal to it that belong to the top-kk (need_eq) and how many elements carry that key in total (neq).
template <class SH, bool COH = false>
__device__ uint32_t radix_select_kth(const float* __restrict__ row, int64_t n, unsigned int kk, SH& sh, unsigned int& need_eq,
                                     unsigned int& neq) {
  const int tid = threadIdx.x, wave = tid >> 6, NT = blockDim.x;   // (any block of >= 256 threads, at most 16 waves)
  uint32_t prefix = 0, mask = 0;
  const int64_t n4 = n >> 2;
  for (int shift = 24; shift >= 0; shift -= 8) {
    for (int i = tid; i < 16 * 256; i += NT) (&sh.hist[0][0])[i] = 0;
    __syncthreads();
    for (int64_t i = tid; i < n4; i += NT) {
      f32x4 v = ld4<COH>(row + 4 * i);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        uint32_t key = f2key(v[e]);
        if ((key & mask) == prefix) atomicAdd(&sh.hist[wave][(key >> shift) & 255], 1u);
      }
    }
    for (int64_t i = 4 * n4 + tid; i < n; i += NT) {
      uint32_t key = f2key(ld1<COH>(row + i));
      if ((key & mask) == prefix) atomicAdd(&sh.hist[wave][(key >> shift) & 255], 1u);
    }
    prefix |= (uint32_t)radix_pick(sh, kk, neq) << shift;
    mask |= 0xFFu << shift;
  }
  need_eq = kk;
  return prefix;
}

// the same for a SHORT row (n <= 4 * blockDim.x, n % 4 == 0, 16-byte aligned): every thread keeps its four keys in registers, so the four
// digit passes read nothing but their LDS histograms (the group maxima of a per-rank shard's sample: 3.9 k values -- k_sample_threshold
// 14.6 -> ~12 us)
template <class SH, bool COH = false>
__device__ uint32_t radix_select_kth_small(const float* __restrict__ row, int n, unsigned int kk, SH& sh) {
  const int tid = threadIdx.x, wave = tid >> 6, NT = blockDim.x;   // (any block of >= 256 threads, at most 16 waves)
  uint32_t key[4];
  const bool have = 4 * tid < n;
  if (have) {
    const f32x4 v = ld4<COH>(row + 4 * tid);
#pragma unroll
    for (int e = 0; e < 4; ++e) key[e] = f2key(v[e]);
  }
  uint32_t prefix = 0, mask = 0;
  unsigned int neq;
  for (int shift = 24; shift >= 0; shift -= 8) {
    for (int i = tid; i < 16 * 256; i += NT) (&sh.hist[0][0])[i] = 0;
    __syncthreads();
    if (have) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if ((key[e] & mask) == prefix) atomicAdd(&sh.hist[wave][(key[e] >> shift) & 255], 1u);
    }
    prefix |= (uint32_t)radix_pick(sh, kk, neq) << shift;
    mask |= 0xFFu << shift;
  }
  return prefix;
}

// the same over the score keys (upper halves) of a packed candidate list
template <class SH>
__device__ uint32_t radix_select_kth_list(const unsigned long long* __restrict__ list, int n, unsigned int kk, SH& sh) {
  const int tid = threadIdx.x, wave = tid >> 6, NT = blockDim.x;   // (any block of >= 256 threads, at most 16 waves)
  uint32_t prefix = 0, mask = 0;
  unsigned int neq;
  for (int shift = 24; shift >= 0; shift -= 8) {
    for (int i = tid; i < 16 * 256; i += NT) (&sh.hist[0][0])[i] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += NT) {
      const uint32_t key = (uint32_t)(list[i] >> 32);
      if ((key & mask) == prefix) atomicAdd(&sh.hist[wave][(key >> shift) & 255], 1u);
    }
    prefix |= (uint32_t)radix_pick(sh, kk, neq) << shift;
    mask |= 0xFFu << shift;
  }
  return prefix;
}

// Exact top-keff of row[0..N) (score desc, row asc) left sorted in sh.cand[0..keff); all SEL_THREADS threads take part.
__device__ void select_topk_sorted(const float* __restrict__ row, int64_t N, int keff, const float* __restrict__ bm, int nblk, SelShared& sh) {
  const int tid = threadIdx.x;
  int ncand = 0;  // number of valid entries in sh.cand, of which the best keff are the answer

  // ---- fast path: threshold = keff-th largest of the per-block maxima (>= keff elements are >= it, so the true top-keff
  //      all pass), then ONE scan of the row gathering everything >= threshold.
  bool done = false;
  if (N <= SEL_CAND) {
    for (int64_t i = tid; i < N; i += SEL_THREADS) sh.cand[i] = sel_pack(f2key(row[i]), i);
    ncand = (int)N;
    done = true;
  } else if (bm != nullptr && nblk >= keff) {
    unsigned int ne, nq_;
    const uint32_t thr = radix_select_kth(bm, nblk, keff, sh, ne, nq_);
    if (tid == 0) { sh.ngt = 0; sh.neq = 0; }
    __syncthreads();
    // only 128-row blocks whose maximum reaches the threshold can hold an element >= threshold: list them (normally ~keff
    // blocks), then scan just those instead of the whole row
    unsigned int* blist = (unsigned int*)sh.eqs;            // 2 * SEL_EQCAP entries (the tie buffers are idle on this path)
    for (int b = tid; b < nblk; b += SEL_THREADS)
      if (f2key(bm[b]) >= thr) {
        const unsigned int p = atomicAdd(&sh.neq, 1u);
        if (p < 2 * SEL_EQCAP) blist[p] = (unsigned int)b;
      }
    __syncthreads();
    const unsigned int nb = sh.neq;
    if (nb <= 2 * SEL_EQCAP) {
      for (unsigned int idx = tid; idx < nb * SP_ROWS; idx += SEL_THREADS) {
        const int64_t i = (int64_t)blist[idx >> 7] * SP_ROWS + (idx & (SP_ROWS - 1));
        if (i < N) {
          const uint32_t key = f2key(row[i]);
          if (key >= thr) {
            const unsigned int p = atomicAdd(&sh.ngt, 1u);
            if (p < SEL_CAND) sh.cand[p] = sel_pack(key, i);
          }
        }
      }
      __syncthreads();
      if (sh.ngt <= SEL_CAND) { ncand = (int)sh.ngt; done = true; }
    }
    __syncthreads();
  }

  if (!done) {
    // ---- exact path: radix select over the whole row, gather > kth (unordered) + the lowest-row-id ties
    unsigned int need_eq, neq;
    const uint32_t kth = radix_select_kth(row, N, keff, sh, need_eq, neq);
    const unsigned int ngt = keff - need_eq;
    if (tid == 0) { sh.ngt = 0; sh.neq = 0; }
    __syncthreads();
    const bool eq_fits = neq <= SEL_EQCAP;
    for (int64_t i = tid; i < N; i += SEL_THREADS) {
      uint32_t key = f2key(row[i]);
      if (key > kth) {
        unsigned int p = atomicAdd(&sh.ngt, 1u);
        sh.cand[p] = sel_pack(key, i);
      } else if (key == kth && eq_fits) {
        unsigned int p = atomicAdd(&sh.neq, 1u);
        sh.eqidx[p] = (uint32_t)i;
      }
    }
    __syncthreads();
    if (eq_fits) {
      int P = 1;
      while (P < (int)neq) P <<= 1;
      for (int i = tid; i < P; i += SEL_THREADS) sh.eqs[i] = i < (int)neq ? (unsigned long long)(0xFFFFFFFFu - sh.eqidx[i]) : 0ull;
      bitonic_sort_desc(sh.eqs, P);   // descending (~idx) == ascending row id
      for (int i = tid; i < (int)need_eq; i += SEL_THREADS) sh.cand[ngt + i] = ((unsigned long long)kth << 32) | sh.eqs[i];
    } else if (tid < 64) {
      // massive tie (degenerate data): ordered scan by one wave, lowest row ids first
      unsigned int taken = 0;
      for (int64_t base = 0; base < N && taken < need_eq; base += 64) {
        int64_t i = base + tid;
        bool hit = i < N && f2key(row[i]) == kth;
        unsigned long long bal = __ballot(hit);
        unsigned int before = __popcll(bal & ((1ull << tid) - 1ull));
        if (hit && taken + before < need_eq) sh.cand[ngt + taken + before] = sel_pack(kth, i);
        taken += __popcll(bal);
      }
    }
    __syncthreads();
    ncand = keff;
  }

  int P = 1;
  while (P < ncand) P <<= 1;
  __syncthreads();
  for (int i = ncand + tid; i < P; i += SEL_THREADS) sh.cand[i] = 0ull;
  bitonic_sort_desc(sh.cand, P);
}

__global__ void __launch_bounds__(SEL_THREADS)
k_topk_select(const float* __restrict__ scores, int64_t ld, int64_t N, int k, int64_t id_base, const float* __restrict__ blkmax, int nblk,
              int nblk_ld, float* __restrict__ out_scores, int64_t* __restrict__ out_ids, const int* __restrict__ gate,
              const int* __restrict__ qflags) {
  __shared__ SelShared sh;
  if (gate != nullptr && *gate == 0) return;                 // fallback launch of the bounded search: nothing overflowed
  if (qflags != nullptr && qflags[blockIdx.x] == 0) return;  // ... or not this query
  const float* row = scores + (int64_t)blockIdx.x * ld;
  float* os = out_scores + (int64_t)blockIdx.x * k;
  int64_t* oi = out_ids + (int64_t)blockIdx.x * k;
  const int tid = threadIdx.x;
  const int keff = (int)(N < (int64_t)k ? N : (int64_t)k);
  for (int i = keff + tid; i < k; i += SEL_THREADS) { os[i] = -FLT_MAX; oi[i] = -1; }
  if (keff == 0) return;
  select_topk_sorted(row, N, keff, blkmax ? blkmax + (int64_t)blockIdx.x * nblk_ld : nullptr, nblk, sh);
  for (int i = tid; i < keff; i += SEL_THREADS) {
    const unsigned long long c = sh.cand[i];
    os[i] = key2f((uint32_t)(c >> 32));
    oi[i] = id_base + sel_row(c);
  }
}

// q . x over D (multiple of 4) fp32 elements by one HALF-wave (32 lanes, two rows per wave in flight): fp64 accumulation of the exact
// fp32 products, one final rounding to fp32 -- the value every search path reports, so scores do not depend on the path, the query
// batch size or the shard layout.  Up to 16 row segments of 512 B are requested before the first is consumed (a row of 2048 floats
// is a single round trip; rescoring is latency-bound gather work).
// candidate rows are gathered once (random 8-KiB rows): non-temporal loads, k_refine_topk 119 -> 94 us at Q = 100 over 1M x 2048
#define REF_ROW_LOAD(p) __builtin_nontemporal_load(p)
__device__ __forceinline__ float exact_dot(const float* __restrict__ x, const float* __restrict__ qrow, int D, int lane) {
  const int sub = lane & 31;
  double acc = 0.0;
  for (int i0 = sub * 4; i0 < D; i0 += 2048) {
    f32x4 xv[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int i = i0 + u * 128;
      xv[u] = i < D ? REF_ROW_LOAD((const f32x4*)(x + i)) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int i = i0 + u * 128;
      if (i < D) {
        const f32x4 qv = *(const f32x4*)(qrow + i);
        acc += (double)xv[u][0] * (double)qv[0] + (double)xv[u][1] * (double)qv[1] + (double)xv[u][2] * (double)qv[2] +
               (double)xv[u][3] * (double)qv[3];
      }
    }
  }
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  return (float)acc;
}

// Select + final step of the plain path (and of the gated fallback of the bounded search) in one launch, RIGOROUS since round 3: the
// matrix scores s6 (six bf16 products, or the fp32 fma chain for <= 32 queries) differ from the exact inner product s by at most
//     eps6(q) = (6 D + 8) 2^-23 |q| R            R >= max |x_row|  (bounds[0])
// (dropped product terms mid*lo, lo*mid, lo*lo <= 2^-23 sum |q_i x_i|; at most 6 D fp32 accumulation steps, each within 2^-23 of a
// partial sum that is itself <= (1 + 2^-7) |q| |x|; Cauchy-Schwarz).  Every row of the exact top-k has s >= S_k >= kth6 - eps6 (k rows
// have s6 >= kth6), hence s6 >= kth6 - 2 eps6: ALL rows at or above that threshold are rescored exactly (fp64 accumulation, one rounding)
// and the best k of them returned.  Usually that is k + a few rows; a near-duplicate cluster with more than SEL_CAND rows inside the band
// takes the streaming form (the score row walked in 2048-row windows, a running exact top-k in LDS): slow (~ms per such query) but exact
// for any cluster size.  (Round 2 selected k + 64 rows by score: a heuristic that a stress run had already caught once.)
__device__ __forceinline__ float block_sum_1024(float v, float* red /* 16 */) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int w = 0; w < 16; ++w) t += red[w];
  return t;
}

__device__ __forceinline__ void select_rescore_query(const float* __restrict__ scores, int64_t ld, int64_t N, int k, int64_t id_base,
                                                     const float* __restrict__ blkmax, int nblk, int nblk_ld, const float* __restrict__ X, int64_t ldx,
                                                     int D, const float* __restrict__ q, float* __restrict__ out_scores, int64_t* __restrict__ out_ids,
                                                     const float* __restrict__ bounds, SelShared& sh, float* s_red) {
  const float* row = scores + (int64_t)blockIdx.x * ld;
  float* os = out_scores + (int64_t)blockIdx.x * k;
  int64_t* oi = out_ids + (int64_t)blockIdx.x * k;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int keff = (int)(N < (int64_t)k ? N : (int64_t)k);
  for (int i = keff + tid; i < k; i += SEL_THREADS) { os[i] = -FLT_MAX; oi[i] = -1; }
  if (keff == 0) return;
  const float* qrow = q + (int64_t)blockIdx.x * D;
  const float* bm = blkmax ? blkmax + (int64_t)blockIdx.x * nblk_ld : nullptr;
  float q2 = 0.f;
  for (int i = tid; i < D; i += SEL_THREADS) { const float v = qrow[i]; q2 += v * v; }
  q2 = block_sum_1024(q2, s_red);
  const float eps6 = (float)(6 * D + 8) * 1.1920929e-7f * sqrtf(q2) * bounds[0] * 1.01f;
  select_topk_sorted(row, N, keff, bm, nblk, sh);            // sh.cand[0..keff): the top-keff by matrix score, sorted
  const float kth6 = key2f((uint32_t)(sh.cand[keff - 1] >> 32));
  const float thr = kth6 - 2.0f * eps6;                      // (a non-finite query gives a NaN threshold: nothing qualifies below, the selection above stands)
  __syncthreads();
  unsigned long long* s_c = sh.cand;
  auto rescore = [&](unsigned long long* list, int n) {      // exact scores of list[0..n) in place: entry c is read and written by the same half-wave
    for (int c0 = wave * 2; c0 < n; c0 += 32) {
      const int c = min(c0 + (lane >> 5), n - 1);
      int64_t r = sel_row(list[c]);
      r = r < 0 ? 0 : (r >= N ? N - 1 : r);
      const float sc = exact_dot(X + r * ldx, qrow, D, lane);
      if ((lane & 31) == 0 && c0 + (lane >> 5) < n) list[c] = sel_pack(f2key(sc), r);
    }
  };
  // ---- every row with s6 >= thr: the qualifying 128-row blocks first (block maxima), then their rows
  unsigned int* blist = (unsigned int*)sh.eqs;               // 2 * SEL_EQCAP entries
  bool overflow = false;
  if (!(thr == thr)) {                                       // NaN band (non-finite query or bound): keep the score selection
    if (tid == 0) sh.ngt = (unsigned int)keff;
  } else if (N <= SEL_CAND || bm == nullptr) {
    if (tid == 0) sh.ngt = 0;
    __syncthreads();
    overflow = N > SEL_CAND;                                 // (no block maxima on a large row: straight to the streaming form)
    if (!overflow)
      for (int64_t i = tid; i < N; i += SEL_THREADS) {
        const float v = row[i];
        if (v >= thr) s_c[atomicAdd(&sh.ngt, 1u)] = sel_pack(f2key(v), i);
      }
  } else {
    if (tid == 0) { sh.ngt = 0; sh.neq = 0; }
    __syncthreads();
    for (int b = tid; b < nblk; b += SEL_THREADS)
      if (bm[b] >= thr) {
        const unsigned int p = atomicAdd(&sh.neq, 1u);
        if (p < 2 * SEL_EQCAP) blist[p] = (unsigned int)b;
      }
    __syncthreads();
    const unsigned int nb = sh.neq;
    overflow = nb > 2 * SEL_EQCAP;
    if (!overflow) {
      for (unsigned int idx = tid; idx < nb * SP_ROWS; idx += SEL_THREADS) {
        const int64_t i = (int64_t)blist[idx >> 7] * SP_ROWS + (idx & (SP_ROWS - 1));
        if (i < N) {
          const float v = row[i];
          if (v >= thr) {
            const unsigned int p = atomicAdd(&sh.ngt, 1u);
            if (p < SEL_CAND) s_c[p] = sel_pack(f2key(v), i);
          }
        }
      }
    }
  }
  __syncthreads();
  overflow = overflow || sh.ngt > SEL_CAND;
  __syncthreads();
  if (!overflow) {
    const int n = (int)sh.ngt;                               // >= keff: the keff selected rows are among them
    rescore(s_c, n);
    __syncthreads();
    int P = 1;
    while (P < n) P <<= 1;
    for (int i = n + tid; i < P; i += SEL_THREADS) s_c[i] = 0ull;
    bitonic_sort_desc(s_c, P);
  } else {
    // ---- streaming form: best[0..2048) = running exact top (sorted, 0-padded), chunk[0..2048) = the band rows of the current window
    unsigned long long* best = s_c;
    unsigned long long* chunk = s_c + SEL_MAXK;
    for (int i = tid; i < SEL_MAXK; i += SEL_THREADS) best[i] = 0ull;
    for (int64_t w0 = 0; w0 < N; w0 += SEL_MAXK) {
      if (tid == 0) sh.ngt = 0;
      __syncthreads();
#pragma unroll
      for (int j = 0; j < SEL_MAXK / SEL_THREADS; ++j) {
        const int64_t i = w0 + tid + j * SEL_THREADS;
        if (i < N) {
          const float v = row[i];
          if (v >= thr) chunk[atomicAdd(&sh.ngt, 1u)] = sel_pack(f2key(v), i);
        }
      }
      __syncthreads();
      const int n = (int)sh.ngt;                             // (wave-uniform for everybody: read after the barrier)
      if (n == 0) continue;
      rescore(chunk, n);
      __syncthreads();
      for (int i = n + tid; i < SEL_MAXK; i += SEL_THREADS) chunk[i] = 0ull;
      bitonic_sort_desc(best, 2 * SEL_MAXK);                 // merge: the best SEL_MAXK (>= keff) of best + chunk stay in front
      __syncthreads();
    }
  }
  for (int i = tid; i < keff; i += SEL_THREADS) {
    const unsigned long long c = s_c[i];
    os[i] = key2f((uint32_t)(c >> 32));
    oi[i] = id_base + sel_row(c);
  }
}

// The last kernel of every bounded search (one workgroup per query).  gate / qflags: the exact fallback runs only for a flagged query of
// a chunk in which something overflowed.  wire (round 4, optional): the query's k results -- whoever wrote them, this workgroup or
// k_refine_merge one launch earlier -- also leave as the 64-bit words of the multi-GPU exchange (lrx_pack_topk's format; row_map as
// there), so a sharded search needs no packing launch between the local search and the all-gather.
__global__ void __launch_bounds__(SEL_THREADS)
k_topk_select_rescore(const float* __restrict__ scores, int64_t ld, int64_t N, int k, int64_t id_base, const float* __restrict__ blkmax, int nblk,
                      int nblk_ld, const float* __restrict__ X, int64_t ldx, int D, const float* __restrict__ q, float* __restrict__ out_scores,
                      int64_t* __restrict__ out_ids, const int* __restrict__ gate, const int* __restrict__ qflags, const float* __restrict__ bounds,
                      unsigned long long* __restrict__ wire, const int64_t* __restrict__ row_map) {
  __shared__ SelShared sh;
  __shared__ float s_red[16];
  const bool idle = (gate != nullptr && *gate == 0) ||                  // fallback launch of the bounded search: nothing overflowed
                    (qflags != nullptr && qflags[blockIdx.x] == 0);     // ... or not this query
  if (!idle) select_rescore_query(scores, ld, N, k, id_base, blkmax, nblk, nblk_ld, X, ldx, D, q, out_scores, out_ids, bounds, sh, s_red);
  if (wire == nullptr) return;
  __syncthreads();                                            // (this workgroup's own stores of the rows it is about to read)
  const float* os = out_scores + (int64_t)blockIdx.x * k;
  const int64_t* oi = out_ids + (int64_t)blockIdx.x * k;
  for (int i = threadIdx.x; i < k; i += SEL_THREADS) {
    int64_t id = oi[i];
    if (id >= 0 && row_map != nullptr) id = row_map[id - id_base];
    wire[(int64_t)blockIdx.x * k + i] = ((unsigned long long)__float_as_uint(os[i]) << 32) | (unsigned long long)(id >= 0 ? (uint32_t)id : 0xFFFFFFFFu);
  }
}

