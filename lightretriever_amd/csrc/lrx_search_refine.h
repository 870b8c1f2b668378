// liblrx search, part 4 -- REFINE: exact rescoring of the band rows (per query and grouped by row), k_refine_topk, k_refine_merge (map: section C).
// Part of the ONE translation unit lrx_search.hip (included there, in source order: filter kernels -> selection -> bounded-search
// device code -> refine kernels; the host driver, the shard maintenance and the exchange kernels stay in lrx_search.hip).  Not a stand-alone header.
#pragma once

// exact rescoring of nc candidate rows (s_cand: row numbers): one half-wave per row (fp64 accumulation of the fp32 products, one
// rounding to fp32); the packed (score, row) pairs go to `mine`.  (A version that streams the rows as 1024-float chunks through two
// register buffers, the next chunk requested before the current one is accumulated, changed nothing: the step is bound by the chip's
// random 8-KiB gather rate, 0.30 GB in ~58 us = 5.2 TB/s at Q = 100, 5.8 TB/s at Q = 256.)
__device__ __forceinline__ void refine_rescore(const float* __restrict__ X, int64_t ldx, int D, const float* qrow, const unsigned long long* s_cand,
                                               int nc, unsigned long long* __restrict__ mine) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int c0 = wave * 2; c0 < nc; c0 += 32) {
    const int c = min(c0 + (lane >> 5), nc - 1);
    const int64_t n = (int64_t)s_cand[c];
    const float sc = exact_dot(X + n * ldx, qrow, D, lane);
    if ((lane & 31) == 0 && c0 + (lane >> 5) < nc) mine[c] = sel_pack(f2key(sc), n);
  }
}

// Refine step of the score-free filter, grid (n_queries, REF_SPLIT): every part finds kth~ in the query's candidate list (radix
// select over ~10^3..10^4 L2-resident entries), takes every REF_SPLIT-th entry of the band [kth~ - 2 eps, inf), rescores those rows
// exactly and publishes the packed (score, row) list (count -1 = list or band overflow); k_refine_merge finishes.
struct RowPairs {                         // row-grouped rescoring (below): NULL pairs = the gather of refine_rescore
  unsigned long long* pairs;              // [n_queries * REF_CAND] (row << 32 | slot in `parts`), in emission order
  unsigned int* total;                    // number of pairs emitted
  unsigned int* grp_cnt;                  // [groups] pairs per group of (1 << grp_shift) rows
  int grp_shift;
};
__global__ void __launch_bounds__(1024)
k_refine_band(const float* __restrict__ X, int64_t N, int64_t ldx, int D, const float* __restrict__ q, const unsigned long long* __restrict__ cand,
              const unsigned int* __restrict__ cnt, const float* __restrict__ eps, int k, unsigned long long* __restrict__ parts,
              int* __restrict__ part_cnt, int nsplit, unsigned int cap, RowPairs rp) {
  __shared__ RadixShared rs;
  __shared__ unsigned long long s_cand[REF_CAND];           // (a part holds REF_CAND / nsplit of them)
  __shared__ __attribute__((aligned(16))) float s_q[REF_QLDS];   // the query row (every rescoring re-reads it; from global its loads serialise)
  __shared__ unsigned int s_ncand;
  const int tid = threadIdx.x;
  const int qi = blockIdx.x, part = blockIdx.y;
  const unsigned int pcand = REF_CAND / nsplit;             // candidate capacity of one part
  const float* qglob = q + (int64_t)qi * D;
  const float* qrow = D <= REF_QLDS ? s_q : qglob;
  if (D <= REF_QLDS)
    for (int i = tid; i < D; i += 1024) s_q[i] = qglob[i];
  if (tid == 0) s_ncand = 0;
  const unsigned int n = cnt[qi * CNT_STRIDE];
  bool overflow = n > cap || n < (unsigned int)k;      // (n < k cannot happen with a finite threshold: k sample rows reach it)
  __syncthreads();
  if (!overflow) {
    const unsigned long long* list = cand + (int64_t)qi * cap;
    // (Round 4, measured and not kept: the k-th score of lists of <= 1024 entries by counting -- every thread one key, ranks from broadcast LDS
    // reads, no barrier-separated passes: 32.1-33.0 us against 28.7-29.0 for this kernel on the 125 k-row shard, same box, three runs each:
    // its lists hold ~800 entries there and the O(n^2 / threads) walk loses to four radix passes.)
    const float kth = key2f(radix_select_kth_list(list, (int)n, (unsigned int)k, rs));
    const float thr = kth - 2.0f * eps[qi];
    for (int i = part + nsplit * tid; i < (int)n; i += nsplit * 1024) {
      const unsigned long long e = list[i];
      if (key2f((uint32_t)(e >> 32)) >= thr) {
        const int64_t row = sel_row(e);
        // never index outside the shard, whatever the list holds: a row that cannot exist sends the query to the exact fallback
        const unsigned int p = row < N ? atomicAdd(&s_ncand, 1u) : atomicAdd(&s_ncand, (unsigned int)REF_CAND + 1u);
        if (p < pcand) s_cand[p] = (unsigned long long)row;
      }
    }
    __syncthreads();
    overflow = s_ncand > pcand;
  }
  const int nc = overflow ? 0 : (int)s_ncand;
  if (rp.pairs != nullptr) {
    // row-grouped rescoring: this part only NAMES its band rows -- (row, slot of `parts` the exact score goes to) -- and counts them per row group
    __shared__ unsigned int s_base;
    if (tid == 0) s_base = nc > 0 ? atomicAdd(rp.total, (unsigned int)nc) : 0u;
    __syncthreads();
    const unsigned int slot0 = (unsigned int)((qi * nsplit + part) * (int)pcand);
    for (int c = tid; c < nc; c += 1024) {
      const unsigned long long row = s_cand[c];
      rp.pairs[s_base + c] = (row << 32) | (unsigned long long)(slot0 + (unsigned int)c);
      atomicAdd(&rp.grp_cnt[row >> rp.grp_shift], 1u);
    }
  } else {
    refine_rescore(X, ldx, D, qrow, s_cand, nc, parts + ((int64_t)qi * nsplit + part) * pcand);
  }
  if (tid == 0) part_cnt[qi * nsplit + part] = overflow ? -1 : nc;
}

// ---- Row-grouped exact rescoring (round 5): many queries x large k over a small shard (the reference's evaluation point: top-1000 of ~1000
// queries per 100 k-row corpus chunk, eval/call_evaluate_mteb.sh:8-10) want every fp32 row several times -- 250 queries x 1 210 band rows
// over 100 k rows: three times -- and the per-query gather above reads it from HBM each time (2.5 GB per chunk of 250 queries against a
// 0.8-GB shard).  Here the (row, slot) pairs the parts emitted are grouped by 16-row group (counting sort: the counts came with the
// pairs), a workgroup stages its group's rows in LDS once and streams the query rows of its pairs from L2 through the same fp64 dot
// product (same association, same bits as exact_dot).
__global__ void __launch_bounds__(1024)
k_pairs_scan(const unsigned int* __restrict__ grp_cnt, unsigned int* __restrict__ grp_off, int ngroups) {   // exclusive scan of ngroups + 1 entries, one workgroup
  __shared__ unsigned int s_w[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // every thread owns a run of consecutive groups (one pass, two barriers, whatever the group count)
  const int per = (ngroups + 1 + 1023) / 1024;
  const int g0 = tid * per, g1 = min(g0 + per, ngroups + 1);
  unsigned int mine = 0;
  for (int g = g0; g < g1; ++g) mine += g < ngroups ? grp_cnt[g] : 0u;
  unsigned int x = mine;                                     // inclusive scan of the threads' totals inside the wave
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const unsigned int y = __shfl_up(x, o, 64); if (lane >= o) x += y; }
  if (lane == 63) s_w[wave] = x;
  __syncthreads();
  unsigned int run = x - mine;
  for (int w = 0; w < wave; ++w) run += s_w[w];
  for (int g = g0; g < g1; ++g) {
    grp_off[g] = run;
    run += g < ngroups ? grp_cnt[g] : 0u;
  }
}
__global__ void __launch_bounds__(256)
k_pairs_scatter(const unsigned long long* __restrict__ pairs, const unsigned int* __restrict__ total, unsigned int* __restrict__ grp_cnt,
                const unsigned int* __restrict__ grp_off, int grp_shift, unsigned long long* __restrict__ sorted) {
  const unsigned int n = *total;
  for (unsigned int i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
    const unsigned long long pr = pairs[i];
    const unsigned int g = (unsigned int)(pr >> 32) >> grp_shift;
    const unsigned int pos = grp_off[g] + atomicSub(&grp_cnt[g], 1u) - 1u;      // (leaves the counts at zero for the next chunk)
    sorted[pos] = pr;
  }
}
// exact_dot with the row in LDS and the query row in global memory: the SAME partial products in the same order (lane sub of a half-wave:
// elements i0 + 128 u + (0..3), i0 = 4 sub, 2048-element blocks), fp64 accumulation, the same xor tree -- bit-identical to exact_dot
__device__ __forceinline__ float exact_dot_lds_row(const float* x_lds, const float* __restrict__ qglob, int D, int lane) {
  const int sub = lane & 31;
  double acc = 0.0;
  for (int i0 = sub * 4; i0 < D; i0 += 2048) {
    f32x4 qv[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int i = i0 + u * 128;
      qv[u] = i < D ? *(const f32x4*)(qglob + i) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int i = i0 + u * 128;
      if (i < D) {
        const f32x4 xv = *(const f32x4*)(x_lds + i);
        acc += (double)xv[0] * (double)qv[u][0] + (double)xv[1] * (double)qv[u][1] + (double)xv[2] * (double)qv[u][2] +
               (double)xv[3] * (double)qv[u][3];
      }
    }
  }
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  return (float)acc;
}
#define ROWGRP_LDS_FLOATS 16384          // 64 KiB of rows per workgroup (two workgroups per CU: one stages while the other multiplies): 8 rows at D = 2048, 4 at 4096
#define ROWGRP_THREADS 512
__global__ void __launch_bounds__(ROWGRP_THREADS)
k_rescore_row_groups(const float* __restrict__ X, int64_t N, int64_t ldx, int D, const float* __restrict__ q, const unsigned long long* __restrict__ sorted,
                     const unsigned int* __restrict__ grp_off, int grp_shift, unsigned long long* __restrict__ parts) {
  __shared__ __attribute__((aligned(16))) float s_x[ROWGRP_LDS_FLOATS];
  const int tid = threadIdx.x, lane = tid & 63, hw = tid >> 5;
  const int g = blockIdx.x;
  const unsigned int p0 = grp_off[g], p1 = grp_off[g + 1];
  if (p0 == p1) return;
  const int64_t r0 = (int64_t)g << grp_shift;
  const int nrows = (int)min((int64_t)1 << grp_shift, N - r0);
  for (int i = tid * 4; i < nrows * D; i += ROWGRP_THREADS * 4) {
    const int rr = i / D, cc = i - rr * D;
    *(f32x4*)(s_x + i) = REF_ROW_LOAD((const f32x4*)(X + (r0 + rr) * ldx + cc));
  }
  __syncthreads();
  for (unsigned int pi = p0 + hw; pi < p1; pi += ROWGRP_THREADS / 32) {
    const unsigned long long pr = sorted[pi];
    const int64_t row = (int64_t)(pr >> 32);
    const unsigned int slot = (unsigned int)pr;
    const float sc = exact_dot_lds_row(s_x + (row - r0) * D, q + (int64_t)(slot / REF_CAND) * D, D, lane);
    if ((lane & 31) == 0) parts[slot] = sel_pack(f2key(sc), row);
  }
}

// Refine step of the score-matrix filter, grid (n_queries, REF_SPLIT): part s of query q owns the 128-row blocks b with b % REF_SPLIT == s:
// it gathers their rows inside the band from the score matrix, rescores them exactly and publishes the packed (score, row) list
// (count -1 = the part's lists overflowed); k_refine_merge finishes.
__global__ void __launch_bounds__(1024)
k_refine_topk(const float* __restrict__ X, int64_t N, int64_t ldx, int D, const float* __restrict__ q, const float* __restrict__ scores, int64_t ld,
              const float* __restrict__ blkmax, int nblk, int nblk_ld, const float* __restrict__ bounds, int k, int64_t id_base,
              const float* __restrict__ out_scores, unsigned long long* __restrict__ parts, int* __restrict__ part_cnt) {
  __shared__ unsigned long long s_cand[REF_PCAND];
  __shared__ unsigned int s_blk[REF_PBLK];
  __shared__ __attribute__((aligned(16))) float s_q[REF_QLDS];
  __shared__ float s_red[32];
  __shared__ unsigned int s_nblk, s_ncand;
  const int tid = threadIdx.x;
  const int qi = blockIdx.x, part = blockIdx.y;
  const float* os = out_scores + (int64_t)qi * k;
  const int keff = (int)(N < (int64_t)k ? N : (int64_t)k);
  if (tid == 0) { s_nblk = 0; s_ncand = 0; }
  if (keff == 0) return;                      // (outputs already padded by k_topk_select)
  const float* qglob = q + (int64_t)qi * D;
  const float* qrow = D <= REF_QLDS ? s_q : qglob;
  const float band = 2.0f * query_eps_block(qglob, D, bounds, D <= REF_QLDS ? s_q : nullptr, s_red);
  const float kth = os[keff - 1];             // k-th largest filter score (written by k_topk_select; nobody writes os before the merge)
  const float thr = kth - band;
  // this part's qualifying 128-row blocks
  const float* bm = blkmax + (int64_t)qi * nblk_ld;
  for (int b = part + REF_SPLIT * tid; b < nblk; b += REF_SPLIT * 1024)
    if (bm[b] >= thr) {
      const unsigned int p = atomicAdd(&s_nblk, 1u);
      if (p < REF_PBLK) s_blk[p] = (unsigned int)b;
    }
  __syncthreads();
  const unsigned int nb = s_nblk;
  bool overflow = nb > REF_PBLK;
  if (!overflow) {
    const float* row = scores + (int64_t)qi * ld;
    const unsigned int total = nb * SP_ROWS;
    for (unsigned int idx0 = tid; idx0 < total; idx0 += 4 * 1024) {     // four independent loads in flight per thread
      float v[4];
      int64_t n[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const unsigned int idx = idx0 + u * 1024;
        n[u] = idx < total ? (int64_t)s_blk[idx >> 7] * SP_ROWS + (idx & (SP_ROWS - 1)) : N;
        v[u] = n[u] < N ? row[n[u]] : -FLT_MAX;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (n[u] < N && v[u] >= thr) {
          const unsigned int p = atomicAdd(&s_ncand, 1u);
          if (p < REF_PCAND) s_cand[p] = (unsigned long long)n[u];
        }
    }
    __syncthreads();
    overflow = s_ncand > REF_PCAND;
  }
  const int nc = overflow ? 0 : (int)s_ncand;
  refine_rescore(X, ldx, D, qrow, s_cand, nc, parts + ((int64_t)qi * REF_SPLIT + part) * REF_PCAND);
  if (tid == 0) part_cnt[qi * REF_SPLIT + part] = overflow ? -1 : nc;
}

// Queries the bounded search sent to its exact six-product fallback (candidate list or band overflow: near-duplicate clusters, rows outside
// fp16's range) since the last reset -- a performance event, not an error: read by lrx_search_fallback_count.
__device__ unsigned int g_search_fallback_queries = 0;
extern "C" int64_t lrx_search_fallback_count(int32_t reset) {
  unsigned int v = 0;
  if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_search_fallback_queries), sizeof(v)) != hipSuccess) return -1;
  if (reset && v) {
    const unsigned int z = 0;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_search_fallback_queries), &z, sizeof(z)) != hipSuccess) return -1;
  }
  return (int64_t)v;
}

// Merge of the REF_SPLIT published lists of a query (one workgroup per query; the kernel boundary orders it after the refine kernel --
// an in-kernel "last part merges" ticket needed device-scope fences that cost more than this launch): sort, write the top-k; a part
// that overflowed flags the query for the gated six-product fallback.
__global__ void __launch_bounds__(1024)
k_refine_merge(const unsigned long long* __restrict__ parts, const int* __restrict__ part_cnt, int64_t N, int k, int64_t id_base,
               float* __restrict__ out_scores, int64_t* __restrict__ out_ids, int* __restrict__ qflags, int* __restrict__ any_flag, int nsplit) {
  __shared__ unsigned long long s_cand[REF_CAND];
  const int tid = threadIdx.x, qi = blockIdx.x;
  const int pcand = REF_CAND / nsplit;
  float* os = out_scores + (int64_t)qi * k;
  int64_t* oi = out_ids + (int64_t)qi * k;
  const int keff = (int)(N < (int64_t)k ? N : (int64_t)k);
  if (keff == 0) return;
  int cnt[REF_SPLIT], tot = 0;                 // nsplit <= REF_SPLIT
  bool any_over = false;
#pragma unroll
  for (int p = 0; p < REF_SPLIT; ++p) {
    cnt[p] = p < nsplit ? part_cnt[qi * nsplit + p] : 0;
    any_over |= cnt[p] < 0;
    tot += cnt[p] < 0 ? 0 : cnt[p];
  }
  if (any_over || tot < keff) {               // (tot < keff: a non-finite query or threshold -- the exact path sorts it out)
    if (tid == 0) { qflags[qi] = 1; atomicOr(any_flag + (qi >> 7), 1); atomicAdd(&g_search_fallback_queries, 1u); }   // (the flag of the query's 128-query group)
    return;
  }
  int base = 0;
#pragma unroll
  for (int p = 0; p < REF_SPLIT; ++p) {
    const unsigned long long* src = parts + ((int64_t)qi * nsplit + p) * pcand;
    for (int i = tid; i < cnt[p]; i += blockDim.x) s_cand[base + i] = src[i];
    base += cnt[p];
  }
  __syncthreads();
  if (tot <= (int)blockDim.x) {
    // the usual case, a few hundred band rows: rank by counting (the packed (score, row) words are distinct, so the ranks are the sorted
    // positions; every thread walks the list with broadcast LDS reads -- no barrier-separated sort stages: 14 -> 12 us).  (Measured for the
    // ~1200 entries of top_k = 1000 with two entries per thread: 52 us against 33 for the 2048-entry bitonic sort -- not extended.)
    if (tid < tot) {
      const unsigned long long me = s_cand[tid];
      int r = 0;
      int j = 0;
      for (; j + 4 <= tot; j += 4)
        r += (s_cand[j] > me ? 1 : 0) + (s_cand[j + 1] > me ? 1 : 0) + (s_cand[j + 2] > me ? 1 : 0) + (s_cand[j + 3] > me ? 1 : 0);
      for (; j < tot; ++j) r += s_cand[j] > me ? 1 : 0;
      if (r < keff) {
        os[r] = key2f((uint32_t)(me >> 32));
        oi[r] = id_base + sel_row(me);
      }
    }
    return;
  }
  int P = 1;
  while (P < tot) P <<= 1;
  for (int i = tot + tid; i < P; i += blockDim.x) s_cand[i] = 0ull;
  if (P == 2 * (int)blockDim.x) bitonic_sort_desc_regs<2>(s_cand, P);          // (the sorts load after their own barrier: the zero fill above is seen)
  else if (P == 4 * (int)blockDim.x) bitonic_sort_desc_regs<4>(s_cand, P);
  else bitonic_sort_desc(s_cand, P);
  for (int i = tid; i < keff; i += blockDim.x) {
    const unsigned long long c = s_cand[i];
    os[i] = key2f((uint32_t)(c >> 32));
    oi[i] = id_base + sel_row(c);
  }
}

