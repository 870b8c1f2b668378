// Flat inner-product search over an HBM-resident fp32 corpus shard (replaces faiss.IndexFlatIP.search).  Map of this file, in source order:
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
//                                     (129..256 queries: the main pass runs on the GEMM kernel, lrx_gemm.hip EPI_EMIT.)
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

// The single-product FILTER of the bounded search runs in FP16 since round 3 (shadow rows, query planes, MFMA 16x16x32 f16): 11
// significant bits instead of bf16's 8 make the rigorous error band ~5x narrower (|x - fp16(x)| <= 2^-11 |x| element-wise), i.e. ~5x
// fewer band rows to rescore and half the candidate-list entries.  16-bit containers stay typed bf16x8 (they are moved, not computed
// on); conversions saturate at +-65504 so a value outside fp16's range shows up as a large measured rounding error E (-> huge band ->
// exact fallback), never as inf / NaN.  The six-product exact path keeps its bf16 hi/mid/lo split (exact for any fp32 value).
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
__device__ __forceinline__ f32x4 mfma_f16(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ _Float16 f2h_sat(float v) { return (_Float16)fminf(fmaxf(v, -65504.f), 65504.f); }
__device__ __forceinline__ __bf16 f2h_bits(float v) { return __builtin_bit_cast(__bf16, f2h_sat(v)); }

#define S_ROWS 256       // corpus rows per workgroup
#define S_BK 32          // floats per k-slice (128 B per row)
#define S_XTILE (S_ROWS * S_BK * 4)
#define SP_ROWS 128      // corpus rows per workgroup of the split-bf16 kernel

template <int QT>
__global__ void __launch_bounds__(256, 1)
k_flat_ip_scores(const float* __restrict__ X, int64_t N, int64_t ldx, int D, const float* __restrict__ Q, int nq,
                 float* __restrict__ scores, int64_t ld, float* __restrict__ blkmax, int nblk, const int* __restrict__ gate) {
  constexpr int QTILE = QT * 16 * S_BK * 4;
  constexpr int STAGE = S_XTILE + QTILE;
  __shared__ __attribute__((aligned(1024))) char smem[2 * STAGE];
  if (gate != nullptr && *gate == 0) return;     // fallback launch of the bounded search: nothing overflowed
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t n0 = (int64_t)blockIdx.x * S_ROWS;

  // staging sources.  X: 32 wave instructions per tile (8 per wave); q: 2*QT instructions (round-robin over waves)
  const float* px[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    int s = (wave * 8 + i) * 64 + lane;
    int row = s >> 3, c = (s & 7) ^ ((row >> 1) & 7);
    int64_t g = min(n0 + row, N - 1);
    px[i] = X + g * ldx + c * 4;
  }
  constexpr int QI = (2 * QT + 3) / 4;  // q instructions per wave (upper bound)
  const float* pq[QI];
#pragma unroll
  for (int i = 0; i < QI; ++i) {
    int j = wave + 4 * i;
    int s = j * 64 + lane;
    int row = s >> 3, c = (s & 7) ^ ((row >> 1) & 7);
    int g = min(row, nq - 1);
    pq[i] = Q + (int64_t)g * D + c * 4;
  }
  auto stage = [&](int st, int k0) {
    char* sX = smem + st * STAGE;
    char* sQ = sX + S_XTILE;
#pragma unroll
    for (int i = 0; i < 8; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(px[i] + k0), (lptr_t)(sX + (wave * 8 + i) * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < QI; ++i) {
      int j = wave + 4 * i;
      if (j < 2 * QT) __builtin_amdgcn_global_load_lds((gptr_t)(pq[i] + k0), (lptr_t)(sQ + j * 1024), 16, 0, 0);
    }
  };

  const int fi = lane & 15, fg = lane >> 4;
  int loff[2];
#pragma unroll
  for (int kb = 0; kb < 2; ++kb) loff[kb] = fi * 128 + (((kb * 4 + fg) ^ (fi >> 1)) << 4);

  f32x4 acc[4][QT];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < QT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = D / S_BK;
  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) stage(cur ^ 1, (kt + 1) * S_BK);
    const char* sX = smem + cur * STAGE + (wave * 64) * 128;
    const char* sQ = smem + cur * STAGE + S_XTILE;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      f32x4 xf[4], qf[QT];
#pragma unroll
      for (int a = 0; a < 4; ++a) xf[a] = *(const f32x4*)(sX + a * 2048 + loff[kb]);
#pragma unroll
      for (int b = 0; b < QT; ++b) qf[b] = *(const f32x4*)(sQ + b * 2048 + loff[kb]);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < QT; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(xf[a][t], qf[b][t], acc[a][b], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // D[i = corpus row][j = query]: lane holds query j = fi, rows fg*4 + {0..3}.  Also the per-(query, 256-row block)
  // maximum, which gives k_topk_select a safe threshold without an extra pass over the scores.
  float* wmax = (float*)smem;  // [4 waves][QT*16]   (LDS is free: the k loop ended with a barrier)
#pragma unroll
  for (int b = 0; b < QT; ++b) {
    int qi = b * 16 + fi;
    float mx = -FLT_MAX;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      int64_t n = n0 + wave * 64 + a * 16 + fg * 4;
      f32x4 v = acc[a][b];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (n + e >= N) v[e] = -FLT_MAX;
        mx = fmaxf(mx, v[e]);
      }
      if (qi < nq) *(f32x4*)(scores + (int64_t)qi * ld + n) = v;
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    if (fg == 0) wmax[wave * (QT * 16) + qi] = mx;
  }
  __syncthreads();
  // maxima at 128-row granularity (two per workgroup) so both score kernels feed k_topk_select the same layout
  if (blkmax != nullptr && tid < 2 * QT * 16) {
    const int hh = tid / (QT * 16), qq = tid - hh * (QT * 16);
    if (qq < nq) blkmax[(int64_t)qq * nblk + 2 * blockIdx.x + hh] = fmaxf(wmax[(2 * hh) * QT * 16 + qq], wmax[(2 * hh + 1) * QT * 16 + qq]);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Q > 32: split-bf16 score kernel.  Each fp32 value v is written EXACTLY as hi + mid + lo (three bf16, 24 mantissa bits);
// q . x = sum over the six products hh, hm, mh, mm, hl, lh (the three dropped ones are <= 2^-24 relative, i.e. below the
// rounding of an fp32 product), each product exact in the bf16 MFMA's fp32 accumulation -> fp32-grade scores at 6/16 of
// the fp32-matrix time, which brings Q = 100 from fp32-MFMA-bound to (nearly) HBM-bound.  X is split in registers by the
// wave that owns the rows (each element once); the queries are split once per search by k_split_queries into fragment
// order so every q fragment is one linear 1-KiB LDS-DMA + one linear ds_read_b128.
// k permutation inside a 32-wide slice (same on both operands): element j of lane group fq is k = 4fq + j (j < 4) or
// 16 + 4fq + (j - 4): keeps both 16-B X reads of a lane conflict-free under the 128-B-row swizzle.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void split3(float v, __bf16& h, __bf16& m, __bf16& l) {
  h = (__bf16)v;
  float r1 = v - (float)h;
  m = (__bf16)r1;
  float r2 = r1 - (float)m;
  l = (__bf16)r2;
}

// qs layout: [D/32 slices][NP planes][QT][64 lanes][8] 16-bit   (NP = 3: bf16 hi/mid/lo, NP = 1: fp16(q))
__device__ __forceinline__ void split_queries_body(const float* __restrict__ Q, int nq, int D, int QT, int NP, __bf16* __restrict__ qs, int gid) {
  int lane = gid & 63, rest = gid >> 6;
  int qt = rest % QT, kt = rest / QT;
  if (kt >= D / 32) return;
  int fi = lane & 15, fq = lane >> 4;
  int row = qt * 16 + fi;
  bf16x8 h, m, l;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    int k = kt * 32 + (j < 4 ? 4 * fq + j : 16 + 4 * fq + (j - 4));
    float v = row < nq ? Q[(int64_t)row * D + k] : 0.f;
    __bf16 a, b, c;
    split3(v, a, b, c);
    h[j] = NP == 1 ? f2h_bits(v) : a; m[j] = b; l[j] = c;      // one plane = the fp16 filter operand; three = the exact bf16 split
  }
  int64_t base = (((int64_t)kt * NP) * QT + qt) * 64 + lane;
  bf16x8* out = (bf16x8*)qs;
  out[base] = h;
  if (NP == 3) {
    out[base + (int64_t)QT * 64] = m;
    out[base + 2 * (int64_t)QT * 64] = l;
  }
}
__global__ void k_split_queries(const float* __restrict__ Q, int nq, int D, int QT, int NP, __bf16* __restrict__ qs, const int* __restrict__ gate) {
  if (gate != nullptr && *gate == 0) return;
  split_queries_body(Q, nq, D, QT, NP, qs, blockIdx.x * blockDim.x + threadIdx.x);
}

// Planes of the gated six-product fallback of a bounded search, written ahead of time by extra workgroups of the query-packing kernel (the
// first launch of the chain) instead of by a launch of their own behind the gate: groups of <= 128 queries, qs3 + g * stride each.
struct PreSplit {
  __bf16* qs3 = nullptr;
  int64_t stride = 0;                     // elements between the groups' plane sets
  int ngroups = 0, nb_xb = 0;             // nb_xb: workgroups of the packing proper
  int nf[2] = {0, 0}, qt[2] = {0, 0}, blocks[2] = {0, 0};
};

// shadow filter: natural k order, 64-wide slices, fp16.  qs layout: [D/64 slices][2 k-steps][QT][64 lanes][8] fp16, lane (fi = query in
// tile, fq) of k-step ks holds k = slice*64 + ks*32 + fq*8 .. +7 (the MFMA 16x16x32 operand layout).
// zero / nzero: ints cleared on the way (the flags and list counters of a bounded search: this is the first kernel of its chain, so the
// clear needs no launch of its own)
__global__ void k_pack_queries_xb(const float* __restrict__ Q, int nq, int D, int QT, __bf16* __restrict__ qs, int* __restrict__ zero, int nzero,
                                  PreSplit ps) {
  if (ps.ngroups > 0 && (int)blockIdx.x >= ps.nb_xb) {
    int b = (int)blockIdx.x - ps.nb_xb, g = 0;
    if (b >= ps.blocks[0]) { b -= ps.blocks[0]; g = 1; }
    split_queries_body(Q + (int64_t)g * 128 * D, ps.nf[g], D, ps.qt[g], 3, ps.qs3 + g * ps.stride, b * blockDim.x + threadIdx.x);
    return;
  }
  const int nb = ps.ngroups > 0 ? ps.nb_xb : (int)gridDim.x;
  int gid = blockIdx.x * blockDim.x + threadIdx.x;
  for (int i = gid; i < nzero; i += nb * blockDim.x) zero[i] = 0;
  int lane = gid & 63, rest = gid >> 6;
  int qt = rest % QT, r2 = rest / QT;
  int ks = r2 & 1, sl = r2 >> 1;
  if (sl >= D / 64) return;
  int fi = lane & 15, fq = lane >> 4;
  int row = qt * 16 + fi;
  bf16x8 h;
#pragma unroll
  for (int j = 0; j < 8; ++j) h[j] = f2h_bits(row < nq ? Q[(int64_t)row * D + sl * 64 + ks * 32 + fq * 8 + j] : 0.f);
  ((bf16x8*)qs)[(((int64_t)sl * 2 + ks) * QT + qt) * 64 + lane] = h;
}

// plain fp16 copy of the queries [nq, D] (RNE, saturating): the B operand of the 256-query filter pass on the GEMM kernel
__global__ void k_round_queries(const float* __restrict__ Q, int64_t n, __bf16* __restrict__ q16) {
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i >= n) return;
  const f32x4 v = *(const f32x4*)(Q + i);
  bf16x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = f2h_bits(v[e]);
  *(bf16x4*)(q16 + i) = o;
}

// RT = 16-row tiles per wave (rows per workgroup = 16 RT WV), NST = LDS stages of the k-slice ring.  The six-product kernel runs
// (RT 2, NST 2, 4 waves, two workgroups per CU: it is bound by the matrix pipe); the single-product filter over fp32 rows (a shard without
// a shadow: rows converted to fp16 in registers) is HBM-bound and runs the deeper / wider shape selected by SPF_RT / SPF_NST.
#ifndef SPF_RT
#define SPF_RT 2
#endif
#ifndef SPF_NST
#define SPF_NST 2
#endif
#ifndef SPF_WV
#define SPF_WV 8
#endif
// EMIT (score-free filter): nothing is stored per (query, row); a lane appends (score key, row) to the query's candidate list only when
// the filter score reaches thr[query] (= a guaranteed lower bound of the k-th largest filter score minus the error band), i.e. for
// ~1e-3 of the scores.  bmode selects the 16*RT*WV-row blocks a launch covers: 0 = all, 1 = the sample (every ss-th block, results
// stored compactly at block index blockIdx.x), 2 = all blocks that are not in the sample.  cap = capacity of a candidate list.
template <int QT, int NP, int RT, int NST, int WV, bool EMIT = false>
__global__ void __launch_bounds__(64 * WV, ((NST * (16 * RT * WV * 128 + ((NP * QT + WV - 1) / WV) * WV * 1024) <= 81920) ? 2 : 1))
k_flat_ip_scores_split(const float* __restrict__ X, int64_t N, int64_t ldx, int D, const __bf16* __restrict__ qs, int nq,
                       float* __restrict__ scores, int64_t ld, float* __restrict__ blkmax, int nblk_ld, const int* __restrict__ gate,
                       int bmode, int ss, int unit, const float* __restrict__ thr, unsigned long long* __restrict__ cand,
                       unsigned int* __restrict__ cnt, int64_t nbx, unsigned int cap) {
  static_assert(!EMIT || NP == 1, "the emitting epilogue belongs to the single-product filter");
  constexpr int RB = 16 * RT * WV;               // corpus rows per workgroup (WV waves x RT 16-row tiles)
  constexpr int QINST = NP * QT;                 // 1-KiB LDS-DMA instructions per q slice
  constexpr int QI4 = (QINST + WV - 1) / WV;     // ... per wave (the last ones re-load the final plane into padding: equal counts per wave)
  constexpr int QBYTES = QI4 * WV * 1024;
  constexpr int XT = RB * S_BK * 4;              // X k-slice: RB rows x 128 B
  constexpr int STAGE = XT + QBYTES;
  constexpr int CW = 2 * RT + QI4;               // DMA instructions per wave per stage
  static_assert((NST - 1) * CW <= 63, "vmcnt immediate");
  constexpr int LDS_BYTES = NST * STAGE;
  __shared__ __attribute__((aligned(1024))) char smem[LDS_BYTES];
  if (gate != nullptr && *gate == 0) return;     // fallback launch of the bounded search: nothing overflowed
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // nbx = blocks of the launch; a grid smaller than that walks them (the gated fallback launch of the bounded search is capped: its
  // workgroups return at once when nothing overflowed, and 78 k of them over a 10M-row shard still cost 19 us)
  for (int64_t bx = blockIdx.x; bx < nbx; bx += gridDim.x) {
  if (bx != (int64_t)blockIdx.x) __syncthreads();   // the previous block's LDS is dead
  int64_t blk = bx;                                // sample units of `unit` consecutive blocks, every ss-th unit is in the sample
  if (bmode == 1) { const int u = (int)(bx / unit); blk = (int64_t)u * ss * unit + (bx - (int64_t)u * unit); }
  else if (bmode == 2) {
    const int u = (int)(bx / unit), g = u / (ss - 1);
    blk = ((int64_t)g * ss + 1 + (u - g * (ss - 1))) * unit + (bx - (int64_t)u * unit);
  }
  const int64_t n0 = blk * RB;                     // corpus rows of this workgroup
  const int64_t n0s = bx * RB;    // where its scores go (compact in sample mode)

  const float* px[2 * RT];
#pragma unroll
  for (int i = 0; i < 2 * RT; ++i) {
    int s = (wave * 2 * RT + i) * 64 + lane;
    int row = s >> 3, c = (s & 7) ^ ((row >> 1) & 7);
    int64_t g = min(n0 + row, N - 1);
    px[i] = X + g * ldx + c * 4;
  }
  const __bf16* pq = qs + (int64_t)lane * 8;     // + (kt*QINST + j) * 512 elements
  auto stage = [&](int st, int kt) {
    char* sX = smem + st * STAGE;
#pragma unroll
    for (int i = 0; i < 2 * RT; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(px[i] + (int64_t)kt * S_BK), (lptr_t)(sX + (wave * 2 * RT + i) * 1024), 16, 0, 2 /* nt: streamed once */);
    char* sQ = sX + XT;
#pragma unroll
    for (int jj = 0; jj < QI4; ++jj) {
      const int j = wave + WV * jj;
      const int jsrc = j < QINST ? j : QINST - 1;
      __builtin_amdgcn_global_load_lds((gptr_t)(pq + ((int64_t)kt * QINST + jsrc) * 512), (lptr_t)(sQ + j * 1024), 16, 0, 0);
    }
  };

  const int fi = lane & 15, fq = lane >> 4;
  const int xs = fi >> 1;
  const int xoff0 = fi * 128 + ((fq ^ xs) << 4);          // chunk fq      : k = 4fq .. 4fq+3
  const int xoff1 = fi * 128 + (((4 + fq) ^ xs) << 4);    // chunk 4 + fq  : k = 16+4fq ..

  f32x4 acc[RT][QT];
#pragma unroll
  for (int a = 0; a < RT; ++a)
#pragma unroll
    for (int b = 0; b < QT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = D / S_BK;
#pragma unroll
  for (int st = 0; st < NST - 1; ++st)
    if (st < nk) stage(st, st);
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt % NST;
    // stage kt has landed when at most the (NST-2) younger stages' instructions of this wave are outstanding
    if (NST > 2 && kt + NST - 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * CW) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                       // everyone's part of stage kt is in LDS; slot (kt-1) % NST is free
    if (kt + NST - 1 < nk) stage((kt + NST - 1) % NST, kt + NST - 1);
    const char* sX = smem + cur * STAGE + (wave * 16 * RT) * 128;
    const char* sQ = smem + cur * STAGE + XT + lane * 16;
    bf16x8 xh[RT], xm[NP == 3 ? RT : 1], xl[NP == 3 ? RT : 1];
#pragma unroll
    for (int a = 0; a < RT; ++a) {
      f32x4 v0 = *(const f32x4*)(sX + a * 2048 + xoff0);
      f32x4 v1 = *(const f32x4*)(sX + a * 2048 + xoff1);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (NP == 3) {
          __bf16 h, m, l;
          split3(v0[j], h, m, l);
          xh[a][j] = h; xm[a][j] = m; xl[a][j] = l;
          split3(v1[j], h, m, l);
          xh[a][4 + j] = h; xm[a][4 + j] = m; xl[a][4 + j] = l;
        } else {                                           // the filter operand: fp16(x), like a shadow row would hold
          xh[a][j] = f2h_bits(v0[j]);
          xh[a][4 + j] = f2h_bits(v1[j]);
        }
      }
    }
#pragma unroll
    for (int b = 0; b < QT; ++b) {
      bf16x8 qh = *(const bf16x8*)(sQ + b * 1024);
      if (NP == 3) {
        bf16x8 qm = *(const bf16x8*)(sQ + (QT + b) * 1024);
        bf16x8 ql = *(const bf16x8*)(sQ + (2 * QT + b) * 1024);
#pragma unroll
        for (int a = 0; a < RT; ++a) {
          f32x4 c = acc[a][b];
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xl[a], qh, c, 0, 0, 0);   // small terms first
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh[a], ql, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xm[a], qm, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xm[a], qh, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh[a], qm, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh[a], qh, c, 0, 0, 0);
          acc[a][b] = c;
        }
      } else {
#pragma unroll
        for (int a = 0; a < RT; ++a) acc[a][b] = mfma_f16(xh[a], qh, acc[a][b]);
      }
    }
  }
  if constexpr (EMIT) {
    // D[i = corpus row][j = query]: lane holds query j = fi, rows fq*4 + {0..3}.  All list reservations of a wave are issued before
    // the first one is waited for (the atomics go to the memory side: ~2 us each, but independent).
    float t[QT];
    unsigned int c[QT][RT], p[QT][RT];
#pragma unroll
    for (int b = 0; b < QT; ++b) {
      const int qi = b * 16 + fi;
      t[b] = qi < nq ? thr[qi] : FLT_MAX;
    }
#pragma unroll
    for (int b = 0; b < QT; ++b)
#pragma unroll
      for (int a = 0; a < RT; ++a) {
        const int64_t n = n0 + wave * 16 * RT + a * 16 + fq * 4;
        c[b][a] = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) c[b][a] += (n + e < N && acc[a][b][e] >= t[b]) ? 1u : 0u;
      }
#pragma unroll
    for (int b = 0; b < QT; ++b)
#pragma unroll
      for (int a = 0; a < RT; ++a) {
        p[b][a] = 0;
        if (c[b][a]) p[b][a] = atomicAdd(&cnt[(b * 16 + fi) * CNT_STRIDE], c[b][a]);
      }
#pragma unroll
    for (int b = 0; b < QT; ++b)
#pragma unroll
      for (int a = 0; a < RT; ++a)
        if (c[b][a]) {
          const int64_t n = n0 + wave * 16 * RT + a * 16 + fq * 4;
          unsigned int pp = p[b][a];
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (n + e < N && acc[a][b][e] >= t[b]) {
              if (pp < cap) cand[(int64_t)(b * 16 + fi) * cap + pp] = sel_pack(f2key(acc[a][b][e]), n + e);
              ++pp;
            }
        }
    return;          // (EMIT implies NP == 1: one block per workgroup)
  }
  __syncthreads();   // the k loop's last LDS reads are done before the epilogue reuses the buffer

  // D[i = corpus row][j = query]: lane holds query j = fi, rows fq*4 + {0..3}.  First the per-(query, 128-row block) maximum,
  // which gives k_topk_select a safe threshold without an extra pass over the scores.
  constexpr int WPG = 8 / RT;                    // waves per 128-row group
  float* wmax = (float*)smem;  // [WV waves][QT*16]
#pragma unroll
  for (int b = 0; b < QT; ++b) {
    int qi = b * 16 + fi;
    float mx = -FLT_MAX;
#pragma unroll
    for (int a = 0; a < RT; ++a) {
      int64_t n = n0 + wave * 16 * RT + a * 16 + fq * 4;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (n + e >= N) acc[a][b][e] = -FLT_MAX;
        mx = fmaxf(mx, acc[a][b][e]);
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    if (fq == 0) wmax[wave * (QT * 16) + qi] = mx;
  }
  __syncthreads();
  if (blkmax != nullptr) {
    for (int t = tid; t < (RB / 128) * QT * 16; t += 64 * WV) {
      const int grp = t / (QT * 16), qi = t % (QT * 16);
      if (qi < nq) {
        float mx = -FLT_MAX;
#pragma unroll
        for (int w = 0; w < WPG; ++w) mx = fmaxf(mx, wmax[(grp * WPG + w) * (QT * 16) + qi]);
        blkmax[(int64_t)qi * nblk_ld + bx * (RB / 128) + grp] = mx;
      }
    }
  }
  // Scores: the accumulator layout would store 64-B pieces into 16 different query rows per instruction (measured: 0.18 ms of
  // the 1.8 ms filter pass at Q = 100).  Staged through LDS instead -- [query][RB rows] with a 16-B pad per query, conflict-free
  // ds_write_b128 -- and written as whole RB*4-byte row segments, 1 KiB contiguous per wave instruction.
  constexpr int SEG = RB * 4 + 16;
  constexpr int QPT = (LDS_BYTES / SEG / 16) < QT ? (LDS_BYTES / SEG / 16) : QT;   // q-tiles staged per pass
  static_assert(QPT >= 1, "epilogue staging does not fit");
  constexpr int NPASS = (QT + QPT - 1) / QPT;
#pragma unroll
  for (int ps = 0; ps < NPASS; ++ps) {
    __syncthreads();
#pragma unroll
    for (int b = 0; b < QT; ++b) {
      if (b / QPT == ps) {
#pragma unroll
        for (int a = 0; a < RT; ++a)
          *(f32x4*)(smem + ((b - ps * QPT) * 16 + fi) * SEG + (wave * 16 * RT + a * 16 + fq * 4) * 4) = acc[a][b];
      }
    }
    __syncthreads();
    const int nqt = (QT - ps * QPT) < QPT ? (QT - ps * QPT) : QPT;
    for (int idx = tid; idx < nqt * 16 * (RB / 4); idx += 64 * WV) {
      const int ql = idx / (RB / 4), c = idx % (RB / 4);
      const int qi = ps * QPT * 16 + ql;
      if (qi < nq)
        *(f32x4*)(scores + (int64_t)qi * ld + n0s + c * 4) = *(const f32x4*)(smem + ql * SEG + c * 16);
    }
  }
  if constexpr (NP != 3) break;   // only the six-product (fallback) instantiations are ever launched with fewer workgroups than blocks;
                                  // as a real loop the single-product kernels went from 60 to 107 VGPRs (two workgroups per CU instead of three)
  }   // blocks of this workgroup
}


// ---------------------------------------------------------------------------------------------------------------
// Shadow filter with the corpus fragments streamed through REGISTERS (tiled shadow only).  A corpus element is used by exactly one
// wave, so staging X in LDS buys nothing but a barrier-coupled two-stage ring (16 KiB in flight per workgroup, 48 KiB per CU: the
// pass ran at bytes-in-flight x latency = 5.3 TB/s).  Here the 16-KiB tile of a (128-row block, k-slice) is stored FRAGMENT-MAJOR --
// [wave 0..7][k-step 0..1][lane][8 bf16], lane (fi, fq) = row 16 wave + fi, k = 32 ks + 8 fq .. + 7, the MFMA 16x16x32 A operand --
// so a wave's fragment is one 1-KiB coalesced global_load_dwordx4 and its prefetch ring is PF k-slices deep in VGPRs (8 per slice).
// Only the q k-slice, which all eight waves share, goes through LDS: a ninth PRODUCER wave requests it one slice ahead by LDS-DMA
// (its own in-order vmcnt, so the consumers' counted waits see nothing but their X loads); one barrier per k-slice.
// ---------------------------------------------------------------------------------------------------------------
template <int QT, int PF, bool EMIT, int RT = 1>
__global__ void __launch_bounds__(576, QT > 8 ? 1 : 2)
k_filter_xreg(const __bf16* __restrict__ Xb, int64_t N, int D, const __bf16* __restrict__ qs, int nq, float* __restrict__ scores, int64_t ld,
              float* __restrict__ blkmax, int nblk_ld, const int* __restrict__ gate, int bmode, int ss, int unit,
              const float* __restrict__ thr, unsigned long long* __restrict__ cand, unsigned int* __restrict__ cnt, int nlaunch, int gmax,
              unsigned int cap) {
  // RT = blocks per workgroup (launch indices RT * blockIdx.x + a < nlaunch): with two, the q slice is fetched once per 256 rows -- the
  // sample pass of a 100-query search had 391 workgroups on 256 CUs, one or two per CU (56 -> 52 us; Q = 128: 61 -> 50 us)
  static_assert(RT == 1 || !EMIT, "the emitting epilogue works on one block");
  constexpr int WV = 8, RB = 128;
  constexpr int QINST = 2 * QT;                    // 1-KiB LDS-DMA instructions per q slice
  constexpr int QBYTES = QINST * 1024;
  constexpr int SEG = RB * 4 + 16;                 // epilogue staging: one query's 128 scores + pad
  constexpr int QB = QT > 8 ? 2 : 4;               // q ring: the producer runs QB-1 slices ahead
  static_assert((QB - 2) * QINST <= 63, "vmcnt immediate");
  constexpr int LDS_BYTES = QB * QBYTES > 16 * SEG ? QB * QBYTES : 16 * SEG;
  __shared__ __attribute__((aligned(1024))) char smem[LDS_BYTES];
  if (gate != nullptr && *gate == 0) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int64_t blks[RT], lis[RT];                       // corpus block and launch index (= where the scores go: compact in sample mode) per slot
#pragma unroll
  for (int a = 0; a < RT; ++a) {
    const int li = min((int)blockIdx.x * RT + a, nlaunch - 1);   // (an odd one out: the last block again, nothing stored)
    int64_t blk = li;
    if (bmode == 1) { const int u = li / unit; blk = (int64_t)u * ss * unit + (li - u * unit); }
    else if (bmode == 2) {
      const int u = li / unit, g = u / (ss - 1);
      blk = ((int64_t)g * ss + 1 + (u - g * (ss - 1))) * unit + (li - u * unit);
    }
    blks[a] = blk;
    lis[a] = li;
  }
  const int nk = D / 64;
  const int fi = lane & 15, fq = lane >> 4;

  f32x4 acc[RT][QT];
#pragma unroll
  for (int a = 0; a < RT; ++a)
#pragma unroll
    for (int b = 0; b < QT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (wave == WV) {
    // ---- producer: q slice kt+1 requested while the consumers work on kt
    const __bf16* pq = qs + (int64_t)lane * 8;
    auto stage_q = [&](int kt) {
      char* sQ = smem + (kt % QB) * QBYTES;
#pragma unroll
      for (int j = 0; j < QINST; ++j)
        __builtin_amdgcn_global_load_lds((gptr_t)(pq + ((int64_t)kt * QINST + j) * 512), (lptr_t)(sQ + j * 1024), 16, 0, 0);
    };
#pragma unroll
    for (int p = 0; p < QB - 1; ++p)
      if (p < nk) stage_q(p);
    for (int kt = 0; kt < nk; ++kt) {
      if (kt + QB - 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((QB - 2) * QINST) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                // q(kt) is in LDS; every consumer has finished slice kt-1 -> its buffer is free
      if (kt + QB - 1 < nk) stage_q(kt + QB - 1);
    }
  } else {
    // ---- consumers: wave w owns rows 16w .. 16w+15 of each block
    const bf16x8* px[RT];                          // + kt*1024 (+64: k-step 1)
#pragma unroll
    for (int a = 0; a < RT; ++a) px[a] = (const bf16x8*)(Xb + (min(blks[a], (N - 1) >> 7) * (int64_t)(D / 64)) * 8192 + wave * 1024) + lane;
    bf16x8 xf[PF][RT][2];
    auto load = [&](int slot, int kt) __attribute__((always_inline)) {
#pragma unroll
      for (int a = 0; a < RT; ++a) {
        xf[slot][a][0] = __builtin_nontemporal_load(px[a] + (int64_t)kt * 1024);
        xf[slot][a][1] = __builtin_nontemporal_load(px[a] + (int64_t)kt * 1024 + 64);
      }
    };
    auto step = [&](int u, int kt, bool fetch) __attribute__((always_inline)) {
      if (fetch) load((u + PF - 1) % PF, kt + PF - 1);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      const char* sQ = smem + (kt % QB) * QBYTES + lane * 16;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int b = 0; b < QT; ++b) {
          const bf16x8 qf = *(const bf16x8*)(sQ + (ks * QT + b) * 1024);
#pragma unroll
          for (int a = 0; a < RT; ++a) acc[a][b] = mfma_f16(xf[u][a][ks], qf, acc[a][b]);
        }
    };
    // steady state: every step of the trip prefetches (no guard -> the compiler's counted vmcnt keeps PF-1 slices in flight);
    // the last trips re-check per step
    int kt0 = 0;
    if (2 * PF - 2 < nk) {
#pragma unroll
      for (int p = 0; p < PF - 1; ++p) {
        load(p, p);
        __builtin_amdgcn_sched_barrier(0);           // issue order = ring order: the counted waits of the loop rely on it
      }
      for (; kt0 + 2 * PF - 2 < nk; kt0 += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) step(u, kt0 + u, true);
      }
    } else {
#pragma unroll
      for (int p = 0; p < PF - 1; ++p)
        if (p < nk) load(p, p);
    }
    for (; kt0 < nk; kt0 += PF) {
#pragma unroll
      for (int u = 0; u < PF; ++u)
        if (kt0 + u < nk) step(u, kt0 + u, kt0 + u + PF - 1 < nk);
    }
  }
  if constexpr (EMIT) {
    if (wave == WV) return;
    float t[QT];
    unsigned int c[QT], p[QT];
#pragma unroll
    for (int b = 0; b < QT; ++b) {
      const int qi = b * 16 + fi;
      t[b] = qi < nq ? thr[qi] : FLT_MAX;
    }
    const int64_t n = blks[0] * RB + wave * 16 + fq * 4;
#pragma unroll
    for (int b = 0; b < QT; ++b) {
      c[b] = 0;
#pragma unroll
      for (int e = 0; e < 4; ++e) c[b] += (n + e < N && acc[0][b][e] >= t[b]) ? 1u : 0u;
    }
#pragma unroll
    for (int b = 0; b < QT; ++b) {
      p[b] = 0;
      if (c[b]) p[b] = atomicAdd(&cnt[(b * 16 + fi) * CNT_STRIDE], c[b]);
    }
#pragma unroll
    for (int b = 0; b < QT; ++b)
      if (c[b]) {
        unsigned int pp = p[b];
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (n + e < N && acc[0][b][e] >= t[b]) {
            if (pp < cap) cand[(int64_t)(b * 16 + fi) * cap + pp] = sel_pack(f2key(acc[0][b][e]), n + e);
            ++pp;
          }
      }
    return;
  }
  float* wmax = (float*)smem;  // [8 waves][QT*16]
  constexpr int QPT = (LDS_BYTES / SEG / 16) < QT ? (LDS_BYTES / SEG / 16) : QT;   // q-tiles staged per pass
  static_assert(QPT >= 1, "epilogue staging does not fit");
  constexpr int NPASS = (QT + QPT - 1) / QPT;
#pragma unroll
  for (int a = 0; a < RT; ++a) {
    if (a > 0 && (int)blockIdx.x * RT + a >= nlaunch) break;     // (uniform)
    const int64_t n0 = blks[a] * RB, n0s = lis[a] * RB;
    __syncthreads();   // all nine waves: the q buffers (or the previous block's staging) are dead, the epilogue reuses them
    if (wave < WV) {
#pragma unroll
      for (int b = 0; b < QT; ++b) {
        const int qi = b * 16 + fi;
        float mx = -FLT_MAX;
        const int64_t n = n0 + wave * 16 + fq * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (n + e >= N) acc[a][b][e] = -FLT_MAX;
          mx = fmaxf(mx, acc[a][b][e]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        if (fq == 0) wmax[wave * (QT * 16) + qi] = mx;
      }
    }
    __syncthreads();
    if (blkmax != nullptr && gmax) {             // maxima of the eight 16-row groups: [query][8 * block + wave], row stride 8 * nblk_ld
      for (int t = tid; t < QT * 16 * WV; t += 576) {
        const int qi = t >> 3, w = t & 7;
        if (qi < nq) blkmax[(int64_t)qi * (8 * (int64_t)nblk_ld) + lis[a] * 8 + w] = wmax[w * (QT * 16) + qi];
      }
    } else if (blkmax != nullptr) {
      for (int t = tid; t < QT * 16; t += 576)
        if (t < nq) {
          float mx = -FLT_MAX;
#pragma unroll
          for (int w = 0; w < WV; ++w) mx = fmaxf(mx, wmax[w * (QT * 16) + t]);
          blkmax[(int64_t)t * nblk_ld + lis[a]] = mx;
        }
    }
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      __syncthreads();
      if (wave < WV) {
#pragma unroll
        for (int b = 0; b < QT; ++b)
          if (b / QPT == ps) *(f32x4*)(smem + ((b - ps * QPT) * 16 + fi) * SEG + (wave * 16 + fq * 4) * 4) = acc[a][b];
      }
      __syncthreads();
      const int nqt = (QT - ps * QPT) < QPT ? (QT - ps * QPT) : QPT;
      for (int idx = tid; idx < nqt * 16 * (RB / 4); idx += 576) {
        const int ql = idx / (RB / 4), c = idx % (RB / 4);
        const int qi = ps * QPT * 16 + ql;
        if (qi < nq) *(f32x4*)(scores + (int64_t)qi * ld + n0s + c * 4) = *(const f32x4*)(smem + ql * SEG + c * 16);
      }
    }
  }
}

// The main pass of the score-free filter as PERSISTENT workgroups: with one 128-row block per workgroup every block paid the latency of
// its first loads and of its list reservations (memory-side atomics, ~2 us) with nothing of its own in flight -- 11 % of the pass.
// Here a workgroup walks blocks blockIdx.x, + gridDim.x, ...: the X ring and the q producer run straight across block boundaries,
// the thresholds stay in registers, and hits (~1e-3 of the scores) are parked in a per-wave LDS list that is written to the queries'
// candidate lists once, at the end (or when it fills up).  RT = blocks worked on at a time (the q slice is read from LDS once for both).
// Requires D / 64 to be a multiple of PF (the ring phase is the same at every block start); other shapes use k_filter_xreg<.., EMIT>.
// (Round 4, measured and not kept: for narrow rows (D = 256: a block is 64 KiB, an epilogue every four k-steps) TWO persistent workgroups per
// CU with half the LDS each, one block at a time, so that one's epilogue runs under the other's loads -- 96 VGPRs, 78 KiB LDS, correct, and
// 1.25M x 256 / Q = 100 went from 0.201-0.206 to 0.276-0.284 ms, 10M x 256 from 0.99 to 1.26: with one block in work the q fragments are read
// from LDS once per block instead of once per two, and that, not the epilogue, is what the narrow-row pass is short of.  Q = 1, 32: no change.)
// LDS bytes of the persistent emitting pass (q ring + per-wave hit lists + thresholds)
template <int QT>
struct EmitLds {
  static constexpr int WV = 8;
  static constexpr int QBYTES = 2 * QT * 1024;
  static constexpr int QB = QT > 8 ? 3 : 4;
  static constexpr int WQC_BYTES = QT * 16 * 4;
  static constexpr int WCAP_FIT = ((160 * 1024 - QB * QBYTES - QT * 64 - 1024 - WV * (WQC_BYTES + 16)) / (WV * 12)) / 64 * 64;
  static constexpr int WCAP = WCAP_FIT > 1024 ? 1024 : WCAP_FIT;
  static constexpr int WL_BYTES = WCAP * 12 + 16 + WQC_BYTES;
  static constexpr int BYTES = QB * QBYTES + WV * WL_BYTES + QT * 16 * 4;
};

// thr_ready / thr_target (fused kernel): the thresholds are published by other workgroups of the SAME launch -- the first block step's K loop
// runs before they are needed; each consumer wave then polls *thr_ready until it reaches thr_target (the selection items were all claimed by
// running workgroups before this workgroup got here, so the wait ends) and loads the thresholds past the L1.  NULL: thr is final at launch.
template <int QT, int PF, int RT>
__device__ __forceinline__ void filter_emit_body(char* smem, const __bf16* __restrict__ Xb, int64_t N, int D, const __bf16* __restrict__ qs, int nq, int nblocks, int bmode, int ss,
                   int unit, const float* __restrict__ thr, unsigned long long* __restrict__ cand, unsigned int* __restrict__ cnt, unsigned int cap,
                   const unsigned int* thr_ready, unsigned int thr_target, unsigned long long* ts = nullptr) {
  constexpr int WV = 8;
  constexpr int QINST = 2 * QT;
  constexpr int QBYTES = QINST * 1024;
  constexpr int QB = QT > 8 ? 3 : 4;               // q ring: the producer runs QB-1 slices ahead
  static_assert((QB - 2) * QINST <= 63, "vmcnt immediate");
  // hits a wave parks in LDS: as many as the 160 KiB of the CU allow next to the q ring (one workgroup per CU).  A pass emits ~5 000 hits per
  // query, i.e. 2.4 x queries per wave: with 320 entries most waves of a 100-query pass had to flush once in mid-pass (one memory-side atomic
  // per hit, the wave waits, the workgroup waits for it at the next barrier): 630 us against 580 us for the same pass with hardly any hits
  constexpr int WQC_BYTES = QT * 16 * 4;          // per wave: hits per query of a mid-pass flush, then the first global slot (see flush)
  constexpr int WCAP_FIT = ((160 * 1024 - QB * QBYTES - QT * 64 - 1024 - WV * (WQC_BYTES + 16)) / (WV * 12)) / 64 * 64;
  constexpr int WCAP = WCAP_FIT > 1024 ? 1024 : WCAP_FIT;
  static_assert(WCAP >= 256, "hit lists do not fit next to the q ring");
  constexpr int WL_BYTES = WCAP * 12 + 16 + WQC_BYTES;
  static_assert(EmitLds<QT>::BYTES == QB * QBYTES + WV * WL_BYTES + QT * 16 * 4, "EmitLds out of sync");
  float* sthr = (float*)(smem + QB * QBYTES + WV * WL_BYTES);   // the thresholds (LDS: they are needed once per block, not per k-step)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (thr_ready == nullptr) {
    for (int i = tid; i < QT * 16; i += 576) sthr[i] = i < nq ? thr[i] : FLT_MAX;
  }
  __syncthreads();
  // this workgroup's blocks: launch indices blockIdx.x + i * gridDim.x, i < nbw (the counts differ by at most one block over the grid),
  // walked RT at a time; an odd one out at the end is worked on with its own block in the second slot (cache hits, result dropped)
  const int nbw = (nblocks - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const int nmine = (nbw + RT - 1) / RT;                                         // steps of RT blocks (>= 1)
  const int nk = D / 64;
  const int64_t total = (int64_t)nmine * nk;                                     // k-steps of this workgroup
  const bool qres = nk <= QB;                      // the whole q fits the ring (D <= 256): staged once, no barrier per k-step

  if (wave == WV) {
    // ---- producer: the q slices, cyclically, QB-1 steps ahead
    const __bf16* pq = qs + (int64_t)lane * 8;
    int hs = 0, hb = 0;                            // head: slice and ring buffer
    auto stage_next = [&]() {
      char* sQ = smem + hb * QBYTES;
#pragma unroll
      for (int j = 0; j < QINST; ++j)
        __builtin_amdgcn_global_load_lds((gptr_t)(pq + ((int64_t)hs * QINST + j) * 512), (lptr_t)(sQ + j * 1024), 16, 0, 0);
      hs = hs + 1 == nk ? 0 : hs + 1;
      hb = hb + 1 == QB ? 0 : hb + 1;
    };
    if (qres) {
      for (int p = 0; p < nk; ++p) stage_next();
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    } else {
#pragma unroll
      for (int p = 0; p < QB - 1; ++p)
        if (p < total) stage_next();
      for (int64_t g = 0; g < total; ++g) {
        if (g + QB - 2 < total) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((QB - 2) * QINST) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();              // q(g) is in LDS; the consumers are done with step g-1 -> its buffer is free
        if (g + QB - 1 < total) stage_next();
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) __builtin_amdgcn_s_barrier();   // the four barriers of the consumers' final flush
    return;
  }

  // ---- consumers: wave w owns rows 16w .. 16w+15 of each of the RT blocks in work
  const int fi = lane & 15, fq = lane >> 4;
  unsigned long long* wl = (unsigned long long*)(smem + QB * QBYTES + wave * WL_BYTES);
  unsigned int* wq = (unsigned int*)(wl + WCAP);
  auto blk_of = [&](int i) -> int {                // launch index -> 128-row block (see k_flat_ip_scores_split); all < 2^25
    if (bmode != 2) return i;
    const unsigned int u = (unsigned int)i / (unsigned int)unit, g = u / (unsigned int)(ss - 1);
    return (int)((g * ss + 1 + (u - g * (ss - 1))) * unit + ((unsigned int)i - u * unit));
  };
  const int last_blk = (int)((N - 1) >> 7);
  auto li_of = [&](int j, int a) -> int { return (int)blockIdx.x + min(j * RT + a, nbw - 1) * (int)gridDim.x; };   // launch index of slot a in step j
  auto base_of = [&](int i) -> const bf16x8* {
    const int b = min(blk_of(i), last_blk);
    return (const bf16x8*)(Xb + ((int64_t)b * (D / 64)) * 8192 + wave * 1024) + lane;
  };
  unsigned int wcnt = 0;                           // entries in this wave's list (wave-uniform)
  unsigned int* wqc = (unsigned int*)((char*)wl + WCAP * 12 + 16);
  // mid-pass flush, by the wave alone, ONE global reservation per (wave, query with hits) -- round 2 reserved per hit (memory-side atomics
  // on ~100 addresses: fine while a pass emitted 5 k hits per query; at top_k = 1000 it emits ~26 k per query, every wave flushes in
  // mid-pass, and the 2.6 M single-hit reservations of a 100-query pass took 3.3 ms).  LDS operations of one wave execute in order, so
  // the phases below need no barrier: count per query (the LDS atomic's return value is the hit's rank inside the wave's batch),
  // reserve, scatter.
  auto flush = [&]() {
    const unsigned int tot = min(wcnt, (unsigned int)WCAP);
    for (int t = lane; t < QT * 16; t += 64) wqc[t] = 0u;
    for (unsigned int i = lane; i < tot; i += 64) {
      const unsigned int col = wq[i];
      wq[i] = col | (atomicAdd(&wqc[col], 1u) << 8);                 // (col < 256, rank < 1024)
    }
    for (int t = lane; t < QT * 16; t += 64) {
      const unsigned int c = wqc[t];
      if (c) wqc[t] = atomicAdd(&cnt[t * CNT_STRIDE], c);
    }
    for (unsigned int i = lane; i < tot; i += 64) {
      const unsigned int e = wq[i], col = e & 255u, slot = wqc[col] + (e >> 8);
      if (slot < cap) cand[(int64_t)col * cap + slot] = wl[i];
    }
    wcnt = 0;
  };

  f32x4 acc[RT][QT];
#pragma unroll
  for (int a = 0; a < RT; ++a)
#pragma unroll
    for (int b = 0; b < QT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  // prefetch head
  int pf_j = 0, pf_kt = 0;
  const bf16x8* pfp[RT];
#pragma unroll
  for (int a = 0; a < RT; ++a) pfp[a] = base_of(li_of(0, a));
  bf16x8 xf[PF][RT][2];
  auto fetch = [&](int slot) {
#pragma unroll
    for (int a = 0; a < RT; ++a) {
      xf[slot][a][0] = __builtin_nontemporal_load(pfp[a]);
      xf[slot][a][1] = __builtin_nontemporal_load(pfp[a] + 64);
    }
    if (++pf_kt == nk) {                          // next group of blocks (past the end: the last group again -- loaded, never used)
      pf_kt = 0;
      pf_j = min(pf_j + 1, nmine - 1);
#pragma unroll
      for (int a = 0; a < RT; ++a) pfp[a] = base_of(li_of(pf_j, a));
    } else {
#pragma unroll
      for (int a = 0; a < RT; ++a) pfp[a] += 1024;
    }
  };
#pragma unroll
  for (int p = 0; p < PF - 1; ++p) {
    fetch(p);
    __builtin_amdgcn_sched_barrier(0);             // issue order = ring order: the counted waits of the loop rely on it
  }
  int qb = 0;
  const int qper = qres ? nk : QB;                 // slice of step g sits in ring buffer g % qper
  unsigned int ovf = 0;
  if (qres) {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();                  // the whole q has landed
    __builtin_amdgcn_sched_barrier(0);
  }
  for (int j = 0; j < nmine; ++j) {
    for (int kt0 = 0; kt0 < nk; kt0 += PF) {
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        fetch((u + PF - 1) % PF);
        __builtin_amdgcn_sched_barrier(0);
        if (!qres) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const char* sQ = smem + qb * QBYTES + lane * 16;
        qb = qb + 1 == qper ? 0 : qb + 1;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
          for (int b = 0; b < QT; ++b) {
            const bf16x8 qf = *(const bf16x8*)(sQ + (ks * QT + b) * 1024);
#pragma unroll
            for (int a = 0; a < RT; ++a) acc[a][b] = mfma_f16(xf[u][a][ks], qf, acc[a][b]);
          }
        }
      }
    }
    if (thr_ready != nullptr && j == 0) {
      // thresholds published by the selection step of this launch: every consumer wave waits for itself and fills the (shared) table with the
      // same values -- a wave reads the table only after its own complete write, so no barrier is needed
      if (ts != nullptr && tid == 0) ts[4] = __builtin_amdgcn_s_memrealtime();
      if (lane == 0)
        while (__hip_atomic_load(thr_ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < thr_target) __builtin_amdgcn_s_sleep(8);
      if (ts != nullptr && tid == 0) ts[5] = __builtin_amdgcn_s_memrealtime();
      // (no acquire fence: the thresholds are read with device-scope loads, which do not go through this XCD's caches; an agent-scope acquire
      // here would invalidate the L2 once per wave -- 2048 times per launch, under the streaming pass)
      for (int i = lane; i < QT * 16; i += 64) sthr[i] = i < nq ? __hip_atomic_load(thr + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : FLT_MAX;
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    // ---- block epilogue: D[i = corpus row][j = query]: lane holds query fi of tile b, rows fq*4 + {0..3}.  LDS only: a wave tile with
    //      more hits than the list holds (near-duplicate rows) pushes the query's counter past the list capacity instead -> flagged, redone by the fallback
#pragma unroll
    for (int a = 0; a < RT; ++a) {
      const int64_t n64 = (int64_t)blk_of(li_of(j, a)) * 128 + wave * 16 + fq * 4;
      const unsigned int n = (unsigned int)n64;
      const int valid = j * RT + a < nbw ? (int)max((int64_t)0, min((int64_t)4, N - n64)) : 0;
#pragma unroll
      for (int b = 0; b < QT; ++b) {
        const f32x4 v = acc[a][b];
        acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float tb = sthr[b * 16 + fi];
        const unsigned int col = b * 16 + fi;
        // the list is private to the wave: slots by ballot + lane prefix, the fill count in a scalar (an LDS atomic per tile with a hit --
        // nearly every tile at ~17 hits per step -- was a serial ~120-cycle round trip each)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const bool h = e < valid && v[e] >= tb;
          const unsigned long long m = __ballot(h);
          if (m) {
            const unsigned int pos = wcnt + __builtin_amdgcn_mbcnt_hi((unsigned int)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m, 0u));
            if (h) {
              if (pos < WCAP) { wl[pos] = sel_pack(f2key(v[e]), n + e); wq[pos] = col; }
              else ovf |= 1u << b;
            }
            wcnt += (unsigned int)__popcll(m);
          }
        }
      }
    }
    if (__builtin_expect(ovf != 0, 0)) {
      for (int b = 0; b < QT; ++b)
        if ((ovf >> b) & 1u) {
          unsigned int col = b * 16 + fi;
          asm volatile("" : "+v"(col));             // (keeps the address arithmetic inside this cold branch)
          __hip_atomic_fetch_add(&cnt[col * CNT_STRIDE], cap + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      ovf = 0;
    }
    // (wave-uniform.)  Flushed at half full: a step of RT blocks adds ~8 hits per wave at k = 100 but ~80 at the reference's top_k = 1000 (~150 at
    // 2048), and a hit that finds the list full flags its query for the exact fallback -- with the 96-entry margin of round 2 every
    // 100-query pass at k = 1000 sent queries there (2.9 ms of six-product pass + select for a 0.7 ms filter pass)
    // (Tried: all waves flushing at the same, host-scheduled steps so that the stalls coincide: 670 -> 681 us at k = 1000, not kept.)
    if (wcnt > WCAP / 2) flush();
  }
  // ---- final flush, by the workgroup: one list reservation per (workgroup, query) instead of one per hit.  The ~5e5 hits of a pass would
  //      otherwise reach the ~100 list counters at the same time, at the end of the pass, and the memory-side atomics of one address
  //      serialise (k = 100: 50 us of tail; with k = 1, i.e. hardly any hits, the same pass took 580 instead of 630 us).
  unsigned int* qcnt = (unsigned int*)smem;        // [QT*16] hits per query, then the running offset (the q ring is dead)
  unsigned int* qbase = qcnt + QT * 16;            // [QT*16] first slot of this workgroup in the query's list
  __builtin_amdgcn_s_barrier();                    // every wave is out of the k loop: nobody reads the q ring any more
  for (int i = tid; i < QT * 16; i += 512) qcnt[i] = 0;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  const unsigned int tot = min(wcnt, (unsigned int)WCAP);
  for (unsigned int i = lane; i < tot; i += 64) atomicAdd(&qcnt[wq[i]], 1u);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  for (int i = tid; i < QT * 16; i += 512) {
    const unsigned int c = qcnt[i];
    if (c) qbase[i] = atomicAdd(&cnt[i * CNT_STRIDE], c);
    qcnt[i] = 0;
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  for (unsigned int i = lane; i < tot; i += 64) {
    const unsigned int col = wq[i];
    const unsigned int slot = qbase[col] + atomicAdd(&qcnt[col], 1u);
    if (slot < cap) cand[(int64_t)col * cap + slot] = wl[i];
  }
}

template <int QT, int PF, int RT>
__global__ void __launch_bounds__(576, (QT > 8 || RT > 1) ? 3 : 5)   // (second argument: waves per SIMD -> two workgroups of nine waves per CU need five)
k_filter_xreg_emit(const __bf16* __restrict__ Xb, int64_t N, int D, const __bf16* __restrict__ qs, int nq, int nblocks, int bmode, int ss,
                   int unit, const float* __restrict__ thr, unsigned long long* __restrict__ cand, unsigned int* __restrict__ cnt, unsigned int cap) {
  __shared__ __attribute__((aligned(1024))) char smem[EmitLds<QT>::BYTES];
  filter_emit_body<QT, PF, RT>(smem, Xb, N, D, qs, nq, nblocks, bmode, ss, unit, thr, cand, cnt, cap, nullptr, 0u);
}

// Device-scope ("sc1") loads / stores: data one workgroup writes and another workgroup of the SAME launch reads (the fused filter kernel) must
// not live in an XCD's L2 -- the eight L2s of the chip are not coherent with each other inside a kernel.  An agent-scope fence would do it too
// (buffer_wbl2 / buffer_inv of the whole L2, per wave that executes it): measured 0.31 vs 0.18 ms on a 125 k-row shard.  COH = false: plain accesses.
template <bool COH>
__device__ __forceinline__ float ld1(const float* p) {
  if constexpr (COH) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else return *p;
}
template <bool COH>
__device__ __forceinline__ f32x4 ld4(const float* p) {          // 16-byte aligned
  if constexpr (COH) {
    const unsigned long long a = __hip_atomic_load((const unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long b = __hip_atomic_load((const unsigned long long*)p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return f32x4{__uint_as_float((unsigned int)a), __uint_as_float((unsigned int)(a >> 32)), __uint_as_float((unsigned int)b), __uint_as_float((unsigned int)(b >> 32))};
  } else return *(const f32x4*)p;
}
__device__ __forceinline__ void st1_coh(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st4_coh(float* p, f32x4 v) {     // 16-byte aligned
  __hip_atomic_store((unsigned long long*)p, (unsigned long long)__float_as_uint(v[0]) | ((unsigned long long)__float_as_uint(v[1]) << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store((unsigned long long*)p + 1, (unsigned long long)__float_as_uint(v[2]) | ((unsigned long long)__float_as_uint(v[3]) << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------------------------------------------------------------------------------------------------------
// FUSED filter chain (round 5): sample pass -> threshold selection -> main pass in ONE persistent launch, without a grid barrier.
//
// The three-launch chain leaves the memory system idle twice (the ramp-down of the sample pass, the ~12-us selection, the ramp-up of the
// main pass): on a per-rank shard of 125 k x 2048 rows that is ~30 of 155 us.  Here every workgroup
//   S: claims sample blocks from a counter (block j of the sample = corpus block j * ss) and scores them like k_filter_xreg (compact score
//      rows + the maxima of the 16-row wave groups), counting each finished block in `done_s`;
//   -- waits until done_s == n_samp.  Every sample block was claimed by a workgroup that is RUNNING (the counter hands out work only to
//      workgroups that execute), so the wait ends whatever share of the grid is resident -- unlike a grid barrier, which deadlocks when
//      two such launches (two searches in flight: pipeline.SearchLanes) each hold part of the chip and wait for their own absent workgroups;
//   T: claims queries from a second counter and runs the selection of k_sample_threshold for them (threshold, eps, the sample rows that
//      open the candidate list), counting in `done_t`;
//   M: walks its share of the non-sample blocks like k_filter_xreg_emit; the first block step's K loop runs BEFORE the thresholds are
//      needed, each consumer wave then waits for done_t == n_queries (claimed work of running workgroups again) and loads them.
// Workgroups that find no selection left go straight to M and stream while the (at most n_queries) others select: HBM never idles.
// ---------------------------------------------------------------------------------------------------------------
// dev aid (LRX_FUSED_PHASES bit 7): per-workgroup phase timestamps (100 MHz s_memrealtime) of the last fused launch, read by lrx_probe_fused_timestamps
__device__ unsigned long long g_fused_ts[1024 * 8];
struct FusedCtl {              // five counters in the zero-initialised ints of the workspace (k_pack_queries_xb clears them)
  unsigned int ctr_s, done_s, ctr_t, done_t, pad;
};

// one 128-row sample block (RT = 1): the body of k_filter_xreg<QT, PF, false, 1> with group maxima; all nine waves call it together.
// li = index of the block inside the sample (where its scores go), blk = corpus block.
template <int QT, int PF>
__device__ __forceinline__ void filter_sample_block(char* smem, const __bf16* __restrict__ Xb, int64_t N, int D, const __bf16* __restrict__ qs, int nq,
                                                    float* __restrict__ scores, int64_t ld, float* __restrict__ gmax, int nblk_ld, int64_t blk, int64_t li) {
  constexpr int WV = 8, RB = 128;
  constexpr int QINST = 2 * QT;
  constexpr int QBYTES = QINST * 1024;
  constexpr int SEG = RB * 4 + 16;                 // epilogue staging: one query's 128 scores + pad
  constexpr int QB = QT > 8 ? 2 : 4;
  static_assert((QB - 2) * QINST <= 63, "vmcnt immediate");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nk = D / 64;
  const int fi = lane & 15, fq = lane >> 4;
  f32x4 acc[QT];
#pragma unroll
  for (int b = 0; b < QT; ++b) acc[b] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (wave == WV) {
    const __bf16* pq = qs + (int64_t)lane * 8;
    auto stage_q = [&](int kt) {
      char* sQ = smem + (kt % QB) * QBYTES;
#pragma unroll
      for (int j = 0; j < QINST; ++j)
        __builtin_amdgcn_global_load_lds((gptr_t)(pq + ((int64_t)kt * QINST + j) * 512), (lptr_t)(sQ + j * 1024), 16, 0, 0);
    };
#pragma unroll
    for (int p = 0; p < QB - 1; ++p)
      if (p < nk) stage_q(p);
    for (int kt = 0; kt < nk; ++kt) {
      if (kt + QB - 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((QB - 2) * QINST) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (kt + QB - 1 < nk) stage_q(kt + QB - 1);
    }
  } else {
    const bf16x8* px = (const bf16x8*)(Xb + (min(blk, (N - 1) >> 7) * (int64_t)(D / 64)) * 8192 + wave * 1024) + lane;
    bf16x8 xf[PF][2];
    auto load = [&](int slot, int kt) __attribute__((always_inline)) {
      xf[slot][0] = __builtin_nontemporal_load(px + (int64_t)kt * 1024);
      xf[slot][1] = __builtin_nontemporal_load(px + (int64_t)kt * 1024 + 64);
    };
    auto step = [&](int u, int kt, bool fetch) __attribute__((always_inline)) {
      if (fetch) load((u + PF - 1) % PF, kt + PF - 1);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      const char* sQ = smem + (kt % QB) * QBYTES + lane * 16;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int b = 0; b < QT; ++b) {
          const bf16x8 qf = *(const bf16x8*)(sQ + (ks * QT + b) * 1024);
          acc[b] = mfma_f16(xf[u][ks], qf, acc[b]);
        }
    };
    int kt0 = 0;
    if (2 * PF - 2 < nk) {
#pragma unroll
      for (int p = 0; p < PF - 1; ++p) {
        load(p, p);
        __builtin_amdgcn_sched_barrier(0);
      }
      for (; kt0 + 2 * PF - 2 < nk; kt0 += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) step(u, kt0 + u, true);
      }
    } else {
#pragma unroll
      for (int p = 0; p < PF - 1; ++p)
        if (p < nk) load(p, p);
    }
    for (; kt0 < nk; kt0 += PF) {
#pragma unroll
      for (int u = 0; u < PF; ++u)
        if (kt0 + u < nk) step(u, kt0 + u, kt0 + u + PF - 1 < nk);
    }
  }
  // ---- epilogue: group maxima from registers, scores through the (dead) q ring
  float* wmax = (float*)smem;                      // [8 waves][QT*16]
  constexpr int LDS_Q = QB * QBYTES > 16 * SEG ? QB * QBYTES : 16 * SEG;
  constexpr int QPT = (LDS_Q / SEG / 16) < QT ? (LDS_Q / SEG / 16) : QT;   // q-tiles staged per pass
  static_assert(QPT >= 1, "epilogue staging does not fit");
  constexpr int NPASS = (QT + QPT - 1) / QPT;
  const int64_t n0 = blk * RB, n0s = li * RB;
  __syncthreads();
  if (wave < WV) {
#pragma unroll
    for (int b = 0; b < QT; ++b) {
      const int qi = b * 16 + fi;
      float mx = -FLT_MAX;
      const int64_t n = n0 + wave * 16 + fq * 4;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (n + e >= N) acc[b][e] = -FLT_MAX;
        mx = fmaxf(mx, acc[b][e]);
      }
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      if (fq == 0) wmax[wave * (QT * 16) + qi] = mx;
    }
  }
  __syncthreads();
  for (int t = tid; t < QT * 16 * WV; t += 576) {
    const int qi = t >> 3, w = t & 7;
    if (qi < nq) st1_coh(gmax + (int64_t)qi * (8 * (int64_t)nblk_ld) + li * 8 + w, wmax[w * (QT * 16) + qi]);
  }
#pragma unroll
  for (int ps = 0; ps < NPASS; ++ps) {
    __syncthreads();
    if (wave < WV) {
#pragma unroll
      for (int b = 0; b < QT; ++b)
        if (b / QPT == ps) *(f32x4*)(smem + ((b - ps * QPT) * 16 + fi) * SEG + (wave * 16 + fq * 4) * 4) = acc[b];
    }
    __syncthreads();
    const int nqt = (QT - ps * QPT) < QPT ? (QT - ps * QPT) : QPT;
    // (read by other workgroups of this launch: device-scope stores, 8 bytes per lane so that one wave instruction writes four whole 128-byte lines)
    for (int idx = tid; idx < nqt * 16 * (RB / 2); idx += 576) {
      const int ql = idx / (RB / 2), c = idx % (RB / 2);
      const int qi = ps * QPT * 16 + ql;
      if (qi < nq) __hip_atomic_store((unsigned long long*)(scores + (int64_t)qi * ld + n0s) + c, *(const unsigned long long*)(smem + ql * SEG + c * 8), __ATOMIC_RELAXED,
                                      __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  __syncthreads();                                 // the staging region is free again (next block's q ring, or the selection's tables)
}

// The sample pass of a LARGE shard as persistent workgroups (same walk as k_filter_xreg_emit, score stores instead of hit lists): with one
// or two blocks per workgroup a 10M x 256 shard launched ~2000 workgroups that each fetched the whole q and paid their own load latency
// (120 us for 5 % of the rows).  Here the q ring and the X ring run across block boundaries (D <= 256: q is fetched once per workgroup),
// the scores of a block leave through a staging region of their own, and the producer wave takes part in the 2 RT barriers of every
// block step's epilogue.  QT <= 8, D / 64 a multiple of PF.
template <int QT, int PF, int RT>
__global__ void __launch_bounds__(576, 3)
k_filter_xreg_store(const __bf16* __restrict__ Xb, int64_t N, int D, const __bf16* __restrict__ qs, int nq, float* __restrict__ scores, int64_t ld,
                    float* __restrict__ gmax, int nblk_ld, int nblocks, int bmode, int ss, int unit) {
  static_assert(QT <= 8, "staging sized for eight query tiles");
  constexpr int WV = 8, RB = 128;
  constexpr int QINST = 2 * QT;
  constexpr int QBYTES = QINST * 1024;
  constexpr int QB = 4;
  static_assert((QB - 2) * QINST <= 63, "vmcnt immediate");
  constexpr int SEG = RB * 4 + 16;                 // staging: one query's 128 scores + pad
  __shared__ __attribute__((aligned(1024))) char smem[QB * QBYTES + QT * 16 * SEG];
  char* stg = smem + QB * QBYTES;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nbw = (nblocks - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;   // this workgroup's blocks: blockIdx.x + i * gridDim.x
  const int nmine = (nbw + RT - 1) / RT;                                               // steps of RT blocks
  const int nk = D / 64;
  const int64_t total = (int64_t)nmine * nk;
  const bool qres = nk <= QB;

  if (wave == WV) {
    // ---- producer: the q slices, cyclically, QB-1 steps ahead; joins the barriers of the block epilogues
    const __bf16* pq = qs + (int64_t)lane * 8;
    int hs = 0, hb = 0;
    auto stage_next = [&]() {
      char* sQ = smem + hb * QBYTES;
#pragma unroll
      for (int j = 0; j < QINST; ++j)
        __builtin_amdgcn_global_load_lds((gptr_t)(pq + ((int64_t)hs * QINST + j) * 512), (lptr_t)(sQ + j * 1024), 16, 0, 0);
      hs = hs + 1 == nk ? 0 : hs + 1;
      hb = hb + 1 == QB ? 0 : hb + 1;
    };
    if (qres) {
      for (int p = 0; p < nk; ++p) stage_next();
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    } else {
#pragma unroll
      for (int p = 0; p < QB - 1; ++p)
        if (p < total) stage_next();
    }
    int64_t g = 0;
    for (int j = 0; j < nmine; ++j) {
      if (!qres)
        for (int kt = 0; kt < nk; ++kt, ++g) {
          if (g + QB - 2 < total) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((QB - 2) * QINST) : "memory");
          else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
          if (g + QB - 1 < total) stage_next();
        }
#pragma unroll
      for (int a = 0; a < RT; ++a)
        if (j * RT + a < nbw) { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_s_barrier(); }
    }
    return;
  }

  // ---- consumers
  const int fi = lane & 15, fq = lane >> 4;
  auto blk_of = [&](int i) -> int {                // launch index -> 128-row block
    if (bmode == 1) { const unsigned int u = (unsigned int)i / (unsigned int)unit; return (int)(u * ss * unit + ((unsigned int)i - u * unit)); }
    if (bmode != 2) return i;
    const unsigned int u = (unsigned int)i / (unsigned int)unit, g = u / (unsigned int)(ss - 1);
    return (int)((g * ss + 1 + (u - g * (ss - 1))) * unit + ((unsigned int)i - u * unit));
  };
  const int last_blk = (int)((N - 1) >> 7);
  auto li_of = [&](int j, int a) -> int { return (int)blockIdx.x + min(j * RT + a, nbw - 1) * (int)gridDim.x; };
  auto base_of = [&](int i) -> const bf16x8* {
    const int b = min(blk_of(i), last_blk);
    return (const bf16x8*)(Xb + ((int64_t)b * (D / 64)) * 8192 + wave * 1024) + lane;
  };
  f32x4 acc[RT][QT];
#pragma unroll
  for (int a = 0; a < RT; ++a)
#pragma unroll
    for (int b = 0; b < QT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  int pf_j = 0, pf_kt = 0;
  const bf16x8* pfp[RT];
#pragma unroll
  for (int a = 0; a < RT; ++a) pfp[a] = base_of(li_of(0, a));
  bf16x8 xf[PF][RT][2];
  auto fetch = [&](int slot) {
#pragma unroll
    for (int a = 0; a < RT; ++a) {
      xf[slot][a][0] = __builtin_nontemporal_load(pfp[a]);
      xf[slot][a][1] = __builtin_nontemporal_load(pfp[a] + 64);
    }
    if (++pf_kt == nk) {
      pf_kt = 0;
      pf_j = min(pf_j + 1, nmine - 1);
#pragma unroll
      for (int a = 0; a < RT; ++a) pfp[a] = base_of(li_of(pf_j, a));
    } else {
#pragma unroll
      for (int a = 0; a < RT; ++a) pfp[a] += 1024;
    }
  };
#pragma unroll
  for (int p = 0; p < PF - 1; ++p) {
    fetch(p);
    __builtin_amdgcn_sched_barrier(0);
  }
  int qb = 0;
  const int qper = qres ? nk : QB;
  if (qres) {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();                  // the whole q has landed
    __builtin_amdgcn_sched_barrier(0);
  }
  for (int j = 0; j < nmine; ++j) {
    for (int kt0 = 0; kt0 < nk; kt0 += PF) {
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        fetch((u + PF - 1) % PF);
        __builtin_amdgcn_sched_barrier(0);
        if (!qres) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const char* sQ = smem + qb * QBYTES + lane * 16;
        qb = qb + 1 == qper ? 0 : qb + 1;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int b = 0; b < QT; ++b) {
            const bf16x8 qf = *(const bf16x8*)(sQ + (ks * QT + b) * 1024);
#pragma unroll
            for (int a = 0; a < RT; ++a) acc[a][b] = mfma_f16(xf[u][a][ks], qf, acc[a][b]);
          }
      }
    }
    // ---- block epilogue: the wave groups' maxima straight from registers, the scores through the staging region
#pragma unroll
    for (int a = 0; a < RT; ++a) {
      if (j * RT + a >= nbw) break;                // (uniform over the workgroup, producer included)
      const int li = li_of(j, a);
      const int64_t n0 = (int64_t)blk_of(li) * RB, n = n0 + wave * 16 + fq * 4;
#pragma unroll
      for (int b = 0; b < QT; ++b) {
        float mx = -FLT_MAX;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (n + e >= N) acc[a][b][e] = -FLT_MAX;
          mx = fmaxf(mx, acc[a][b][e]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const int qi = b * 16 + fi;
        if (fq == 0 && qi < nq) gmax[(int64_t)qi * (8 * (int64_t)nblk_ld) + (int64_t)li * 8 + wave] = mx;
      }
      __builtin_amdgcn_s_barrier();                // the previous block's scores have left the staging region
#pragma unroll
      for (int b = 0; b < QT; ++b) {
        *(f32x4*)(stg + (b * 16 + fi) * SEG + (wave * 16 + fq * 4) * 4) = acc[a][b];
        acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                // staged
      for (int idx = tid; idx < QT * 16 * (RB / 4); idx += 512) {
        const int ql = idx / (RB / 4), c = idx % (RB / 4);
        if (ql < nq) __builtin_nontemporal_store(*(const f32x4*)(stg + ql * SEG + c * 16), (f32x4*)(scores + (int64_t)ql * ld + (int64_t)li * RB + c * 4));
      }
    }
  }
}

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
      // (LRX_EMIT_PERSIST_MIN_BPC, dev builds: main passes with fewer blocks per CU than this run one workgroup per block)
      static const int persist_min_bpc = lrx_dev_knob("LRX_EMIT_PERSIST_MIN_BPC", 0);
      const bool persistent = emit && gate == nullptr && (dim / 64) % XPF == 0 && fm.bmode != 1 && nwg < (1ll << 31) && nwg >= (int64_t)persist_min_bpc * n_cu;
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

// ---------------------------------------------------------------------------------------------------------------
// Bounded two-pass search (rows with known bounds R >= max |x_row| and E >= max |x_row - fp16(x_row)|): the same exact top-k at close
// to ONE pass over the fp16 shadow of the shard, without ever writing a [queries, rows] score matrix.
//   filter  s~ = fp16(q) . fp16(x) with ONE f16 MFMA product (instead of six bf16 ones), fp32 accumulation.  With q~ = fp16(q), x~ = fp16(x):
//           s - s~ = (q - q~).x + q~.(x - x~) + (accumulation error), so by Cauchy-Schwarz
//             |s - s~| <= eps(q) = |q - q~| R + |q~| E + (D + 32) 2^-23 |q~| R          (query_eps_block; |q - q~| and |q~| are computed
//           from the actual query, E from the actual rows at commit: ~7e-4 |q| R for normalised rows at D = 2048 -- a third of it the
//           accumulation term; the bf16 filter of round 2 had 3.7e-3).
//   sample  every ss-th 128-row block is scored first, into a small compact matrix; T' = its k-th largest score is a lower bound of
//           kth~, the k-th largest filter score of the whole shard (k sample rows reach it).
//   main    all other blocks; the epilogue keeps only rows with s~ >= T' - 2 eps, appended to a per-query candidate list (~1e-3 of the
//           rows).  Every row of the exact top-k is in the list: its exact score is >= the k-th largest exact score >= the k-th
//           largest of (s~ - eps), so its s~ >= kth~ - 2 eps >= T' - 2 eps.
//   refine  kth~ = k-th largest s~ of the list (exact: the list holds every row >= T' - 2 eps), the rows with s~ >= kth~ - 2 eps
//           are rescored exactly from the fp32 rows (fp64 accumulation, rounded once to fp32) by REF_SPLIT workgroups per query and
//           sorted (score desc, row asc).  Typical band content at 1M x 2048 normalised rows, k = 100: ~130 rows.
//   fallback: a query whose list or band overflows (near-duplicate corpora) raises a device flag; the six-product pass + select +
//           rescore are always enqueued behind it, gated on that flag (they return at once when it is 0), and overwrite only the
//           flagged queries.  No host synchronisation anywhere.
//   Shards below 16 Ki rows keep the score-matrix filter (two launches less in the dependency chain).  Both give the same result --
//   everything ends in the same exact rescoring of a superset of the top-k.
// ---------------------------------------------------------------------------------------------------------------
#define REF_CAND 4096
#define REF_BLK 8192
#ifndef REF_SPLIT
#define REF_SPLIT 4                       // workgroups per query (phase stamps: the exact rescoring is bound by what ONE CU can fetch)
#endif
#define REF_PCAND (REF_CAND / REF_SPLIT)  // candidate capacity of one part
#define REF_PBLK (REF_BLK / REF_SPLIT)
#define REF_QLDS 8192                     // query rows up to this many floats are staged in LDS by the refine kernels

struct RadixShared {
  unsigned int hist[16][256];
  unsigned int bucket, kk, cnt;
};

// eps(q) of the header comment; all threads of the (<= 1024-thread) block take part, fixed summation order.  Optionally stages the
// query row in LDS (s_q).  bounds = {R, E}; E <= 0 means "not measured": bounded from R below.
__device__ float query_eps_block(const float* __restrict__ qglob, int D, const float* __restrict__ bounds, float* s_q, float* s_red /* 32 */) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  float a = 0.f, b = 0.f;
  for (int i = tid; i < D; i += blockDim.x) {
    const float v = qglob[i];
    if (s_q != nullptr) s_q[i] = v;
    const float r = (float)f2h_sat(v), d = v - r;
    a += r * r;
    b += d * d;
  }
  a = wave_sum(a);
  b = wave_sum(b);
  if (lane == 0) { s_red[wave] = a; s_red[16 + wave] = b; }
  __syncthreads();
  float A = 0.f, B = 0.f;
  for (int w = 0; w < nw; ++w) { A += s_red[w]; B += s_red[16 + w]; }
  __syncthreads();
  // E <= 0 = "not measured" (a C / torch-op caller passing {R, 0} with a shadow; FlatIPIndex always maintains E).  Still a BOUND: an element
  // inside fp16's normal range is off by <= 2^-11 |x|, a subnormal one (|x| < 2^-14) by <= 2^-25, so |row - fp16(row)| <= 2^-11 R +
  // sqrt(D) 2^-25 whenever no element can exceed 65504, i.e. R <= 65504; beyond that nothing is known about the saturated elements: E = R,
  // the band is useless and the query takes the rigorous six-product fallback.
  const float R = bounds[0];
  const float E = bounds[1] > 0.f ? bounds[1] : (R <= 65504.f ? R * 0.00048828125f + sqrtf((float)D) * 2.9802322e-8f : R);
  const float accum = (float)(D + 32) * 1.1920929e-7f;   // 2^-23 per accumulated term
  return (sqrtf(B) * R + sqrtf(A) * (E + accum * R * 1.01f)) * 1.0001f + 1e-30f;
}

// Sample step of the score-free filter (one workgroup per query): T' = a lower bound of the k-th largest filter score of the shard taken from
// the compact sample scores, thr = T' - 2 eps, and the sample rows reaching thr open the query's candidate list.  Sample-local row j is
// corpus row (j / rb) * ss * rb + j % rb.  gsz = rows per entry of `blkmax`:
//   128: T' = the k-th largest sample score (select_topk_sorted over the block maxima + the qualifying blocks);
//   16 (register-streaming kernels: maxima of the 16-row wave groups, row stride 8 * nblk_ld): T' = the k-th largest GROUP maximum -- k
//       different rows reach it, so it is a lower bound too, and with ~30 groups per wanted row it is the ~(1.02 k)-th score: one radix
//       select over nblk * 8 values instead of select + gather + sort over the scores (40 -> 15 us at 1M x 2048, k = 100; 78 -> 41 us at 10M x 256).
// (device function: one workgroup of 256 .. 1024 threads works on query qi -- k_sample_threshold below, and the selection step inside the fused
// filter kernel.  SORTED = false compiles the select_topk_sorted branch out (1024-thread code; the fused kernel's plan guarantees >= k groups).)
struct ThrShared {
  SelShared sh;
  float s_red[32];
  unsigned int s_fill;
};
template <bool SORTED, bool COH = false>
__device__ __forceinline__ void sample_threshold_query(ThrShared& ts, int qi, const float* __restrict__ scores, int64_t ld_s, int64_t Ns, int k,
                   const float* __restrict__ blkmax, int nblk, int nblk_ld,
                   const float* __restrict__ q, int D, const float* __restrict__ bounds, int rb, int ss, int64_t N, float* __restrict__ thr_out,
                   float* __restrict__ eps_out, unsigned long long* __restrict__ cand, unsigned int* __restrict__ cnt, int gsz, unsigned int cap) {
  SelShared& sh = ts.sh;
  float* s_red = ts.s_red;
  unsigned int& s_fill = ts.s_fill;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, NT = blockDim.x;
  const float* row = scores + (int64_t)qi * ld_s;
  const int ng = gsz == 16 ? nblk * 8 : nblk;                    // entries of this query's maxima
  const float* bm = blkmax + (int64_t)qi * (gsz == 16 ? 8 * (int64_t)nblk_ld : (int64_t)nblk_ld);
  float kth;
  if (gsz == 16 && nblk >= 8 * k && nblk <= 2 * SEL_CAND) {
    // a large sample (10M x 256: 31 k groups): the k-th largest BLOCK maximum is as good a bound (k different rows reach it; with >= 8 k
    // blocks it is the ~(1.06 k)-th score) and the four passes of its select run over an LDS copy of 1/8 of the values (43 -> 22 us)
    float* bmaxL = (float*)sh.cand;
    for (int b = tid; b < nblk; b += NT) {
      const f32x4 g0 = ld4<COH>(bm + (int64_t)b * 8), g1 = ld4<COH>(bm + (int64_t)b * 8 + 4);
      bmaxL[b] = fmaxf(fmaxf(fmaxf(g0[0], g0[1]), fmaxf(g0[2], g0[3])), fmaxf(fmaxf(g1[0], g1[1]), fmaxf(g1[2], g1[3])));
    }
    for (int b = nblk + tid; b < ((nblk + 3) & ~3); b += NT) bmaxL[b] = -FLT_MAX;
    __syncthreads();
    unsigned int ne, nq_;
    kth = key2f(radix_select_kth(bmaxL, nblk, (unsigned int)k, sh, ne, nq_));
  } else if (gsz == 16 && ng >= k && ng <= 4 * NT) {
    kth = key2f(radix_select_kth_small<SelShared, COH>(bm, ng, (unsigned int)k, sh));
  } else if (gsz == 16 && ng >= k) {
    unsigned int ne, nq_;
    kth = key2f(radix_select_kth<SelShared, COH>(bm, ng, (unsigned int)k, sh, ne, nq_));
  } else if constexpr (SORTED) {
    // (group maxima: fewer than k groups -- a shard of a few thousand rows -- fall back to the scores themselves, without block pruning)
    select_topk_sorted(row, Ns, k, gsz == 16 ? nullptr : bm, gsz == 16 ? 0 : nblk, sh);   // (the plan guarantees >= 2k valid sample rows)
    kth = key2f((uint32_t)(sh.cand[k - 1] >> 32));
  } else {
    kth = -FLT_MAX;                                              // (not reachable: plan_chunk admits the fused launch only with >= k sample groups)
  }
  __syncthreads();
  const float eps = query_eps_block(q + (int64_t)qi * D, D, bounds, nullptr, s_red);
  const float thr = kth - 2.0f * eps;
  // this workgroup is the only writer of the query's list until the main pass starts: slots come from an LDS counter (a global
  // atomic per hit cost ~2 us of round trip per qualifying block and wave: 49 -> 3x us for the kernel), the count is stored once
  unsigned long long* list = cand + (int64_t)qi * cap;
  if (tid == 0) { s_fill = 0; sh.neq = 0; }
  __syncthreads();
  if (gsz == 16) {
    // qualifying 16-row groups first (all threads), then their rows, 16 lanes per group
    unsigned int* glist = (unsigned int*)sh.eqs;                 // 2 * SEL_EQCAP entries
    for (int g = tid; g < ng; g += NT)
      if (ld1<COH>(bm + g) >= thr) {
        const unsigned int p = atomicAdd(&sh.neq, 1u);
        if (p < 2 * SEL_EQCAP) glist[p] = (unsigned int)g;
      }
    __syncthreads();
    const unsigned int ngl = sh.neq;
    if (ngl > 2 * SEL_EQCAP) {                                   // (near-duplicate rows: more groups than any list would hold -> exact fallback)
      if (tid == 0) s_fill = cap + 1;
    } else {
      for (unsigned int idx = tid; idx < ngl * 16; idx += NT) {
        const int64_t j = (int64_t)glist[idx >> 4] * 16 + (idx & 15);
        const float v = ld1<COH>(row + j);
        const int64_t g = (j / rb) * ((int64_t)ss * rb) + (j % rb);
        if (g < N && v >= thr) {
          const unsigned int p = atomicAdd(&s_fill, 1u);
          if (p < cap) list[p] = sel_pack(f2key(v), g);
        }
      }
    }
  } else {
    for (int b = wave; b < nblk; b += NT / 64)
      if (bm[b] >= thr) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int64_t j = (int64_t)b * SP_ROWS + h * 64 + lane;
          const float v = row[j];
          const int64_t g = (j / rb) * ((int64_t)ss * rb) + (j % rb);
          if (g < N && v >= thr) {
            const unsigned int p = atomicAdd(&s_fill, 1u);
            if (p < cap) list[p] = sel_pack(f2key(v), g);
          }
        }
      }
  }
  __syncthreads();
  // Round 4 -- clumpy samples.  The k-th largest GROUP maximum equals the ~(1.02 k)-th score only when high scores are spread over the
  // groups (iid rows).  On a corpus stored cluster by cluster the rows of a 16-row group score alike: the k best groups then span the ~k/8
  // best sampled blocks instead of the k best rows, T' drops to the score of a far worse cluster, and every member of every better cluster
  // -- tens of thousands of rows per query -- passes the filter (measured: 18 k hits per query, 69 of 100 queries over the list capacity
  // at 1M x 2048 in 1 000 contiguous clusters).  The rows just collected are ALL sample rows >= T'_group - 2 eps, so when there are many
  // more than k of them their k-th largest IS the k-th largest sample score: a radix select over the short list gives the row-exact
  // bound.  The list keeps its extra entries (the refine step selects by score anyway).  iid rows never take this branch (~1.3 k entries).
  const unsigned int nfill = s_fill;
  float thr_final = thr;
  if (gsz == 16 && nfill > 2u * (unsigned int)k && nfill <= cap)
    thr_final = fmaxf(thr, key2f(radix_select_kth_list(list, (int)nfill, (unsigned int)k, sh)) - 2.0f * eps);
  if (tid == 0) {
    if constexpr (COH) {                // read by other workgroups of this launch (the main phase): device-scope stores
      st1_coh(thr_out + qi, thr_final);
      st1_coh(eps_out + qi, eps);
      __hip_atomic_store(cnt + qi * CNT_STRIDE, nfill, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      thr_out[qi] = thr_final;
      eps_out[qi] = eps;
      cnt[qi * CNT_STRIDE] = nfill;       // (the chunk's memset zeroed it; nobody else has touched it yet)
    }
  }
}

__global__ void __launch_bounds__(SEL_THREADS)
k_sample_threshold(const float* __restrict__ scores, int64_t ld_s, int64_t Ns, int k, const float* __restrict__ blkmax, int nblk, int nblk_ld,
                   const float* __restrict__ q, int D, const float* __restrict__ bounds, int rb, int ss, int64_t N, float* __restrict__ thr_out,
                   float* __restrict__ eps_out, unsigned long long* __restrict__ cand, unsigned int* __restrict__ cnt, int gsz, unsigned int cap) {
  __shared__ ThrShared ts;
  sample_threshold_query<true>(ts, blockIdx.x, scores, ld_s, Ns, k, blkmax, nblk, nblk_ld, q, D, bounds, rb, ss, N, thr_out, eps_out, cand, cnt, gsz, cap);
}

// ---- the fused filter kernel (see filter_sample_block above for the design): selection + the two passes in one persistent launch
template <int QT>
struct FusedLds {
  static constexpr int SAMPLE = ((QT > 8 ? 2 : 4) * 2 * QT * 1024) > 16 * (128 * 4 + 16) ? ((QT > 8 ? 2 : 4) * 2 * QT * 1024) : 16 * (128 * 4 + 16);
  static constexpr int A = EmitLds<QT>::BYTES > SAMPLE ? EmitLds<QT>::BYTES : SAMPLE;
  static constexpr int BYTES = A > (int)sizeof(ThrShared) + 64 ? A : (int)sizeof(ThrShared) + 64;
};

template <int QT, int PF, int RT>
__global__ void __launch_bounds__(576, 3)
k_filter_fused(const __bf16* __restrict__ Xb, int64_t N, int D, const __bf16* __restrict__ qs, int nq, float* __restrict__ scores, int64_t ld_s,
               float* __restrict__ gmax, int nblk_s, int nblk_ld_s, int nsamp, int nmain, int ss, int k, const float* __restrict__ qf32,
               const float* __restrict__ bounds, float* __restrict__ thr, float* __restrict__ eps, unsigned long long* __restrict__ cand,
               unsigned int* __restrict__ cnt, unsigned int cap, FusedCtl* __restrict__ ctl, int phases) {
  __shared__ __attribute__((aligned(1024))) char smem[FusedLds<QT>::BYTES];
  __shared__ unsigned int s_item;
  const int tid = threadIdx.x;
  // (phases: bit 0 = S, 1 = T, 2 = M -- all three in the product; LRX_FUSED_PHASES = 1 or 3 lets the old kernels take over the later ones (bisecting
  // aid), bit 7 records the phase timestamps lrx_probe_fused_timestamps reads)
  unsigned long long* ts = (phases & 128) && blockIdx.x < 1024 ? g_fused_ts + blockIdx.x * 8 : nullptr;
  if (ts != nullptr && tid == 0) { ts[0] = __builtin_amdgcn_s_memrealtime(); ts[2] = 0; ts[4] = 0; ts[5] = 0; }
  // (Claim loops: ONE single-thread region per iteration, in the middle of the loop body.  With "if (tid == 0) count; } ... top: if (tid == 0)
  // claim" the compiler merged the two regions across the back edge and structurised the result as nested exec-mask loops -- lanes 1..63 of
  // wave 0 then ran on through the barriers of the next iteration before lane 0 had claimed its item: the first version of this kernel hung.)
  // ---- S: sample blocks
  if (phases & 1) {
    if (tid == 0) s_item = atomicAdd(&ctl->ctr_s, 1u);
    for (;;) {
      __syncthreads();
      const unsigned int li = __builtin_amdgcn_readfirstlane(s_item);   // (scalar: the loop exit is a uniform branch)
      if (li >= (unsigned int)nsamp) break;
      filter_sample_block<QT, PF>(smem, Xb, N, D, qs, nq, scores, ld_s, gmax, nblk_ld_s, (int64_t)li * ss, (int64_t)li);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // every thread's device-scope score / maxima stores have landed ...
      __syncthreads();                                            // (also: every thread has read s_item)
      if (tid == 0) {
        __hip_atomic_fetch_add(&ctl->done_s, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ... before the block is counted
        s_item = atomicAdd(&ctl->ctr_s, 1u);
      }
    }
  }
  __syncthreads();
  if (ts != nullptr && tid == 0) ts[1] = __builtin_amdgcn_s_memrealtime();
  // ---- T: selection for the queries this workgroup can claim (first look: nothing claimed -> nothing to wait for)
  if (phases & 2) {
    if (tid == 0) s_item = atomicAdd(&ctl->ctr_t, 1u);
    for (;;) {
      __syncthreads();
      const unsigned int qi = __builtin_amdgcn_readfirstlane(s_item);
      if (qi >= (unsigned int)nq) break;
      if (tid == 0)
        while (__hip_atomic_load(&ctl->done_s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned int)nsamp) __builtin_amdgcn_s_sleep(8);
      __syncthreads();
      if (ts != nullptr && tid == 0 && ts[2] == 0) ts[2] = __builtin_amdgcn_s_memrealtime();
      sample_threshold_query<false, true>(*(ThrShared*)smem, (int)qi, scores, ld_s, (int64_t)nsamp * 128, k, gmax, nblk_s, nblk_ld_s, qf32, D, bounds, 128, ss, N,
                                    thr, eps, cand, cnt, 16, cap);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // (thread 0's device-scope stores of thr / eps / list count have landed)
      __syncthreads();
      if (tid == 0) {
        __hip_atomic_fetch_add(&ctl->done_t, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_item = atomicAdd(&ctl->ctr_t, 1u);
      }
    }
  }
  __syncthreads();
  if (ts != nullptr && tid == 0) ts[3] = __builtin_amdgcn_s_memrealtime();
  // ---- M: this workgroup's share of the other blocks (static: nobody waits for a main block)
  if ((phases & 4) && nmain > 0 && (int)blockIdx.x < (nmain + RT - 1) / RT)
    filter_emit_body<QT, PF, RT>(smem, Xb, N, D, qs, nq, nmain, 2, ss, 1, thr, cand, cnt, cap, &ctl->done_t, (unsigned int)nq, ts);
  if (ts != nullptr && tid == 0) ts[6] = __builtin_amdgcn_s_memrealtime();
}

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
  // (LRX_SS_MAX / LRX_SS_FORCE, dev builds: A/B runs of the sample stride on one box; read once, thread-safe)
  static const int ss_force = lrx_dev_knob("LRX_SS_FORCE", 0) >= 2 ? lrx_dev_knob("LRX_SS_FORCE", 0) : 0;
  static const int ss_max = lrx_dev_knob("LRX_SS_MAX", 0) >= 2 ? lrx_dev_knob("LRX_SS_MAX", 0) : SAMPLE_SS_MAX;
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
      const int unit = p.gemm ? 2 : 1;                        // sample units of 256 rows = two blocks of the 128-row filter kernel
      FilterMode fs;
      fs.bmode = 1; fs.ss = p.ss; fs.unit = unit; fs.nblocks = p.nsamp_wg * unit;
      fs.group_max = shadow;                                  // the register-streaming kernels hand over the maxima of their 16-row wave groups
      if (clear_in_pack) { fs.zero = flg; fs.nzero = (int)nclear; }
      fs.presplit = presplit;
      FusedArgs fa;
      if (p.fused && clear_in_pack) {
        fa.ld_s = p.ld_s; fa.nblk_s = (int)p.nblk_s; fa.nblk_ld_s = (int)p.nblk_ld_s; fa.nsamp = (int)p.nsamp_wg; fa.nmain = (int)p.nmain_wg; fa.ss = p.ss; fa.k = k;
        fa.bounds = row_bounds; fa.thr = thr; fa.eps = eps; fa.cand = cand; fa.cnt = cnt; fa.cap = p.cap;
        fa.ctl = (FusedCtl*)(flg + ints_before_cnt(nq) - 8);
        static const int fused_phases = []() { const int v = lrx_dev_knob("LRX_FUSED_PHASES", 7); return (v & 7) == 1 || (v & 7) == 3 || (v & 7) == 7 ? (v & 135) : 7; }();   // (dev builds)
        fa.phases = fused_phases;
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
      if (fs.fused == nullptr || !(fa.phases & 2))
      hipLaunchKernelGGL(k_sample_threshold, dim3(nq), dim3(SEL_THREADS), 0, s, (const float*)scores, p.ld_s, p.nsamp_wg * p.rb, k, (const float*)blkmax,
                         (int)p.nblk_s, (int)p.nblk_ld_s, qc, dim, row_bounds, p.rb, p.ss, n_rows, thr, eps, cand, cnt, fs.group_max ? 16 : 128, p.cap);
      LRX_LAUNCH_CHECK();
      if (fs.fused != nullptr && (fa.phases & 4)) {
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
