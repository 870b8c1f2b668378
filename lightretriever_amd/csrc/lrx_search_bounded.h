// liblrx search, part 3 -- the BOUNDED two-pass search's device side: error bound, k_sample_threshold, the fused filter launch (map: section C).
// Part of the ONE translation unit lrx_search.hip (included there, in source order: filter kernels -> selection -> bounded-search
// device code -> refine kernels; the host driver, the shard maintenance and the exchange kernels stay in lrx_search.hip).  Not a stand-alone header.
#pragma once

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
// MEMORY ORDERING: data one workgroup publishes for another INSIDE this launch (sample scores, group maxima, thr, eps, list
// counts, the done_s / done_t counters) moves by device-scope (sc1) stores and loads issued in program order, with RELAXED atomics on the
// counters and `s_waitcnt vmcnt(0)` between a workgroup's data stores and its counter increment.  That is NOT the HIP / LLVM memory model's
// release / acquire pairing -- it relies on gfx9-family behaviour: a wave's vector-memory stores to device-coherent lines reach the L2 / fabric
// in issue order once vmcnt has drained, and sc1 loads bypass the non-coherent per-XCD L2 lines.  Agent-scope release / acquire was measured:
// buffer_wbl2 / buffer_inv per wave put the launch at 0.31 ms against 0.18 ms for the three-launch chain (0.24 with one fencing wave per
// workgroup), profiles/r05_fused_ab.txt.  Hence the assumption is pinned to the architecture this library is written for:
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "k_filter_fused: the cross-workgroup publication protocol is validated on gfx942 / gfx950 only (see the comment above)"
#endif
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
  // (phases: bit 0 = S, 1 = T, 2 = M -- the host always passes all three; bit 7 (dev builds: LRX_FUSED_PHASES=128) records the phase timestamps
  // lrx_probe_fused_timestamps reads)
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

