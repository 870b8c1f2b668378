// liblrx search, part 1 -- SCORE / FILTER kernels: what a search streams the shard through (map: head of lrx_search.hip, section A).
// Part of the ONE translation unit lrx_search.hip (included there, in source order: filter kernels -> selection -> bounded-search
// device code -> refine kernels; the host driver, the shard maintenance and the exchange kernels stay in lrx_search.hip).  Not a stand-alone header.
#pragma once

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

