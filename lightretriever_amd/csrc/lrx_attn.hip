// Varlen causal GQA attention forward for packed sequences (S <= max_positions), gfx950 MFMA 32x32x16; q, k, v, P in fp16 (see attn_mfma), output bf16.
//
// Work item = (sequence, 64-row q tile, kv head).  Workgroup = 2*GRP waves: wave w handles q head (kvh*GRP + w%GRP)
// and the 32-row half (w / GRP) of the q tile, so the K/V tile staged in LDS is shared by all GRP q heads of the group.
//
// Per wave, per 32-key sub-tile:
//   S^T[key, q] = K . Q^T      (A = K rows from LDS via ds_read_b128, B = Q fragments kept in registers)
//     -> the score column of one q row lives in ONE lane (16 regs) + its lane^32 partner: row max / row sum are
//        15 VALU ops + one cross-half shuffle, no LDS.
//   P^T (fp16) is exactly the B-operand layout of the next MFMA (accumulator tile as operand), so
//   O^T[d, q] += V^T . P^T     (A = V^T fragments read from the row-major V tile with ds_read_b64_tr_b16)
//   online softmax state (m, l) is one scalar per lane.
//
// K/V tiles (64 keys) are staged with 16-byte global_load_lds into two LDS stages; the K image is XOR-swizzled for
// conflict-free ds_read_b128 row reads, the V image for conflict-free transposed reads (swizzle applied on the
// per-lane source address, undone on the read).
#include "lrx_common.h"
#include <type_traits>

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;

// q, k, v and the softmax probabilities are FP16 (round 3): 11 significant bits instead of bf16's 8 on both MFMA operands of QK^T and
// P.V.  tools/exp/rounding_budget.py: at the 32-layer Llama-3.1-8B dims the bf16 roundings of q|k|v were the largest single share of
// 1 - cos against the fp32 model (9e-4 of 2.4e-3 with the double rounding around RoPE).  The fused QKV + RoPE epilogue writes fp16
// (saturating at +-65504), the 16-bit containers below stay typed bf16x8 (LDS-DMA, swizzles and transposing reads move bits) and are
// re-interpreted at the MFMA; the attention OUTPUT stays bf16 (it is the bf16 A operand of the O-projection GEMM).
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
__device__ __forceinline__ f32x16 attn_mfma(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ __bf16 attn_cvt(float v) { return __builtin_bit_cast(__bf16, (_Float16)v); }      // fp16 bits in the 16-bit container
__device__ __forceinline__ float attn_f(__bf16 v) { return (float)__builtin_bit_cast(_Float16, v); }
typedef __attribute__((address_space(3))) const char* lds_char_ptr;

template <int D>
struct AttnGeom {
  static constexpr int ROW_BYTES = D * 2;          // bytes per key row in LDS
  static constexpr int CH = D / 8;                 // 16-B chunks per row
  static constexpr int TILE_BYTES = 64 * ROW_BYTES;
  static constexpr int ROWS_PER_INST = 1024 / ROW_BYTES;  // rows filled by one LDS-DMA wave instruction
  static constexpr int INSTS = TILE_BYTES / 1024;
  __device__ static __forceinline__ int xk(int row) { return D == 64 ? ((row >> 1) & 7) : (row & 15); }
  __device__ static __forceinline__ int xv(int row) { return D == 64 ? (((row >> 1) & 1) << 2) : ((row & 3) << 2); }
};

#define LAZY_T 8.0f   // exponent head-room of the lazy softmax reference maximum

constexpr int attn_dma_waves(int insts, int nw) {   // largest divisor of insts that is <= nw
  int d = nw < insts ? nw : insts;
  while (insts % d != 0) --d;
  return d;
}

// the 16-bit output element: bf16, or (out_f16, the precise stream's fp16 GEMM operands) fp16 -- |o| <= max |v| <= 65504: no saturation to count
__device__ __forceinline__ __bf16 o16(float v, int out_f16) { return out_f16 ? __builtin_bit_cast(__bf16, (_Float16)v) : f2bf(v); }

template <int D, int GRP>
__global__ void __launch_bounds__(128 * GRP)
k_attn_varlen_causal(const __bf16* __restrict__ qkv, const int32_t* __restrict__ cu, int nqt, int nq, int nkv,
                     __bf16* __restrict__ out, float scale_log2, int last_tile_only, int nparts, int n_seqs, int n_items, int gs, int out_f16) {
  using G = AttnGeom<D>;
  constexpr int NW = 2 * GRP;
  constexpr int KS = D / 16;  // k-steps of the QK^T product
  constexpr int DT = D / 32;  // 32-row tiles of O^T
  // K/V ring: 3 stages with a counted vmcnt (two tiles in flight; a tile's compute, ~0.5 us, is shorter than the load
  // latency, so one tile of prefetch leaves every barrier waiting on HBM).
  // The first DW waves (a divisor of the instruction count, e.g. 8 of the 12 waves of a GQA-6 group) issue the LDS-DMA, the same
  // number each, so the counted wait is one immediate for everybody (waves without loads have nothing outstanding).
  // (Tried: running the workgroup's two halves half a tile apart, two barriers per tile, so that one half's softmax sits beside
  // the other's MFMAs -- 1.143 -> 1.173 ms at d = 128: the loop is not bound by that pairing, see the ablation in DESIGN.md.)
  constexpr int DW = attn_dma_waves(G::INSTS, NW);
  constexpr int NST = 3;
  static_assert(NST * 2 * G::TILE_BYTES <= 96 * 1024, "K/V ring");
  constexpr int PER_TILE = 2 * (G::INSTS / DW);  // LDS-DMA instructions per issuing wave per tile (K + V)
  // Per-wave staging block (32 rows x D bf16, private to the wave): the wave's Q rows arrive in it by LDS-DMA (whole 2D-byte row
  // segments per request instead of 32 rows x 32 B per load instruction) and its O rows leave through it (16 B per lane, whole
  // row segments per store instruction instead of 32 rows x 16 B).  Ablation at d = 128: the 8-byte-piece stores cost 0.20 ms and
  // the row-gather Q loads 0.16 ms of a 1.25 ms launch.
  constexpr int QO_BYTES = 32 * G::ROW_BYTES;
  constexpr int QINST = QO_BYTES / 1024;         // LDS-DMA instructions per Q block
  __shared__ __attribute__((aligned(1024))) char smem[NST * 2 * G::TILE_BYTES + NW * QO_BYTES];  // [stage][K|V] | [wave] Q/O block

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform: branches on it stay scalar
  const int r = lane & 31, h = lane >> 5;
  const int64_t RS = (int64_t)(nq + 2 * nkv) * D;
  const int grp_total = nq / nkv;

  // Persistent workgroups: the grid is one workgroup per CU-slot, each walks a list of work items (sequence, kv head[, part], q tile).
  // (A 96-KiB-LDS workgroup is alone on its CU; with one item per workgroup the CU sat empty a third of the time between a
  // workgroup's exit and its successor's first instruction -- PMC: 64 workgroups x 18 us of wave lifetime per CU in a 1.85 ms launch.)
  // The list is built for L2 reuse: the nqt q tiles of one (sequence, kv head) re-read the same K/V tiles (4.5x at S = 512), so they
  // run at the same time on `gs` workgroups of ONE XCD (blocks b and b + 8 share an XCD: observed dispatch rule, speed only) --
  // in q-tile-major order over the whole launch every re-read came from HBM and the load + barrier skeleton alone took 0.77 of
  // the 1.4 ms.  Slot j of a group takes q tiles j, j + gs, ... of its group's current pair, from the long end on even steps and
  // from the short end on odd ones, so every slot sees the same number of K/V tiles over two steps.
  // The K/V tiles of all the items of a workgroup form ONE stream through the ring: the prefetch runs two tiles ahead of the
  // compute across item boundaries (its own walker over the same item list), and the next item's Q fragments are requested during
  // the current item's last tile -- an item boundary costs no load latency.
  const int ny = nkv * nparts;
  const int n_pairs = n_seqs * ny;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int g4 = gs > 0 ? slot / gs : 0, jslot = gs > 0 ? slot % gs : 0, gpx = gs > 0 ? (int)(gridDim.x >> 3) / gs : 1;
  struct Walk { int step, kq, item; };
  struct Item { int pair, qt, s0, len; };      // pair < 0: end of the list
  auto next_item = [&](Walk& w) -> Item {
    for (;;) {
      int pair, qt_sel = -1;
      if (gs > 0) {
        // v-th (kv head of a sequence, part) of this XCD: the parts of one kv head (they read the same K/V) sit in neighbouring
        // groups of the same XCD
        const int v = w.step * gpx + g4;
        pair = ((v / nparts) * 8 + xcd) * nparts + v % nparts;
        if ((v / nparts) * 8 >= n_pairs / nparts) return Item{-1, 0, 0, 0};
        if (pair >= n_pairs) { ++w.step; w.kq = 0; continue; }
        const int idx = jslot + w.kq * gs;
        if (idx >= nqt) { ++w.step; w.kq = 0; continue; }
        qt_sel = (w.step & 1) ? idx : nqt - 1 - idx;
        ++w.kq;
      } else {                         // one item per (pair[, q tile]), round-robin (last-tile mode, odd grids)
        if (w.item >= n_items) return Item{-1, 0, 0, 0};
        pair = last_tile_only ? w.item : w.item % n_pairs;
        if (!last_tile_only) qt_sel = nqt - 1 - w.item / n_pairs;
        w.item += gridDim.x;
      }
      const int b = pair / ny;
      const int s0 = cu[b], len = cu[b + 1] - s0;
      // last_tile_only: one q tile per sequence, the one holding its last token (all the pooled path needs of the last layer)
      const int qt = last_tile_only ? ((len - 1) >> 6) : qt_sel;
      if (len <= 0 || qt * 64 >= len) continue;
      return Item{pair, qt, s0, len};
    }
  };

  // ---- staging: instruction j (0..INSTS-1) of a tile fills LDS bytes [j*1024, j*1024+1024): slot s = j*64 + lane,
  //      row = s / CH, chunk position cs = s % CH, holding logical chunk cs ^ x(row)
  auto stage = [&](int st, const Item& it, int kt) {
    char* sK = smem + st * (2 * G::TILE_BYTES);
    char* sV = sK + G::TILE_BYTES;
    if (wave >= DW) return;
    const int hk_ = (it.pair % ny) / nparts;
    const __bf16* kbase = qkv + (int64_t)it.s0 * RS + (int64_t)(nq + hk_) * D;
    const __bf16* vbase = kbase + (int64_t)nkv * D;
#pragma unroll
    for (int jj = 0; jj < G::INSTS / DW; ++jj) {
      const int j = wave + jj * DW;
      int s = j * 64 + lane;
      int row = s / G::CH, cs = s % G::CH;
      int grow = min(kt * 64 + row, it.len - 1);
      const __bf16* kp = kbase + (int64_t)grow * RS + ((cs ^ G::xk(row)) << 3);
      const __bf16* vp = vbase + (int64_t)grow * RS + ((cs ^ G::xv(row)) << 3);
      __builtin_amdgcn_global_load_lds((gptr_t)kp, (lptr_t)(sK + j * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)vp, (lptr_t)(sV + j * 1024), 16, 0, 0);
    }
  };
  // prefetch side of the stream
  Walk wp = {0, 0, (int)blockIdx.x};
  Item ip = next_item(wp);
  int ktp = 0, sp_ = 0;                      // next tile of ip to request, ring slot it goes to
  auto stage_next = [&]() -> bool {
    if (ip.pair < 0) return false;
    stage(sp_, ip, ktp);
    sp_ = sp_ == NST - 1 ? 0 : sp_ + 1;
    if (++ktp > ip.qt) { ip = next_item(wp); ktp = 0; }
    return true;
  };

  // ---- lane-constant LDS read offsets: ONE register each, the other k-steps / d tiles by XOR (both swizzles are XORs of the 16-B chunk
  //      index and the chunk's own k-step / d-tile bits are disjoint from the rest): at head_dim 128 this kernel sits at its 256-VGPR
  //      budget, twelve offset registers are ten too many.
  // K row read: row (sub*32 + r), logical chunk 2ks + h;  V transposed read: 16-lane group g = lane>>4, i = lane&15, qd = i>>2, p = i&3;
  // block row = key_base + qd, columns dt*32 + 16*(g&1) + 4p .. +3  ->  logical chunk dt*4 + 2*(g&1) + (p>>1), byte 8*(p&1) inside it
  const int koff0 = r * G::ROW_BYTES + ((h ^ G::xk(r)) << 4);
  int voff0;
  {
    const int g = lane >> 4, i = lane & 15, qd = i >> 2, p = i & 3;
    voff0 = qd * G::ROW_BYTES + (((2 * (g & 1) + (p >> 1)) ^ G::xv(qd)) << 4) + 8 * (p & 1);
  }

  // Q block of an item for this wave: rows q0 .. q0 + 31 of its head, requested into the wave's staging block with the K-tile
  // swizzle (so the fragment reads are the K row reads: koff); fragments Q[q0 + r][16 ks + 8 h .. +7] (B operand)
  char* const sW = smem + NST * 2 * G::TILE_BYTES + wave * QO_BYTES;
  auto request_q = [&](const Item& it) {
    const int yy = it.pair % ny;
    const int hk_ = yy / nparts, part_ = yy - hk_ * nparts;
    const int hig = part_ * GRP + (wave % GRP);
    const int q0_ = it.qt * 64 + (wave / GRP) * 32;
    if (q0_ >= it.len || hig >= grp_total) return;           // (an inactive wave of this item: nothing to fetch)
    const __bf16* qb = qkv + (int64_t)it.s0 * RS + (int64_t)(hk_ * grp_total + hig) * D;
#pragma unroll
    for (int j = 0; j < QINST; ++j) {
      const int s_ = j * 64 + lane;
      const int row = s_ / G::CH, cs = s_ % G::CH;
      const int grow = min(q0_ + row, it.len - 1);
      __builtin_amdgcn_global_load_lds((gptr_t)(qb + (int64_t)grow * RS + ((cs ^ G::xk(row)) << 3)), (lptr_t)(sW + j * 1024), 16, 0, 0);
    }
  };
  auto read_q = [&](bf16x8 (&dst)[KS]) {
    int kb = koff0;
    asm volatile("" : "+v"(kb));              // (not hoisted into KS loop-invariant registers)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) dst[ks] = *(const bf16x8*)(sW + (kb ^ (ks << 5)));
  };

  Walk wc = {0, 0, (int)blockIdx.x};
  Item ic = next_item(wc);
  if (ic.pair < 0) return;
  bf16x8 qf[KS];
  request_q(ic);
  // prologue: two tiles of the stream in flight, the first one landed
  const bool t0 = stage_next(), t1 = stage_next();
  (void)t0;
  if (t1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_TILE) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (DW < NW && wave >= DW) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // waves without tile requests: the counted wait does not cover their Q block
  __builtin_amdgcn_s_barrier();
  read_q(qf);                                    // (requested before the first two tiles: landed with the counted wait above)
  int cur = 0;

  while (ic.pair >= 0) {
  const Item inext = next_item(wc);
  const int yy = ic.pair % ny;
  const int hk = yy / nparts, part = yy - hk * nparts;
  const int s0 = ic.s0, len = ic.len, qt = ic.qt;
  const int qtile0 = qt * 64;
  const int head_in_grp = part * GRP + (wave % GRP);
  const int hq = hk * grp_total + min(head_in_grp, grp_total - 1);
  const int q0 = qtile0 + (wave / GRP) * 32;
  const bool active = q0 < len && head_in_grp < grp_total;

  f32x16 o[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int t = 0; t < 16; ++t) o[dt][t] = 0.f;
  float m = -1e30f, l = 0.f;

  auto qk_product = [&](const char* kt_base, const bf16x8 (&qf_)[KS]) -> f32x16 {
    bf16x8 kf[KS];
    int kb = koff0;
    asm volatile("" : "+v"(kb));
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) kf[ks] = *(const bf16x8*)(kt_base + (kb ^ (ks << 5)));
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // inline-constant C operand
    f32x16 acc = attn_mfma(kf[0], qf_[0], zero);
#pragma unroll
    for (int ks = 1; ks < KS; ++ks) acc = attn_mfma(kf[ks], qf_[ks], acc);
    // (Round 3, measured and not kept: sched_group_barriers that put three fragment reads ahead of the MFMA chain and one read per MFMA
    // after it -- left alone, hipcc walks the eight fragments of a head_dim-128 product through ONE register quad, read / lgkmcnt(0) / MFMA --
    // 1.207 ms against 1.17-1.19: the second wave of the SIMD already covers those waits.)
    return acc;
  };
  // lane <-> lane^32 exchange on the VALU (v_permlane32_swap) instead of an LDS round trip (ds_bpermute)
  auto xhalf_max = [](float x) -> float {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
  };
  auto softmax_pv = [&](f32x16& s, const char* v_base, bool diag) {
    // ---- V^T fragments: issued now (latency hides under the softmax VALU) as inline asm: the ds_read_tr builtin makes
    //      hipcc drain ALL in-flight LDS-DMA (s_waitcnt vmcnt(0)) before every read, serialising the prefetch ring.
    s16x4 vt[2][DT][2];
    {
      const uint32_t vb = (uint32_t)(uintptr_t)(lds_char_ptr)(v_base + 4 * h * G::ROW_BYTES);
      int vo = voff0;
      asm volatile("" : "+v"(vo));
#pragma unroll
      for (int sp = 0; sp < 2; ++sp)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          const uint32_t a = vb + sp * 16 * G::ROW_BYTES + (vo ^ (dt << 6));
          asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(vt[sp][dt][0]) : "v"(a));
          asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(vt[sp][dt][1]) : "v"(a), "i"(8 * G::ROW_BYTES));
        }
    }
    // ---- causal mask (diagonal sub-tile only: scalar branch); online softmax in the exp2 domain with the
    //      1/sqrt(d)*log2(e) scale folded into the exponent FMA: p = exp2(s*c - m*c), m tracked on the raw scores
    if (diag) {
      const int lim = r - 4 * h;   // reg t holds key (t&3) + 8*(t>>2) + 4h (relative): masked iff that exceeds r
#pragma unroll
      for (int t = 0; t < 16; ++t) s[t] = ((t & 3) + 8 * (t >> 2) > lim) ? -1e30f : s[t];
    }
    float mloc = fmaxf(fmaxf(s[0], s[1]), s[2]);
#pragma unroll
    for (int t = 3; t < 15; t += 2) mloc = fmaxf(fmaxf(mloc, s[t]), s[t + 1]);
    mloc = xhalf_max(fmaxf(mloc, s[15]));
    // Lazy reference maximum: m moves only when the row maximum has outgrown it by more than LAZY_T in the exponent (a factor
    // 2^LAZY_T on p); until then p = exp2((s - m) c) <= 2^LAZY_T stays far inside fp32 / bf16 range and alpha is EXACTLY 1, so the
    // rescaling of the O accumulators (64 multiplies per sub-tile at d = 128, a third of the softmax VALU work) is skipped for
    // the whole wave almost always after the first tile.  O / l is the same ratio whatever reference the exponentials use.
    const bool grow = (mloc - m) * scale_log2 > LAZY_T;
    const float mnew = grow ? mloc : m;
    const float alpha = grow ? __builtin_amdgcn_exp2f((m - mnew) * scale_log2) : 1.0f;
    m = mnew;
    const float mc = -mnew * scale_log2;
    {
      const f32x2 c2 = {scale_log2, scale_log2}, m2 = {mc, mc};
      f32x2 ps2 = {0.f, 0.f};
#pragma unroll
      for (int t = 0; t < 16; t += 2) {
        f32x2 e = f32x2{s[t], s[t + 1]} * c2 + m2;
        e[0] = __builtin_amdgcn_exp2f(e[0]);
        e[1] = __builtin_amdgcn_exp2f(e[1]);
        s[t] = e[0];
        s[t + 1] = e[1];
        ps2 += e;
      }
      l = l * alpha + (ps2[0] + ps2[1]);
    }
    if (!__all(alpha == 1.0f)) {  // wave-uniform: no q row of this wave raised its running max -> nothing to rescale (exact)
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int t = 0; t < 16; ++t) o[dt][t] *= alpha;
    }
    // ---- P^T -> bf16 B fragments: k-step sp uses regs 8sp .. 8sp+7
    bf16x8 pf[2];
#pragma unroll
    for (int sp = 0; sp < 2; ++sp)
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) pf[sp][jj] = attn_cvt(s[8 * sp + jj]);
    // ---- O^T += V^T P^T
    if (DT == 2) {
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(vt[0][0][0]), "+v"(vt[0][0][1]), "+v"(vt[0][1][0]), "+v"(vt[0][1][1]), "+v"(vt[1][0][0]), "+v"(vt[1][0][1]),
                     "+v"(vt[1][1][0]), "+v"(vt[1][1][1])
                   :
                   : "memory");
    } else {
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(vt[0][0][0]), "+v"(vt[0][0][1]), "+v"(vt[0][1][0]), "+v"(vt[0][1][1]), "+v"(vt[0][2 % DT][0]), "+v"(vt[0][2 % DT][1]),
                     "+v"(vt[0][3 % DT][0]), "+v"(vt[0][3 % DT][1]), "+v"(vt[1][0][0]), "+v"(vt[1][0][1]), "+v"(vt[1][1][0]), "+v"(vt[1][1][1]),
                     "+v"(vt[1][2 % DT][0]), "+v"(vt[1][2 % DT][1]), "+v"(vt[1][3 % DT][0]), "+v"(vt[1][3 % DT][1])
                   :
                   : "memory");
    }
#pragma unroll
    for (int sp = 0; sp < 2; ++sp)
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        union { struct { s16x4 a, b; } s; bf16x8 v; } u;
        u.s.a = vt[sp][dt][0]; u.s.b = vt[sp][dt][1];
        o[dt] = attn_mfma(u.v, pf[sp], o[dt]);
      }
  };

  const int nkt = qt + 1;
  for (int kt = 0; kt < nkt; ++kt) {
    if (kt == nkt - 1 && inext.pair >= 0) request_q(inext);     // the next item's Q block: requested a whole tile before it is used
    const char* sK = smem + cur * (2 * G::TILE_BYTES);
    const char* sV = sK + G::TILE_BYTES;
    // Software pipeline inside the wave: the QK^T MFMAs of BOTH 32-key sub-tiles are issued first (K fragments read in
    // one batch), so the second product runs on the matrix pipe while the VALU does the first sub-tile's softmax, and
    // the first P.V runs under the second softmax.
    const bool two = (kt * 64 + 32 <= q0);   // wave-uniform: second sub-tile not entirely above the diagonal
    const bool more = stage_next();                              // tile (this + 2) of the stream, whichever item it belongs to
    if (active) {
      f32x16 s0_ = qk_product(sK, qf);
      f32x16 s1_;
      if (two) s1_ = qk_product(sK + 32 * G::ROW_BYTES, qf);
      softmax_pv(s0_, sV, kt * 64 == q0);
      if (two) softmax_pv(s1_, sV + 32 * G::ROW_BYTES, kt * 64 + 32 == q0);
    }
    if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_TILE) : "memory");   // the next tile landed, the one after may stay in flight
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    cur = cur == NST - 1 ? 0 : cur + 1;
  }
  // the staging block now holds the next item's Q rows (older than the tile the counted wait just covered): fragments out first,
  // then the block is free for this item's O rows
  if (DW < NW && wave >= DW) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (see the prologue)
  if (inext.pair >= 0) read_q(qf);
  if (active) {
    float ltot;
    {
      auto rr = __builtin_amdgcn_permlane32_swap(__float_as_uint(l), __float_as_uint(l), false, false);
      ltot = __uint_as_float(rr[0]) + __uint_as_float(rr[1]);
    }
    const float inv = 1.0f / ltot;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the Q fragment reads are done with the block
    // O^T accumulators -> bf16 rows in the block: lane (r, h) owns row r, 4 consecutive columns dt*32 + 8*g + 4h; 16-B chunk index
    // XOR (row mod chunks-per-row) keeps both the 8-byte writes and the 16-byte row reads off each other's banks
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int g4_ = 0; g4_ < 4; ++g4_) {
        bf16x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = o16(o[dt][4 * g4_ + e] * inv, out_f16);
        *(bf16x4*)(sW + r * G::ROW_BYTES + ((((dt * 4 + g4_) ^ (r & (G::CH - 1))) << 4) | (h << 3))) = v;
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // same wave wrote, same wave reads: no barrier
    // Read-out: all QINST row segments into registers first, then QINST buffer stores through the sequence's descriptor (rows >= len fall
    // outside num_records and are dropped: no branch per store).  The LDS addresses are recomputed from the lane id HERE (the empty asm
    // keeps them from being hoisted out of the item loop): hoisted, four of them were spilled, and every scratch reload came with a
    // vmcnt(0) that waited for the store issued just before it -- four serial store round trips, 6 us per item
    // (tools/exp/attn_trace_tiled.py: 35 % of the launch at S = 512).
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(out + (int64_t)s0 * ((int64_t)nq * D)), 0, len * nq * (D * 2), 0x00020000);
    u32x4 ov[QINST];
#pragma unroll
    for (int j = 0; j < QINST; ++j) {
      const int s_ = j * 64 + ln;
      const int row = s_ / G::CH, ch = s_ % G::CH;
      ov[j] = *(const u32x4*)(sW + row * G::ROW_BYTES + ((ch ^ (row & (G::CH - 1))) << 4));
    }
#pragma unroll
    for (int j = 0; j < QINST; ++j) {
      const int s_ = j * 64 + ln;
      const int row = s_ / G::CH, ch = s_ % G::CH;
      __builtin_amdgcn_raw_buffer_store_b128(ov[j], orsrc, ((q0 + row) * nq + hq) * (D * 2) + ch * 16, 0, 0);
    }
  }
  ic = inext;
  }  // items
}

// ---------------------------------------------------------------------------------------------------------------
// Round 5: the same tile arithmetic on a PREBUILT work list, with buffer addressing (lrx_attn_varlen_causal_items; 1.10 -> 1.00 ms at
// 256 x 512 tokens, 32 / 8 heads, d = 128, bit-identical output: profiles/r05_attn_rework.txt).
//
// tools/exp/attn_trace_tiled.py on the kernel above: a 64-key tile step = 0.58 us of REQUESTING the next tile + 1.6 compute + 0.75
// barrier + 0.28 loop, and an item paid another 3.7 us around its steps -- 1.5 of them in the item walker (runtime integer divisions for
// (sequence, kv head, part, q tile), two dependent scalar loads of cu[], once per walker) and the per-lane 64-bit source addresses of
// 4 + 8 LDS-DMA instructions (row clamp, multiply by the row stride, swizzle): VALU work that the two waves of a SIMD pay for one after
// the other.  PMC (tools/pmc_attn.sh): 202 VALU instructions per 32 x 32 sub-tile against 16 MFMAs, VALU issue 50 % + MFMA 32 % of the
// SIMD cycles at an effective 1.65 GHz -- the kernel is bound by what it issues, so this one issues less:
//   * the items of every workgroup are written once per (cu_seqlens, geometry) by k_attn_build_items as 16-byte records; the stream and
//     the compute side fetch them with one s_load_dwordx4 each, one item ahead of use;
//   * K, V and Q rows are fetched with buffer_load_dwordx4 ... lds through a descriptor whose base is the tile's first row and whose
//     num_records ends at the sequence's last row: rows past the end are dropped by the range check (no clamp), the per-lane offsets
//     (row * stride + swizzled chunk) are kernel constants, the kv-head column goes into the scalar offset: a tile request is a
//     descriptor update on the SALU plus the load instructions;
//   * the LDS addresses of the 24 fragment reads of a sub-tile are kernel constants plus the ring-stage base, sub-tile / half / k-step
//     offsets ride in the instructions' immediate fields (48 of the 202 were v_xor + v_add pairs in front of those reads);
//   * the output addresses are recomputed from the hardware lane counter inside the item epilogue: nothing lane-derived is live across
//     the tile loop, hipcc spills nothing (a reload there waits, with its vmcnt(0), for the tile requests in flight).
// LDS is zero-filled once per workgroup: a dropped row leaves its ring bytes alone, and P = 0 times a stale NaN would poison O.
// ---------------------------------------------------------------------------------------------------------------
// (a free __device__ function: called from a lambda of the kernel, the host pass of hipcc drops the whole kernel stub without a diagnostic)
__device__ __forceinline__ void attn_buf_load_lds16(__amdgpu_buffer_rsrc_t rs, lptr_t dst, int voff, int soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, dst, 16, voff, soff, 0, 0);
}
typedef __attribute__((ext_vector_type(4))) int i32x4;   // one item: {first token, length, q tile | part << 16, kv head}; length 0 ends a workgroup's list
// The work list: int32 list_start[n_wg + 1] (padded to a multiple of 4 entries), then the workgroups' item lists back to back, each ended by a
// zero item.  Only items that exist are stored (q tiles past a sequence's end are dropped here), so the list is bounded by
// (total_tokens / 64 + n_seqs) x kv heads x parts + one end marker per workgroup -- known from the batch size alone.
// One block, one thread per workgroup: count its items (the walker of k_attn_varlen_causal), scan, walk again and store.

__global__ void __launch_bounds__(1024)
k_attn_build_items(const int32_t* __restrict__ cu, int32_t* __restrict__ list_start, i32x4* __restrict__ items, int cap_items, int n_wg, int nqt, int nkv,
                   int nparts, int n_seqs, int n_items, int gs, int last_tile_only, int* __restrict__ overflow) {
  __shared__ int s_scan[1024];
  const int b = threadIdx.x;
  const int ny = nkv * nparts, n_pairs = n_seqs * ny;
  const int xcd = b & 7, slot = b >> 3;
  const int g4 = gs > 0 ? slot / gs : 0, jslot = gs > 0 ? slot % gs : 0, gpx = gs > 0 ? (n_wg >> 3) / gs : 1;
  auto walk = [&](i32x4* dst) -> int {          // dst == nullptr: count only
    int n = 0;
    auto emit = [&](int pair, int qt_sel) {
      const int sq = pair / ny;
      const int s0 = cu[sq], len = cu[sq + 1] - s0;
      const int qt = last_tile_only ? ((len - 1) >> 6) : qt_sel;      // last_tile_only: the q tile holding the sequence's last token
      if (len <= 0 || qt * 64 >= len) return;
      if (dst) {
        const int yy = pair - sq * ny, hk = yy / nparts, part = yy - hk * nparts;
        dst[n] = i32x4{s0, len, qt | (part << 16), hk};
      }
      ++n;
    };
    if (gs > 0) {
      // (the order is explained at k_attn_varlen_causal: the q tiles of one (sequence, kv head) on `gs` workgroups of one XCD; slot j of a
      // group takes q tiles j, j + gs, ... from the long end on even steps and from the short end on odd ones)
      for (int step = 0;; ++step) {
        const int v = step * gpx + g4;
        if ((v / nparts) * 8 >= n_pairs / nparts) break;
        const int pair = ((v / nparts) * 8 + xcd) * nparts + v % nparts;
        if (pair >= n_pairs) continue;
        for (int idx = jslot; idx < nqt; idx += gs) emit(pair, (step & 1) ? idx : nqt - 1 - idx);
      }
    } else {
      for (int item = b; item < n_items; item += n_wg)
        emit(last_tile_only ? item : item % n_pairs, last_tile_only ? 0 : nqt - 1 - item / n_pairs);
    }
    return n;
  };
  const int mine = b < n_wg ? walk(nullptr) + 1 : 0;            // + the end marker
  s_scan[b] = mine;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {                          // inclusive scan (n_wg <= 1024)
    const int v = b >= o ? s_scan[b - o] : 0;
    __syncthreads();
    s_scan[b] += v;
    __syncthreads();
  }
  if (b >= n_wg) return;
  const int start = s_scan[b] - mine;
  if (s_scan[n_wg - 1] + 2 > cap_items) {                        // (cannot happen: lrx_attn_items_bytes bounds the same walk)
    // every workgroup's list becomes the empty list at slot 0 (the launch then reads nothing out of range and computes nothing); the
    // counter is part of lrx_device_error_count, which the encoders' callers check: loud, not silently wrong rows
    if (b == 0) {
      atomicAdd(overflow, 1);
      for (int i = 0; i < 3 && i < cap_items; ++i) items[i] = i32x4{0, 0, 0, 0};
    }
    list_start[b] = 0;
    if (b == n_wg - 1) list_start[n_wg] = 0;
    return;
  }
  list_start[b] = start;
  if (b == n_wg - 1) list_start[n_wg] = s_scan[b];
  const int n = walk(items + start);
  items[start + n] = i32x4{0, 0, 0, 0};
  if (b == n_wg - 1) { items[start + n + 1] = i32x4{0, 0, 0, 0}; items[start + n + 2] = i32x4{0, 0, 0, 0}; }   // padding behind the last end marker
}

template <int D, int GRP>
__global__ void __launch_bounds__(128 * GRP)
k_attn_stream(const __bf16* __restrict__ qkv, const int32_t* __restrict__ list_start, const i32x4* __restrict__ items, int nq, int nkv,
              __bf16* __restrict__ out, float scale_log2, int out_f16) {
  using G = AttnGeom<D>;
  constexpr int NW = 2 * GRP;
  constexpr int KS = D / 16, DT = D / 32;
  constexpr int DW = attn_dma_waves(G::INSTS, NW);
  constexpr int NST = 3;
  constexpr int PER_TILE = 2 * (G::INSTS / DW);
  constexpr int QO_BYTES = 32 * G::ROW_BYTES;
  constexpr int QINST = QO_BYTES / 1024;
  constexpr int SMEM = NST * 2 * G::TILE_BYTES + NW * QO_BYTES;
  __shared__ __attribute__((aligned(1024))) char smem[SMEM];  // [stage][K|V] | [wave] Q/O block

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int RSB = (nq + 2 * nkv) * (D * 2);          // bytes per token row of q|k|v
  const int grp_total = nq / nkv;
  const i32x4* const lst = items + list_start[blockIdx.x];

  i32x4 ic = lst[0];
  if (ic[1] == 0) return;
  for (int i = tid * 16; i < SMEM; i += 128 * GRP * 16) *(u32x4*)(smem + i) = u32x4{0u, 0u, 0u, 0u};
  __syncthreads();

  // ---- per-lane source offsets of the tile rows (instruction j = wave + jj * DW fills LDS bytes [j*1024, j*1024 + 1024): slot s = j*64 + lane,
  //      row s / CH, chunk position s % CH holding logical chunk cs ^ x(row)) and of the first Q instruction
  int voffK[G::INSTS / DW], voffV[G::INSTS / DW];
#pragma unroll
  for (int jj = 0; jj < G::INSTS / DW; ++jj) {
    const int s = (wave + jj * DW) * 64 + lane;
    const int row = s / G::CH, cs = s % G::CH;
    voffK[jj] = row * RSB + ((cs ^ G::xk(row)) << 4);
    voffV[jj] = row * RSB + ((cs ^ G::xv(row)) << 4);
  }
  const int vq0 = (lane / G::CH) * RSB + (((lane % G::CH) ^ G::xk(lane / G::CH)) << 4);

  auto stage = [&](int st, const i32x4& it, int kt) {
    if (wave >= DW) return;
    char* sK = smem + st * (2 * G::TILE_BYTES);
    char* sV = sK + G::TILE_BYTES;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)qkv + (int64_t)(it[0] + kt * 64) * RSB), 0, (it[1] - kt * 64) * RSB, 0x00020000);
    const int soK = (nq + it[3]) * (D * 2), soV = soK + nkv * (D * 2);
#pragma unroll
    for (int jj = 0; jj < G::INSTS / DW; ++jj) {
      const int j = wave + jj * DW;
      attn_buf_load_lds16(rs, (lptr_t)(sK + j * 1024), voffK[jj], soK);
      attn_buf_load_lds16(rs, (lptr_t)(sV + j * 1024), voffV[jj], soV);
    }
  };
  // prefetch side of the stream: its own cursor over the list, the record after the current one already loaded
  // prefetch side of the stream: its own index into the list, the item after the current one already loaded (a load past the list's
  // end marker reads the next workgroup's first item or the builder's padding: never used)
  int pi = 0;
  i32x4 ip = ic, ipn = lst[1];
  int ktp = 0, sp_ = 0;
  auto stage_next = [&]() -> bool {
    if (ip[1] == 0) return false;
    stage(sp_, ip, ktp);
    sp_ = sp_ == NST - 1 ? 0 : sp_ + 1;
    if (++ktp > (ip[2] & 0xffff)) { ip = ipn; ++pi; ipn = lst[pi + 1]; ktp = 0; }
    return true;
  };

  const int koff0 = r * G::ROW_BYTES + ((h ^ G::xk(r)) << 4);
  int voff0;
  {
    const int g = lane >> 4, i = lane & 15, qd = i >> 2, p = i & 3;
    voff0 = qd * G::ROW_BYTES + (((2 * (g & 1) + (p >> 1)) ^ G::xv(qd)) << 4) + 8 * (p & 1);
  }

  // Q block of an item for this wave (rows q0 .. q0 + 31 of its head, K-tile swizzle): instruction j covers rows j*RPI .. + RPI - 1, and
  // x(j*RPI + row) = x(j*RPI) ^ x(row) for both geometries (the two terms use disjoint bits), so its offsets are vq0 ^ const + const
  char* const sW = smem + NST * 2 * G::TILE_BYTES + wave * QO_BYTES;
  auto request_q = [&](const i32x4& it) {
    constexpr int RPI = G::ROWS_PER_INST;
    const int qt_ = it[2] & 0xffff, part_ = it[2] >> 16;
    const int hig = part_ * GRP + (wave % GRP);
    const int q0_ = qt_ * 64 + (wave / GRP) * 32;
    if (q0_ >= it[1] || hig >= grp_total) return;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)qkv + (int64_t)(it[0] + q0_) * RSB), 0, (it[1] - q0_) * RSB, 0x00020000);
    const int so = (it[3] * grp_total + hig) * (D * 2);
#pragma unroll
    for (int j = 0; j < QINST; ++j)
      attn_buf_load_lds16(rs, (lptr_t)(sW + j * 1024), (vq0 ^ (G::xk(j * RPI) << 4)) + j * RPI * RSB, so);
  };
  auto read_q = [&](bf16x8 (&dst)[KS]) {
    int kb = koff0;
    asm volatile("" : "+v"(kb));
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) dst[ks] = *(const bf16x8*)(sW + (kb ^ (ks << 5)));
  };

  int ci = 0;
  bf16x8 qf[KS];
  request_q(ic);
  const bool t0 = stage_next(), t1 = stage_next();
  (void)t0;
  if (t1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_TILE) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (DW < NW && wave >= DW) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  read_q(qf);
  int cur = 0;

  while (ic[1] != 0) {
  ++ci;
  const i32x4 inext = lst[ci];
  const int s0 = ic[0], len = ic[1], qt = ic[2] & 0xffff, part = ic[2] >> 16, hk = ic[3];
  const int qtile0 = qt * 64;
  const int head_in_grp = part * GRP + (wave % GRP);
  const int hq = hk * grp_total + min(head_in_grp, grp_total - 1);
  const int q0 = qtile0 + (wave / GRP) * 32;
  const bool active = q0 < len && head_in_grp < grp_total;

  f32x16 o[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int t = 0; t < 16; ++t) o[dt][t] = 0.f;
  float m = -1e30f, l = 0.f;

  // LDS addresses of the fragment reads: the per-lane offsets (swizzle included) are kernel constants, a tile adds its ring-stage base
  // ONCE (set_tile_base), sub-tile, half and k-step-pair offsets ride in the instructions' immediate fields.  (PMC, tools/pmc_attn.sh: 202
  // VALU instructions per sub-tile against 16 MFMAs -- VALU issue 50 % + MFMA 32 % of the SIMD cycles, and 48 of the 202 were the
  // v_xor + v_add pairs in front of the 24 fragment reads.)
  int kofs[KS], vofs[DT];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) kofs[ks] = koff0 ^ (ks << 5);
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) vofs[dt] = 4 * h * G::ROW_BYTES + (voff0 ^ (dt << 6));
  uint32_t ka[KS], va[DT];     // this tile's addresses
  auto set_tile_base = [&](int st) {
    const uint32_t kb_ = (uint32_t)(uintptr_t)(lds_char_ptr)(smem + st * (2 * G::TILE_BYTES));
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) ka[ks] = kb_ + kofs[ks];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) va[dt] = kb_ + G::TILE_BYTES + vofs[dt];
  };
  auto qk_product = [&](int u) -> f32x16 {
    bf16x8 kf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) kf[ks] = *(const bf16x8*)((lds_char_ptr)(uintptr_t)ka[ks] + u * 32 * G::ROW_BYTES);
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // inline-constant C operand
    f32x16 acc = attn_mfma(kf[0], qf[0], zero);
#pragma unroll
    for (int ks = 1; ks < KS; ++ks) acc = attn_mfma(kf[ks], qf[ks], acc);
    return acc;
  };
  auto xhalf_max = [](float x) -> float {
    auto rr = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(rr[0]), __uint_as_float(rr[1]));
  };
  // (the sub-tile arithmetic below is the one documented at k_attn_varlen_causal: V^T fragments by inline-asm transposing reads, lazy
  // reference maximum, exp2 with the scale folded in, P^T as the B operand of the second product)
  struct VFrag { s16x4 v[2][DT][2]; };
  auto read_v = [&](int u, VFrag& f) {
#pragma unroll
    for (int sp = 0; sp < 2; ++sp)
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        if (u == 0) {
          asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f.v[sp][dt][0]) : "v"(va[dt]), "i"(sp * 16 * G::ROW_BYTES));
          asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f.v[sp][dt][1]) : "v"(va[dt]), "i"(sp * 16 * G::ROW_BYTES + 8 * G::ROW_BYTES));
        } else {
          asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f.v[sp][dt][0]) : "v"(va[dt]), "i"(32 * G::ROW_BYTES + sp * 16 * G::ROW_BYTES));
          asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f.v[sp][dt][1]) : "v"(va[dt]), "i"(32 * G::ROW_BYTES + sp * 16 * G::ROW_BYTES + 8 * G::ROW_BYTES));
        }
      }
  };
  auto wait_v = [&](VFrag& f) {
    if constexpr (DT == 2) {
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(f.v[0][0][0]), "+v"(f.v[0][0][1]), "+v"(f.v[0][1][0]), "+v"(f.v[0][1][1]), "+v"(f.v[1][0][0]), "+v"(f.v[1][0][1]),
                     "+v"(f.v[1][1][0]), "+v"(f.v[1][1][1])
                   :
                   : "memory");
    } else {
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(f.v[0][0][0]), "+v"(f.v[0][0][1]), "+v"(f.v[0][1][0]), "+v"(f.v[0][1][1]), "+v"(f.v[0][2 % DT][0]), "+v"(f.v[0][2 % DT][1]),
                     "+v"(f.v[0][3 % DT][0]), "+v"(f.v[0][3 % DT][1]), "+v"(f.v[1][0][0]), "+v"(f.v[1][0][1]), "+v"(f.v[1][1][0]), "+v"(f.v[1][1][1]),
                     "+v"(f.v[1][2 % DT][0]), "+v"(f.v[1][2 % DT][1]), "+v"(f.v[1][3 % DT][0]), "+v"(f.v[1][3 % DT][1])
                   :
                   : "memory");
    }
  };
  auto pv = [&](const VFrag& f, const bf16x8 (&pf)[2]) {
#pragma unroll
    for (int sp = 0; sp < 2; ++sp)
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        union { struct { s16x4 a, b; } s; bf16x8 v; } u;
        u.s.a = f.v[sp][dt][0]; u.s.b = f.v[sp][dt][1];
        o[dt] = attn_mfma(u.v, pf[sp], o[dt]);
      }
  };
  auto rescale = [&](float alpha) {
    if (!__all(alpha == 1.0f)) {  // wave-uniform: no q row of this wave raised its running max -> nothing to rescale (exact)
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int t = 0; t < 16; ++t) o[dt][t] *= alpha;
    }
  };
  // scores -> P^T fragments; returns the factor the O accumulators are due BEFORE this sub-tile's P.V is added
  auto mask_diag = [&](f32x16& s) {          // causal mask of a sub-tile on the diagonal: reg t holds key (t&3) + 8*(t>>2) + 4h (relative), masked iff that exceeds r
    const int lim = r - 4 * h;
#pragma unroll
    for (int t = 0; t < 16; ++t) s[t] = ((t & 3) + 8 * (t >> 2) > lim) ? -1e30f : s[t];
  };
  auto softmax = [&](f32x16& s, bf16x8 (&pf)[2]) -> float {
    float mloc = fmaxf(fmaxf(s[0], s[1]), s[2]);
#pragma unroll
    for (int t = 3; t < 15; t += 2) mloc = fmaxf(fmaxf(mloc, s[t]), s[t + 1]);
    mloc = xhalf_max(fmaxf(mloc, s[15]));
    const bool grow = (mloc - m) * scale_log2 > LAZY_T;
    const float mnew = grow ? mloc : m;
    const float alpha = grow ? __builtin_amdgcn_exp2f((m - mnew) * scale_log2) : 1.0f;
    m = mnew;
    const float mc = -mnew * scale_log2;
    {
      const f32x2 c2 = {scale_log2, scale_log2}, m2 = {mc, mc};
      f32x2 ps2 = {0.f, 0.f};
#pragma unroll
      for (int t = 0; t < 16; t += 2) {
        f32x2 e = f32x2{s[t], s[t + 1]} * c2 + m2;
        e[0] = __builtin_amdgcn_exp2f(e[0]);
        e[1] = __builtin_amdgcn_exp2f(e[1]);
        s[t] = e[0];
        s[t + 1] = e[1];
        ps2 += e;
      }
      l = l * alpha + (ps2[0] + ps2[1]);
    }
#pragma unroll
    for (int sp = 0; sp < 2; ++sp)
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) pf[sp][jj] = attn_cvt(s[8 * sp + jj]);
    return alpha;
  };

  const int nkt = qt + 1;
  for (int kt = 0; kt < nkt; ++kt) {
    if (kt == nkt - 1 && inext[1] != 0) request_q(inext);
    const bool two = (kt * 64 + 32 <= q0);   // wave-uniform: second sub-tile not entirely above the diagonal
    const bool more = stage_next();
    if (active) {
      // Both QK^T products first, each sub-tile's V^T fragments requested before its softmax.  Measured on this kernel and not kept
      // (tools/exp/README.md, round 5; 0.978 ms as it stands): the two row halves of a head -- the two waves of a SIMD -- walking the
      // sub-tiles in different orders so that one's LDS phases meet the other's VALU phases (+4 %); all eight K fragments in registers before
      // the first MFMA (+2 %); one branch-poor copy of this code in which hipcc does put the first softmax between the MFMAs of the second
      // QK^T and the second softmax between those of the first P.V (+1.5 %); 64 q rows per wave, one wave per SIMD, O in the accumulation
      // registers (+25 %).  PMC: VALU issue 50 % + MFMA 32 % of the SIMD cycles at an effective 1.65 GHz -- what pays is fewer instructions.
      set_tile_base(cur);
      VFrag vf;
      bf16x8 pf[2];
      f32x16 s0_ = qk_product(0);
      f32x16 s1_;
      if (two) s1_ = qk_product(1);
      read_v(0, vf);
      if (kt * 64 == q0) mask_diag(s0_);
      rescale(softmax(s0_, pf));
      wait_v(vf);
      pv(vf, pf);
      if (two) {
        read_v(1, vf);
        if (kt * 64 + 32 == q0) mask_diag(s1_);
        rescale(softmax(s1_, pf));
        wait_v(vf);
        pv(vf, pf);
      }
    }
    if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_TILE) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    cur = cur == NST - 1 ? 0 : cur + 1;
  }
  if (DW < NW && wave >= DW) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (inext[1] != 0) read_q(qf);
  if (active) {
    float ltot;
    {
      auto rr = __builtin_amdgcn_permlane32_swap(__float_as_uint(l), __float_as_uint(l), false, false);
      ltot = __uint_as_float(rr[0]) + __uint_as_float(rr[1]);
    }
    const float inv = 1.0f / ltot;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // the lane id from the hardware counter, behind a volatile zero: every address below is recomputed HERE.  Derived from threadIdx they
    // are values live across the whole kernel, hipcc (at its 256 VGPRs) spills some, and a reload here comes with a vmcnt(0) that waits
    // for the tile requests in flight
    int zero_;
    asm volatile("v_mov_b32 %0, 0" : "=v"(zero_));
    const unsigned ln = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, zero_));
    const unsigned r_ = ln & 31, h_ = ln >> 5;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int g4_ = 0; g4_ < 4; ++g4_) {
        bf16x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = o16(o[dt][4 * g4_ + e] * inv, out_f16);
        *(bf16x4*)(sW + r_ * G::ROW_BYTES + ((((dt * 4 + g4_) ^ (r_ & (G::CH - 1))) << 4) | (h_ << 3))) = v;
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(out + (int64_t)s0 * ((int64_t)nq * D)), 0, len * nq * (D * 2), 0x00020000);
    u32x4 ov[QINST];
#pragma unroll
    for (int j = 0; j < QINST; ++j) {
      const unsigned s_ = j * 64 + ln;
      const unsigned row = s_ / G::CH, ch = s_ % G::CH;
      ov[j] = *(const u32x4*)(sW + row * G::ROW_BYTES + ((ch ^ (row & (G::CH - 1))) << 4));
    }
#pragma unroll
    for (int j = 0; j < QINST; ++j) {
      const unsigned s_ = j * 64 + ln;
      const unsigned row = s_ / G::CH, ch = s_ % G::CH;
      __builtin_amdgcn_raw_buffer_store_b128(ov[j], orsrc, ((q0 + row) * nq + hq) * (D * 2) + ch * 16, 0, 0);
    }
  }
  ic = inext;
  }  // items
}

// ---------------------------------------------------------------------------------------------------------------
// d = 64, S <= 512: K/V-resident kernel.  One workgroup = (sequence, kv head): the whole K and V of that kv head
// (<= 512 keys x 128 B, 64 KiB each) is staged ONCE into LDS (the tiled kernel above re-loads every K/V tile for each of
// the q tiles that needs it: 4.5x the bytes at S = 512, which saturates the per-CU load path), then 16 waves work through
// the (q head of the group, 32-row q block) tasks with no barrier in the main loop.  Tasks are handed out from an LDS
// counter, LARGEST FIRST (q block nsub-1 of every head, then nsub-2, ...): a wave that finishes takes the next one, so
// the pair ends on 1-sub-tile tasks.  (Round 3: the static assignment -- wave w -> head w % grp, blocks g, 2NG-1-g, ... --
// gave every wave the same number of sub-tiles but not the same time: with 4 waves per SIMD sharing its VALU a wave's
// pace depends on its neighbours, and the waves of a pair finished up to 40 % apart; a lone wave runs at ~45 % of what
// four do together.  tools/exp/attn_trace.py: 17 % of a pair was spent waiting at its final barrier.)  Any group size.
// Inside a task the QK^T product of sub-tile u+1 is issued before the softmax of sub-tile u.
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024)
k_attn_resident64(const __bf16* __restrict__ qkv, const int32_t* __restrict__ cu, int nq, int nkv, __bf16* __restrict__ out,
                  float scale_log2, int n_pairs, int out_f16) {
  constexpr int D = 64;
  using G = AttnGeom<D>;
  constexpr int KS = D / 16, DT = D / 32;
  constexpr int MAXK = 512;
  __shared__ __attribute__((aligned(1024))) char smem[2 * MAXK * G::ROW_BYTES];  // K [512][128 B] | V [512][128 B]
  char* const sKb = smem;
  char* const sVb = smem + MAXK * G::ROW_BYTES;
  __shared__ int s_next_task;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int64_t RS = (int64_t)(nq + 2 * nkv) * D;
  const int grp = nq / nkv;
  // persistent workgroups (one per CU: 128 KiB of LDS) walk the (sequence, kv head) pairs: a workgroup's successor used to start
  // ~10 us after its exit (see k_attn_varlen_causal)
  for (int pair = blockIdx.x; pair < n_pairs; pair += gridDim.x) {
  const int b = pair / nkv, hk = pair - b * nkv;
  const int s0 = cu[b], len = cu[b + 1] - s0;
  if (len <= 0) continue;
  const __bf16* kbase = qkv + (int64_t)s0 * RS + (int64_t)(nq + hk) * D;
  const __bf16* vbase = kbase + (int64_t)nkv * D;
  const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(out + (int64_t)s0 * ((int64_t)nq * D)), 0, len * nq * (D * 2), 0x00020000);

  // ---- stage all K/V rows of this kv head (rows >= len are clamped copies; the causal mask hides them)
  const int ninst = ((len + 63) >> 6) * G::INSTS;   // 1-KiB LDS-DMA instructions per operand
  for (int j = wave; j < ninst; j += 16) {
    int s = j * 64 + lane;
    int row = s / G::CH, cs = s % G::CH;
    int grow = min(row, len - 1);
    const __bf16* kp = kbase + (int64_t)grow * RS + ((cs ^ G::xk(row)) << 3);
    const __bf16* vp = vbase + (int64_t)grow * RS + ((cs ^ G::xv(row)) << 3);
    __builtin_amdgcn_global_load_lds((gptr_t)kp, (lptr_t)(sKb + j * 1024), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gptr_t)vp, (lptr_t)(sVb + j * 1024), 16, 0, 0);
  }

  // lane constants of the fragment reads: ONE register each, the other k-steps / d tiles by XOR (the swizzles are XORs of the 16-B chunk
  // index, so chunk (2 ks + h) ^ x = (h ^ x) ^ (2 ks)) -- the arrays cost registers this 128-VGPR kernel spills otherwise, and every
  // scratch reload is followed by a vmcnt(0) that also waits for the previous task's output stores
  const int koff0 = r * G::ROW_BYTES + ((h ^ G::xk(r)) << 4);
  int voff0;
  {
    const int g4 = lane >> 4, i4 = lane & 15, qd = i4 >> 2, p4 = i4 & 3;
    voff0 = qd * G::ROW_BYTES + (((2 * (g4 & 1) + (p4 >> 1)) ^ G::xv(qd)) << 4) + 8 * (p4 & 1);
  }
  const int nsub = (len + 31) >> 5;
  const int n_tasks = nsub * grp;
  // task t = (q block nsub - 1 - t / grp, head t % grp): the first 16 are the waves' own, the rest come from the counter (reset here: every
  // wave left the previous pair's task loop before the barrier that ends it)
  if (tid == 0) s_next_task = 16;
  bf16x8 qf[KS];
  auto load_q = [&](const int t) {
    const int qrow = min((nsub - 1 - t / grp) * 32 + r, len - 1);
    const __bf16* qp = qkv + ((int64_t)s0 + qrow) * RS + (int64_t)(hk * grp + t % grp) * D + h * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = *(const bf16x8*)(qp + ks * 16);
  };
  // The same loads as inline asm, for the prefetch of the NEXT task's fragments: hipcc's wait insertion does not see them.  With counted
  // loads it guards every later product of the task with vmcnt(3..0) (any path on which qf might still be in flight), and those waits also
  // cover the previous task's output stores: ~2 us of store acknowledgement exposed per task (0.07 ms of a 0.61-ms launch).  The one wait
  // these loads need is the explicit vmcnt(4) at the top of a task: the four loads are older than the task's four (always issued) stores.
  auto prefetch_q = [&](const int t) {
    const int qrow = min((nsub - 1 - t / grp) * 32 + r, len - 1);
    const __bf16* qp = qkv + ((int64_t)s0 + qrow) * RS + (int64_t)(hk * grp + t % grp) * D + h * 8;
    asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:32\n\t"
                 "global_load_dwordx4 %2, %4, off offset:64\n\tglobal_load_dwordx4 %3, %4, off offset:96"
                 : "=&v"(qf[0]), "=&v"(qf[1]), "=&v"(qf[2]), "=&v"(qf[3])
                 : "v"(qp)
                 : "memory");
  };
  static_assert(KS == 4, "prefetch_q issues four 16-byte loads per lane");
  int t = wave;
  if (t < n_tasks) load_q(t);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // K/V staged (and the first task's q fragments here)
  __syncthreads();

  while (t < n_tasks) {
    int tn = 0;
    if (lane == 0) tn = atomicAdd(&s_next_task, 1);   // LDS; the next task is known before this one starts: its q fragments are prefetched below
    tn = __builtin_amdgcn_readfirstlane(tn);
    const bool have_next = tn < n_tasks;
    const int i = nsub - 1 - t / grp;                 // this task's q block (wave-uniform)
    const int hq = hk * grp + t % grp;
    const int q0 = i * 32;
    // issue priority by task size: the SIMD's arbiter serves its OLDEST ready wave first, so with four VALU-bound waves the youngest one
    // crawls (tools/exp/attn_trace.py: 3.0 us per sub-tile against 0.6 for its neighbours) and the pair ended on that wave's first, large
    // task.  A large task now outranks the smaller ones the other waves have moved on to.
    if (i >= 12) __builtin_amdgcn_s_setprio(3);
    else if (i >= 8) __builtin_amdgcn_s_setprio(2);
    else if (i >= 4) __builtin_amdgcn_s_setprio(1);
    else __builtin_amdgcn_s_setprio(0);
    f32x16 o[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int t = 0; t < 16; ++t) o[dt][t] = 0.f;
    float m = -1e30f, l = 0.f;

    auto qk_product = [&](const char* kt_base) -> f32x16 {
      bf16x8 kf[KS];
      int kb = koff0;
      asm volatile("" : "+v"(kb));            // (keeps the three XORed copies from being hoisted to kernel entry and spilled)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) kf[ks] = *(const bf16x8*)(kt_base + (kb ^ (ks << 5)));
      const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      f32x16 acc = attn_mfma(kf[0], qf[0], zero);
#pragma unroll
      for (int ks = 1; ks < KS; ++ks) acc = attn_mfma(kf[ks], qf[ks], acc);
      return acc;
    };

    auto softmax_pv = [&](f32x16& sc, const int u) {
      if (u == i) {   // diagonal sub-tile
        const int lim = r - 4 * h;
#pragma unroll
        for (int t = 0; t < 16; ++t) sc[t] = ((t & 3) + 8 * (t >> 2) > lim) ? -1e30f : sc[t];
      }
      float mloc = fmaxf(fmaxf(sc[0], sc[1]), sc[2]);
#pragma unroll
      for (int t = 3; t < 15; t += 2) mloc = fmaxf(fmaxf(mloc, sc[t]), sc[t + 1]);
      mloc = fmaxf(mloc, sc[15]);
      {
        auto rr = __builtin_amdgcn_permlane32_swap(__float_as_uint(mloc), __float_as_uint(mloc), false, false);
        mloc = fmaxf(__uint_as_float(rr[0]), __uint_as_float(rr[1]));
      }
      const bool grow = (mloc - m) * scale_log2 > LAZY_T;     // lazy reference maximum: see k_attn_varlen_causal
      const float mnew = grow ? mloc : m;
      const float alpha = grow ? __builtin_amdgcn_exp2f((m - mnew) * scale_log2) : 1.0f;
      m = mnew;
      const float mc = -mnew * scale_log2;
      {
        // packed fp32 math (v_pk_fma_f32 / v_pk_add_f32: two scores per VALU slot); exp2 itself is scalar-per-lane.  (Round 3: the same
        // with single-issue v_fma / v_add forced by inline asm -- the microarchitecture guide prices a packed op beside MFMAs above the
        // two it replaces -- measured 0.618-0.622 against 0.604 ms here: with four waves per SIMD the fewer issue slots win.)
        const f32x2 c2 = {scale_log2, scale_log2}, m2 = {mc, mc};
        f32x2 ps2 = {0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 16; t += 2) {
          f32x2 e = f32x2{sc[t], sc[t + 1]} * c2 + m2;
          e[0] = __builtin_amdgcn_exp2f(e[0]);
          e[1] = __builtin_amdgcn_exp2f(e[1]);
          sc[t] = e[0];
          sc[t + 1] = e[1];
          ps2 += e;
        }
        l = l * alpha + (ps2[0] + ps2[1]);
      }
      // ---- V^T fragments (inline asm: see the tiled kernel); requested only now: 16 registers that are not live under the exponentials
      s16x4 vt[2][DT][2];
      {
        const uint32_t vb = (uint32_t)(uintptr_t)(lds_char_ptr)(sVb + (u * 32 + 4 * h) * G::ROW_BYTES);
        int vo = voff0;
        asm volatile("" : "+v"(vo));
#pragma unroll
        for (int sp = 0; sp < 2; ++sp)
#pragma unroll
          for (int dt = 0; dt < DT; ++dt) {
            const uint32_t a = vb + sp * 16 * G::ROW_BYTES + (vo ^ (dt << 6));
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(vt[sp][dt][0]) : "v"(a));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(vt[sp][dt][1]) : "v"(a), "i"(8 * G::ROW_BYTES));
          }
      }
      if (!__all(alpha == 1.0f)) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
          for (int t = 0; t < 16; ++t) o[dt][t] *= alpha;
      }
      bf16x8 pf[2];
#pragma unroll
      for (int sp = 0; sp < 2; ++sp)
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) pf[sp][jj] = attn_cvt(sc[8 * sp + jj]);
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(vt[0][0][0]), "+v"(vt[0][0][1]), "+v"(vt[0][1][0]), "+v"(vt[0][1][1]), "+v"(vt[1][0][0]), "+v"(vt[1][0][1]),
                     "+v"(vt[1][1][0]), "+v"(vt[1][1][1])
                   :
                   : "memory");
#pragma unroll
      for (int sp = 0; sp < 2; ++sp)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          union { struct { s16x4 a, b; } s; bf16x8 v; } uu;
          uu.s.a = vt[sp][dt][0]; uu.s.b = vt[sp][dt][1];
          o[dt] = attn_mfma(uu.v, pf[sp], o[dt]);
        }
    };
    // two named score accumulators ping-pong (no register copies): product u+1 is issued before softmax u.  The q fragments of the next
    // task are requested right after this task's LAST product (prefetch_q: loads the compiler does not count, see there).
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // this task's q fragments have landed (the 4 output stores of the previous task may still fly)
    f32x16 sa = qk_product(sKb), sb;
    if (i == 0 && have_next) prefetch_q(tn);
    for (int u = 0; u <= i; u += 2) {
      if (u + 1 <= i) {
        sb = qk_product(sKb + (u + 1) * 32 * G::ROW_BYTES);
        if (u + 1 == i && have_next) prefetch_q(tn);
      }
      softmax_pv(sa, u);
      if (u + 1 > i) break;
      if (u + 2 <= i) {
        sa = qk_product(sKb + (u + 2) * 32 * G::ROW_BYTES);
        if (u + 2 == i && have_next) prefetch_q(tn);
      }
      softmax_pv(sb, u + 1);
    }
    // ---- normalise and store this task's rows
    float ltot;
    {
      auto rr = __builtin_amdgcn_permlane32_swap(__float_as_uint(l), __float_as_uint(l), false, false);
      ltot = __uint_as_float(rr[0]) + __uint_as_float(rr[1]);
    }
    const float inv = 1.0f / ltot;
    const int q = q0 + r;
    // ---- store: lane (r, h) holds of row r the 4-element groups d = dt*32 + 8*g4 + 4*h + (0..3) -- written as they lie that is sixteen
    //      8-byte pieces per 128-B row from eight store instructions that each touch 64 lines (measured: 0.10 ms of a 0.64-ms launch,
    //      tools/exp/attn_resident_ablation.sh).  One v_permlane32_swap per register pair hands lane (r, 0) the partner's dt = 0 groups and
    //      lane (r, 1) the partner's dt = 1 groups: every lane then owns 64 contiguous bytes of its row, four 16-byte stores.
    uint32_t pk[DT][4][2];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          union { bf16x2 v; uint32_t u; } cv;
          cv.v[0] = o16(o[dt][4 * g4 + 2 * e] * inv, out_f16);
          cv.v[1] = o16(o[dt][4 * g4 + 2 * e + 1] * inv, out_f16);
          pk[dt][g4][e] = cv.u;
        }
    static_assert(DT == 2, "the lane-pair exchange below assumes two 32-wide d tiles");
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      u32x4 w;
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        // (vdst.upper <-> src.lower) lower lanes: [0] own dt0, [1] partner's dt0; upper lanes: [0] partner's dt1, [1] own dt1
        auto rr = __builtin_amdgcn_permlane32_swap(pk[0][g4][e], pk[1][g4][e], false, false);
        w[e] = rr[0];
        w[2 + e] = rr[1];
      }
      // a raw buffer store through the sequence's descriptor: rows >= len fall outside num_records and are dropped by the range check, so
      // the task always issues exactly four stores -- the compiler can then wait for the prefetched q fragments with vmcnt(4); with a
      // branch around each store it had to use vmcnt(0), which also waits for these stores to be acknowledged (~2 us per task)
      __builtin_amdgcn_raw_buffer_store_b128(w, orsrc, (q * nq + hq) * (D * 2) + 64 * h + 16 * g4, 0, 0);
    }
    t = tn;
  }
  __syncthreads();     // every wave is done with this pair's K/V before the next pair is staged over it
  }  // pairs
}

// ---------------------------------------------------------------------------------------------------------------
// Suffix-over-shared-prefix attention (EmbeddingBag construction, finetune/nonctx_emb_utils.py:239-313): every sequence
// is [prefix (identical for all sequences)] + [S2 own tokens].  The prefix K/V of this layer were computed once; here
// each suffix query attends to the P1 prefix keys plus its own suffix keys up to itself.  Tiny per-row work (P1 + S2 keys),
// so no MFMA: one wave per (sequence, q head), lane = head-dim element, scores by wave reduction.
// ---------------------------------------------------------------------------------------------------------------
template <int D>
__global__ void __launch_bounds__(512)
k_attn_prefix(const __bf16* __restrict__ qkv, const __bf16* __restrict__ prefix_kv, int n_seqs, int S2, int P1, int nq, int nkv,
              __bf16* __restrict__ out, float scale, int out_f16) {
  constexpr int E = D / 64;   // output elements per lane
  constexpr int JB = 4;       // queries handled per pass (they share the K/V reads)
  __shared__ float q_lds[8][JB][D];
  const int seq = blockIdx.x, hk = blockIdx.y;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int grp = nq / nkv;
  const int hq = hk * grp + wave;
  const int64_t RS = (int64_t)(nq + 2 * nkv) * D;          // suffix qkv row stride
  const int64_t PS = (int64_t)2 * nkv * D;                 // prefix kv row stride (k block | v block)
  const __bf16* kpre = prefix_kv + (int64_t)hk * D;
  const __bf16* vpre = prefix_kv + (int64_t)(nkv + hk) * D;
  const __bf16* ksuf = qkv + (int64_t)seq * S2 * RS + (int64_t)(nq + hk) * D;
  const __bf16* vsuf = qkv + (int64_t)seq * S2 * RS + (int64_t)(nq + nkv + hk) * D;
  for (int j0 = 0; j0 < S2; j0 += JB) {
    const int nj = min(JB, S2 - j0);
    // scaled queries of this pass -> LDS (every lane reads every element of them below)
    for (int jj = 0; jj < nj; ++jj)
#pragma unroll
      for (int e = 0; e < E; ++e)
        q_lds[wave][jj][lane * E + e] = attn_f(qkv[((int64_t)seq * S2 + j0 + jj) * RS + (int64_t)hq * D + lane * E + e]) * scale;
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): same-wave LDS writes visible to the reads below
    float m[JB], l[JB], o[JB][E];
#pragma unroll
    for (int jj = 0; jj < JB; ++jj) { m[jj] = -1e30f; l[jj] = 0.f;
#pragma unroll
      for (int e = 0; e < E; ++e) o[jj][e] = 0.f; }
    const int nkeys = P1 + j0 + nj;                        // keys visible to the last query of this pass
    for (int c0 = 0; c0 < nkeys; c0 += 64) {
      // ---- scores: lane = key, full head-dim dot product per lane
      const int key = c0 + lane;
      const bool valid = key < nkeys;
      const __bf16* kp = key < P1 ? kpre + (int64_t)key * PS : ksuf + (int64_t)(key - P1) * RS;
      float sc[JB];
#pragma unroll
      for (int jj = 0; jj < JB; ++jj) sc[jj] = 0.f;
      if (valid) {
#pragma unroll
        for (int e8 = 0; e8 < D / 8; ++e8) {
          const bf16x8 kv = *(const bf16x8*)(kp + e8 * 8);
          float kf[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) kf[i] = attn_f(kv[i]);
#pragma unroll
          for (int jj = 0; jj < JB; ++jj) {
            if (jj < nj) {
              const f32x4 qa = *(const f32x4*)&q_lds[wave][jj][e8 * 8], qb = *(const f32x4*)&q_lds[wave][jj][e8 * 8 + 4];
              sc[jj] += qa[0] * kf[0] + qa[1] * kf[1] + qa[2] * kf[2] + qa[3] * kf[3] + qb[0] * kf[4] + qb[1] * kf[5] + qb[2] * kf[6] + qb[3] * kf[7];
            }
          }
        }
      }
      float pr[JB], alpha[JB];
#pragma unroll
      for (int jj = 0; jj < JB; ++jj) {
        const bool vis = valid && key <= P1 + j0 + jj;     // causal: query j0+jj sees prefix + own suffix keys <= itself
        const float s_ = vis ? sc[jj] : -1e30f;
        const float mn = fmaxf(m[jj], wave_max(s_));
        alpha[jj] = __expf(m[jj] - mn);
        const float pw = vis ? __expf(s_ - mn) : 0.f;
        l[jj] = l[jj] * alpha[jj] + wave_sum(pw);
        pr[jj] = attn_f(attn_cvt(pw));                      // P is rounded to fp16 before P.V like the tiled kernels
        m[jj] = mn;
#pragma unroll
        for (int e = 0; e < E; ++e) o[jj][e] *= alpha[jj];
      }
      // ---- P.V: lane = head-dim element, uniform loop over the keys of this chunk
      const int kend = min(64, nkeys - c0);
      for (int kk = 0; kk < kend; ++kk) {
        const int key_u = c0 + kk;
        const __bf16* vp = key_u < P1 ? vpre + (int64_t)key_u * PS : vsuf + (int64_t)(key_u - P1) * RS;
        float vf[E];
#pragma unroll
        for (int e = 0; e < E; ++e) vf[e] = attn_f(vp[lane * E + e]);
#pragma unroll
        for (int jj = 0; jj < JB; ++jj) {
          const float pk = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pr[jj]), kk));
#pragma unroll
          for (int e = 0; e < E; ++e) o[jj][e] += pk * vf[e];
        }
      }
    }
#pragma unroll
    for (int jj = 0; jj < JB; ++jj) {
      if (jj < nj) {
        const float inv = 1.0f / l[jj];
#pragma unroll
        for (int e = 0; e < E; ++e)
          out[((int64_t)seq * S2 + j0 + jj) * ((int64_t)nq * D) + (int64_t)hq * D + lane * E + e] = o16(o[jj][e] * inv, out_f16);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// The same on the matrix cores (round 2; P1 <= 64, grp * S2 <= 16 waves at d = 64, 12 at d = 128; other shapes keep the kernel above).  Workgroup = (block of 32 sequences, kv head); wave w =
// (q head w % grp of the group, suffix position j = w / grp): its 32 query rows are the j-th suffix token of 32 consecutive sequences.
// The prefix K/V of the kv head (identical for every sequence) sit in LDS once per workgroup, in the K / V tile images of the kernels
// above: S^T = Kpre . Q^T by MFMA (one q row per lane -> lane-local softmax), O^T += Vpre^T . P^T by MFMA.  Only the <= S2 own keys of
// a row (different for every row) are VALU work: a 2 x d/2-element dot product per key, split over the two lanes that share the row.
// ---------------------------------------------------------------------------------------------------------------
template <int D>
__global__ void __launch_bounds__(D == 64 ? 1024 : 768)     // d = 128 needs ~150 VGPRs: at most 12 waves per workgroup
k_attn_prefix_mfma(const __bf16* __restrict__ qkv, const __bf16* __restrict__ prefix_kv, int n_seqs, int S2, int P1, int nq, int nkv,
                   __bf16* __restrict__ out, float scale_log2, int out_f16) {
  using G = AttnGeom<D>;
  constexpr int KS = D / 16, DT = D / 32;
  constexpr int MAXP = 64;                          // prefix keys held in LDS (two 32-key MFMA tiles)
  __shared__ __attribute__((aligned(1024))) char smem[2 * MAXP * G::ROW_BYTES];   // Kpre image | Vpre image
  char* const sK = smem;
  char* const sV = smem + MAXP * G::ROW_BYTES;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nwaves = blockDim.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int grp = nq / nkv;
  const int hk = blockIdx.y, seq0 = blockIdx.x * 32;
  const int hq = hk * grp + wave % grp, j = wave / grp;
  const int64_t RS = (int64_t)(nq + 2 * nkv) * D, PS = (int64_t)2 * nkv * D;
  const int nt = (P1 + 31) >> 5;                    // 32-key prefix tiles (1 or 2)

  // ---- prefix K/V of this kv head -> LDS (rows >= P1: clamped copies, masked below)
  {
    const __bf16* kpre = prefix_kv + (int64_t)hk * D;
    const __bf16* vpre = prefix_kv + (int64_t)(nkv + hk) * D;
    const int ninst = nt * 32 * G::ROW_BYTES / 1024;
    for (int jj = wave; jj < ninst; jj += nwaves) {
      const int s_ = jj * 64 + lane;
      const int row = s_ / G::CH, cs = s_ % G::CH;
      const int grow = min(row, P1 - 1);
      __builtin_amdgcn_global_load_lds((gptr_t)(kpre + (int64_t)grow * PS + ((cs ^ G::xk(row)) << 3)), (lptr_t)(sK + jj * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(vpre + (int64_t)grow * PS + ((cs ^ G::xv(row)) << 3)), (lptr_t)(sV + jj * 1024), 16, 0, 0);
    }
  }
  // ---- this lane's query row and its own suffix keys / values (issued before the wait: they overlap the staging)
  const int seq = min(seq0 + r, n_seqs - 1);
  const __bf16* rowq = qkv + ((int64_t)seq * S2 + j) * RS;
  bf16x8 qf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) qf[ks] = *(const bf16x8*)(rowq + (int64_t)hq * D + ks * 16 + h * 8);
  float so[4];                                      // own-key scores (keys 0..j of the sequence's suffix), raw dot products
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    so[i] = -1e30f;
    if (i <= j) {
      const __bf16* kp = qkv + ((int64_t)seq * S2 + i) * RS + (int64_t)(nq + hk) * D + h * 8;
      float acc = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const bf16x8 kv = *(const bf16x8*)(kp + ks * 16);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc += attn_f(qf[ks][e]) * attn_f(kv[e]);
      }
      auto rr = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc), __float_as_uint(acc), false, false);
      so[i] = __uint_as_float(rr[0]) + __uint_as_float(rr[1]);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // ---- prefix scores: S^T[key, q] per 32-key tile; reg t of lane (r, h) = key (t&3) + 8(t>>2) + 4h of the tile, query row r
  f32x16 sc[2];
  float mloc = -1e30f;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    if (t < nt) {
      bf16x8 kf[KS];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) kf[ks] = *(const bf16x8*)(sK + (t * 32 + r) * G::ROW_BYTES + (((2 * ks + h) ^ G::xk(r)) << 4));
      const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      f32x16 acc = attn_mfma(kf[0], qf[0], zero);
#pragma unroll
      for (int ks = 1; ks < KS; ++ks) acc = attn_mfma(kf[ks], qf[ks], acc);
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int key = t * 32 + (u & 3) + 8 * (u >> 2) + 4 * h;
        acc[u] = key < P1 ? acc[u] : -1e30f;
        mloc = fmaxf(mloc, acc[u]);
      }
      sc[t] = acc;
    }
  }
  {
    auto rr = __builtin_amdgcn_permlane32_swap(__float_as_uint(mloc), __float_as_uint(mloc), false, false);
    mloc = fmaxf(__uint_as_float(rr[0]), __uint_as_float(rr[1]));
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) mloc = fmaxf(mloc, so[i]);
  const float mc = -mloc * scale_log2;
  float l = 0.f;
  // ---- O^T = Vpre^T . P^T over the prefix tiles
  f32x16 o[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int u = 0; u < 16; ++u) o[dt][u] = 0.f;
  const int g16 = lane >> 4, i16 = lane & 15, qd = i16 >> 2, p4 = i16 & 3;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    if (t < nt) {
      s16x4 vt[2][DT][2];
      const uint32_t vb = (uint32_t)(uintptr_t)(lds_char_ptr)(sV + (t * 32 + 4 * h) * G::ROW_BYTES);
#pragma unroll
      for (int sp = 0; sp < 2; ++sp)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          const uint32_t a = vb + sp * 16 * G::ROW_BYTES + qd * G::ROW_BYTES + (((dt * 4 + 2 * (g16 & 1) + (p4 >> 1)) ^ G::xv(qd)) << 4) + 8 * (p4 & 1);
          asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(vt[sp][dt][0]) : "v"(a));
          asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(vt[sp][dt][1]) : "v"(a), "i"(8 * G::ROW_BYTES));
        }
      bf16x8 pf[2];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const float pw = __builtin_amdgcn_exp2f(sc[t][u] * scale_log2 + mc);
        l += pw;
        pf[u >> 3][u & 7] = attn_cvt(pw);
      }
      if (DT == 2) {
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(vt[0][0][0]), "+v"(vt[0][0][1]), "+v"(vt[0][1][0]), "+v"(vt[0][1][1]), "+v"(vt[1][0][0]), "+v"(vt[1][0][1]),
                       "+v"(vt[1][1][0]), "+v"(vt[1][1][1])
                     :
                     : "memory");
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(vt[0][0][0]), "+v"(vt[0][0][1]), "+v"(vt[0][1][0]), "+v"(vt[0][1][1]), "+v"(vt[0][2 % DT][0]), "+v"(vt[0][2 % DT][1]),
                       "+v"(vt[0][3 % DT][0]), "+v"(vt[0][3 % DT][1]), "+v"(vt[1][0][0]), "+v"(vt[1][0][1]), "+v"(vt[1][1][0]), "+v"(vt[1][1][1]),
                       "+v"(vt[1][2 % DT][0]), "+v"(vt[1][2 % DT][1]), "+v"(vt[1][3 % DT][0]), "+v"(vt[1][3 % DT][1])
                     :
                     : "memory");
      }
#pragma unroll
      for (int sp = 0; sp < 2; ++sp)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          union { struct { s16x4 a, b; } s; bf16x8 v; } u_;
          u_.s.a = vt[sp][dt][0]; u_.s.b = vt[sp][dt][1];
          o[dt] = attn_mfma(u_.v, pf[sp], o[dt]);
        }
    }
  }
  {
    auto rr = __builtin_amdgcn_permlane32_swap(__float_as_uint(l), __float_as_uint(l), false, false);
    l = __uint_as_float(rr[0]) + __uint_as_float(rr[1]);       // both lanes of a row: the row's prefix sum
  }
  // ---- own keys: p rounded to bf16 before P.V like everywhere else; lane (r, h) owns output columns dt*32 + 8g + 4h + e
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (i <= j) {
      const float pw = __builtin_amdgcn_exp2f(so[i] * scale_log2 + mc);
      l += pw;
      const float pb = attn_f(attn_cvt(pw));
      const __bf16* vp = qkv + ((int64_t)seq * S2 + i) * RS + (int64_t)(nq + nkv + hk) * D + 4 * h;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const bf16x4 vv = *(const bf16x4*)(vp + dt * 32 + 8 * g);
#pragma unroll
          for (int e = 0; e < 4; ++e) o[dt][4 * g + e] += pb * attn_f(vv[e]);
        }
    }
  }
  if (seq0 + r < n_seqs) {
    const float inv = 1.0f / l;
    __bf16* op = out + ((int64_t)seq * S2 + j) * ((int64_t)nq * D) + (int64_t)hq * D + 4 * h;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        bf16x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = o16(o[dt][4 * g + e] * inv, out_f16);
        *(bf16x4*)(op + dt * 32 + 8 * g) = v;
      }
  }
}

extern "C" int lrx_attn_prefix_suffix(const void* qkv, const void* prefix_kv, int32_t n_seqs, int32_t suffix_len, int32_t prefix_len,
                                      int32_t num_q_heads, int32_t num_kv_heads, int32_t head_dim, void* out, void* stream) {
  return lrx_attn_prefix_suffix_ex(qkv, prefix_kv, n_seqs, suffix_len, prefix_len, num_q_heads, num_kv_heads, head_dim, out, 0, stream);
}
// (out_f16 != 0: the output rows are written as fp16 -- the O-projection's operand under lrx_encoder_config.precise_stream = 2)
int lrx_attn_prefix_suffix_ex(const void* qkv, const void* prefix_kv, int32_t n_seqs, int32_t suffix_len, int32_t prefix_len, int32_t num_q_heads,
                              int32_t num_kv_heads, int32_t head_dim, void* out, int out_f16, void* stream) {
  LRX_CHECK_ARG(head_dim == 64 || head_dim == 128, "attn_prefix: head_dim=%d unsupported", head_dim);
  LRX_CHECK_ARG(num_kv_heads > 0 && num_q_heads % num_kv_heads == 0 && num_q_heads / num_kv_heads <= 8, "attn_prefix: bad head counts");
  LRX_CHECK_ARG(suffix_len > 0 && prefix_len >= 0, "attn_prefix: bad lengths");
  if (n_seqs == 0) return LRX_OK;
  const float scale = 1.0f / sqrtf((float)head_dim);
  const int grp_ = num_q_heads / num_kv_heads;
  if (prefix_len >= 1 && prefix_len <= 64 && suffix_len <= 4 && grp_ * suffix_len <= (head_dim == 64 ? 16 : 12)) {   // matrix-core kernel
    dim3 g((unsigned)lrx_cdiv(n_seqs, 32), num_kv_heads), b(64 * grp_ * suffix_len);
    const float sl2 = scale * 1.4426950408889634f;
    if (head_dim == 64)
      hipLaunchKernelGGL(k_attn_prefix_mfma<64>, g, b, 0, (hipStream_t)stream, (const __bf16*)qkv, (const __bf16*)prefix_kv, n_seqs, suffix_len,
                         prefix_len, num_q_heads, num_kv_heads, (__bf16*)out, sl2, out_f16);
    else
      hipLaunchKernelGGL(k_attn_prefix_mfma<128>, g, b, 0, (hipStream_t)stream, (const __bf16*)qkv, (const __bf16*)prefix_kv, n_seqs, suffix_len,
                         prefix_len, num_q_heads, num_kv_heads, (__bf16*)out, sl2, out_f16);
    LRX_LAUNCH_CHECK();
    return LRX_OK;
  }
  dim3 grid(n_seqs, num_kv_heads), block(64 * (num_q_heads / num_kv_heads));
  if (head_dim == 64)
    hipLaunchKernelGGL(k_attn_prefix<64>, grid, block, 0, (hipStream_t)stream, (const __bf16*)qkv, (const __bf16*)prefix_kv, n_seqs, suffix_len,
                       prefix_len, num_q_heads, num_kv_heads, (__bf16*)out, scale, out_f16);
  else
    hipLaunchKernelGGL(k_attn_prefix<128>, grid, block, 0, (hipStream_t)stream, (const __bf16*)qkv, (const __bf16*)prefix_kv, n_seqs, suffix_len,
                       prefix_len, num_q_heads, num_kv_heads, (__bf16*)out, scale, out_f16);
  LRX_LAUNCH_CHECK();
  return LRX_OK;
}

// CU count of the device that was current at the FIRST call (cached per process; include/lrx.h says so).  A C++11 function-local static:
// concurrent first calls from several host threads initialise it once.
static int attn_cu_count() {
  static const int n = []() {
    int dev = 0;
    hipDeviceProp_t prop;
    return (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
  }();
  return n;
}

template <int D, int GRP>
static int launch_attn(const void* qkv, const int32_t* cu, int n_seqs, int max_seqlen, int nq, int nkv, void* out, int last_tile_only,
                       hipStream_t s, int nparts, int out_f16) {
  int nqt = (int)lrx_cdiv(max_seqlen, 64);
  float scale_log2 = (1.0f / sqrtf((float)D)) * 1.4426950408889634f;
  const int64_t n_items = (int64_t)(last_tile_only ? n_seqs : n_seqs * nqt) * nkv * nparts;
  LRX_CHECK_ARG(n_items < (1ll << 31), "attn: %lld work items", (long long)n_items);
  // persistent workgroups, as many as the chip keeps resident at once: the d = 128 ring (96 KiB) and the 256-VGPR budget admit one
  // per CU; d = 64 workgroups are small enough for more (2-stage 32-KiB rings), so they get a slot count that covers that
  const int n_cu = attn_cu_count();
  const int per_cu = D == 128 ? 1 : (GRP <= 2 ? 4 : 2);
  const int64_t slots = (int64_t)n_cu * per_cu;
  // grouped item list (see the kernel): needs a full grid that splits evenly over the 8 XCDs; gs = slots of one XCD that share a
  // pair's K/V = the largest power of two <= min(q tiles, slots per XCD)
  int gs = 0;
  if (!last_tile_only && n_items >= slots && slots % 8 == 0) {
    const int spx = (int)(slots / 8);
    gs = 1;
    while (gs * 2 <= nqt && gs * 2 <= spx && spx % (gs * 2) == 0) gs *= 2;
  }
  hipLaunchKernelGGL((k_attn_varlen_causal<D, GRP>), dim3((unsigned)(n_items < slots ? n_items : slots)), dim3(128 * GRP), 0, s,
                     (const __bf16*)qkv, cu, nqt, nq, nkv, (__bf16*)out, scale_log2, last_tile_only, nparts, n_seqs, (int)n_items, gs, out_f16);
  LRX_LAUNCH_CHECK();
  return LRX_OK;
}

// ---- the item-list kernel: plan (shared by the list size, the list builder and the launch), builder, launch
struct AttnPlan { int grp, nparts, nqt, gs, n_wg, slots; int64_t n_items; };
static AttnPlan attn_plan(int n_seqs, int max_seqlen, int nq, int nkv, int head_dim, int last_tile_only) {
  AttnPlan p;
  const int g = nq / nkv;
  if (head_dim == 64) { p.grp = g <= 8 ? g : 8; p.nparts = g <= 8 ? 1 : (g + 7) / 8; }
  else if (g <= 4) { p.grp = g; p.nparts = 1; }
  else if (g <= 6) { p.grp = 3; p.nparts = 2; }
  else { p.grp = 4; p.nparts = (g + 3) / 4; }
  p.nqt = (int)lrx_cdiv(max_seqlen, 64);
  p.n_items = (int64_t)(last_tile_only ? n_seqs : (int64_t)n_seqs * p.nqt) * nkv * p.nparts;
  const int per_cu = head_dim == 128 ? 1 : (p.grp <= 2 ? 4 : 2);
  // (at most 1024 workgroups: the list builder is ONE 1024-thread block with a thread per workgroup -- 256 CUs x 4 on MI355X is exactly
  // that; a part with more CUs runs the same persistent walk on 1024 of its slots instead of failing)
  const int64_t slots_chip = (int64_t)attn_cu_count() * per_cu;
  const int64_t slots = slots_chip < 1024 ? slots_chip : 1024;
  p.gs = 0;
  if (!last_tile_only && p.n_items >= slots && slots % 8 == 0) {
    const int spx = (int)(slots / 8);
    p.gs = 1;
    while (p.gs * 2 <= p.nqt && p.gs * 2 <= spx && spx % (p.gs * 2) == 0) p.gs *= 2;
  }
  p.n_wg = (int)(p.n_items < slots ? p.n_items : slots);
  p.slots = (int)slots;
  return p;
}
__device__ int g_attn_items_overflow;
extern "C" int lrx_debug_attn_items_overflow(int* out) {   // tests: the builder never ran out of list slots (synchronises)
  LRX_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_attn_items_overflow), sizeof(int)));
  return LRX_OK;
}
unsigned int lrx_attn_list_overflows(int* ok, int reset) {   // part of lrx_device_error_count (lrx_elementwise.hip)
  int v = 0;
  *ok = hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_attn_items_overflow), sizeof(v)) == hipSuccess;
  if (*ok && reset && v) {
    const int z = 0;
    *ok = hipMemcpyToSymbol(HIP_SYMBOL(g_attn_items_overflow), &z, sizeof(z)) == hipSuccess;
  }
  return (unsigned int)v;
}
static int attn_check_layout(int32_t num_q_heads, int32_t num_kv_heads, int32_t head_dim) {
  LRX_CHECK_ARG(head_dim == 64 || head_dim == 128, "attn: head_dim=%d unsupported (64 or 128)", head_dim);
  LRX_CHECK_ARG(num_kv_heads > 0 && num_q_heads % num_kv_heads == 0, "attn: nq=%d not a multiple of nkv=%d", num_q_heads, num_kv_heads);
  return LRX_OK;
}
// layout of a work list: int32 list_start[hdr] | 16-byte items (k_attn_build_items); the header is sized for the chip's slot count, so the
// items start at the same offset whatever the batch
static size_t attn_list_hdr_bytes(const AttnPlan& p) { return (size_t)((p.slots + 1 + 3) & ~3) * 4; }
static int64_t attn_list_item_bound(const AttnPlan& p, int n_seqs, int total_tokens, int nkv, int last_tile_only) {
  // every sequence has ceil(len / 64) q tiles: at most total_tokens / 64 + n_seqs over the batch, and at most n_seqs x nqt
  const int64_t by_tokens = (int64_t)total_tokens / 64 + n_seqs, by_tiles = (int64_t)n_seqs * p.nqt;
  const int64_t tiles = last_tile_only ? n_seqs : (by_tokens < by_tiles ? by_tokens : by_tiles);
  return tiles * nkv * p.nparts + p.slots + 3;      // + an end marker per workgroup + padding
}
extern "C" size_t lrx_attn_items_bytes(int32_t n_seqs, int32_t total_tokens, int32_t max_seqlen, int32_t num_q_heads, int32_t num_kv_heads,
                                       int32_t head_dim, int32_t last_tile_only) {
  if (n_seqs <= 0 || max_seqlen <= 0 || total_tokens <= 0 || attn_check_layout(num_q_heads, num_kv_heads, head_dim) != LRX_OK) return 0;
  const AttnPlan p = attn_plan(n_seqs, max_seqlen, num_q_heads, num_kv_heads, head_dim, last_tile_only);
  return attn_list_hdr_bytes(p) + 16 * (size_t)attn_list_item_bound(p, n_seqs, total_tokens, num_kv_heads, last_tile_only);
}
static bool attn_uses_resident64(int head_dim, int max_seqlen, int last_tile_only);
extern "C" int lrx_attn_build_items(const int32_t* cu_seqlens, int32_t n_seqs, int32_t total_tokens, int32_t max_seqlen, int32_t num_q_heads,
                                    int32_t num_kv_heads, int32_t head_dim, int32_t last_tile_only, void* items, size_t items_bytes, void* stream) {
  int rc = attn_check_layout(num_q_heads, num_kv_heads, head_dim);
  if (rc) return rc;
  if (n_seqs <= 0 || total_tokens <= 0) return LRX_OK;
  LRX_CHECK_ARG(max_seqlen > 0, "attn: max_seqlen must be > 0");
  // the launch with these arguments runs the K/V-resident kernel, which derives its tasks itself: nothing to build, nothing read
  if (attn_uses_resident64(head_dim, max_seqlen, last_tile_only)) return LRX_OK;
  const AttnPlan p = attn_plan(n_seqs, max_seqlen, num_q_heads, num_kv_heads, head_dim, last_tile_only);
  LRX_CHECK_ARG(p.n_items < (1ll << 31), "attn: %lld work items", (long long)p.n_items);
  LRX_CHECK_ARG(p.n_wg <= 1024, "attn: %d workgroups (the list builder is one block)", p.n_wg);   // (attn_plan clamps its slots to 1024)
  const size_t need = attn_list_hdr_bytes(p) + 16 * (size_t)attn_list_item_bound(p, n_seqs, total_tokens, num_kv_heads, last_tile_only);
  LRX_CHECK_ARG(items != nullptr && ((uintptr_t)items & 15) == 0, "attn: the work list must be 16-byte aligned device memory");
  LRX_CHECK_ARG(items_bytes >= need, "attn: work list %zu B < required %zu B", items_bytes, need);
  int* ovf = nullptr;
  LRX_HIP(hipGetSymbolAddress((void**)&ovf, HIP_SYMBOL(g_attn_items_overflow)));
  const int64_t cap_items = (int64_t)((items_bytes - attn_list_hdr_bytes(p)) / 16);
  hipLaunchKernelGGL(k_attn_build_items, dim3(1), dim3(1024), 0, (hipStream_t)stream, cu_seqlens, (int32_t*)items, (i32x4*)((char*)items + attn_list_hdr_bytes(p)),
                     (int)(cap_items < (1ll << 30) ? cap_items : (1ll << 30)), p.n_wg, p.nqt, num_kv_heads, p.nparts, n_seqs, (int)p.n_items, p.gs, last_tile_only, ovf);
  LRX_LAUNCH_CHECK();
  return LRX_OK;
}
static bool attn_uses_resident64(int head_dim, int max_seqlen, int last_tile_only) {
  // LRX_ATTN_TILED=1 (dev builds only): A/B runs of the two d = 64 kernels (read once; C++11 static initialisation is thread-safe)
  static const int force_tiled = lrx_dev_knob("LRX_ATTN_TILED", 0);
  return !force_tiled && head_dim == 64 && max_seqlen <= 512 && !last_tile_only;
}
static int launch_resident64(const void* qkv, const int32_t* cu_seqlens, int n_seqs, int num_q_heads, int num_kv_heads, void* out, hipStream_t s, int out_f16 = 0) {
  const float scale_log2 = (1.0f / sqrtf(64.0f)) * 1.4426950408889634f;
  const int n_cu64 = attn_cu_count();
  const int n_pairs = n_seqs * num_kv_heads;
  dim3 grid(n_pairs < n_cu64 ? n_pairs : n_cu64), block(1024);
  hipLaunchKernelGGL(k_attn_resident64, grid, block, 0, s, (const __bf16*)qkv, cu_seqlens, num_q_heads, num_kv_heads, (__bf16*)out, scale_log2, n_pairs, out_f16);
  LRX_LAUNCH_CHECK();
  return LRX_OK;
}
extern "C" int lrx_attn_varlen_causal_items(const void* qkv, const int32_t* cu_seqlens, const void* items, size_t items_bytes, int32_t n_seqs,
                                            int32_t total_tokens, int32_t max_seqlen, int32_t num_q_heads, int32_t num_kv_heads, int32_t head_dim,
                                            void* out, int32_t last_tile_only, void* stream) {
  return lrx_attn_varlen_causal_items_ex(qkv, cu_seqlens, items, items_bytes, n_seqs, total_tokens, max_seqlen, num_q_heads, num_kv_heads, head_dim, out,
                                         last_tile_only, 0, stream);
}
// (out_f16 != 0: the output rows are written as fp16 -- the O-projection's operand under lrx_encoder_config.precise_stream = 2)
int lrx_attn_varlen_causal_items_ex(const void* qkv, const int32_t* cu_seqlens, const void* items, size_t items_bytes, int32_t n_seqs, int32_t total_tokens,
                                    int32_t max_seqlen, int32_t num_q_heads, int32_t num_kv_heads, int32_t head_dim, void* out, int32_t last_tile_only,
                                    int out_f16, void* stream) {
  int rc = attn_check_layout(num_q_heads, num_kv_heads, head_dim);
  if (rc) return rc;
  LRX_CHECK_ARG(max_seqlen > 0 || total_tokens == 0, "attn: max_seqlen must be > 0");
  if (total_tokens == 0 || n_seqs == 0) return LRX_OK;
  const int nq = num_q_heads, nkv = num_kv_heads;
  hipStream_t s = (hipStream_t)stream;
  if (attn_uses_resident64(head_dim, max_seqlen, last_tile_only)) return launch_resident64(qkv, cu_seqlens, n_seqs, nq, nkv, out, s, out_f16);
  const AttnPlan p = attn_plan(n_seqs, max_seqlen, nq, nkv, head_dim, last_tile_only);
  const size_t need = attn_list_hdr_bytes(p) + 16 * (size_t)attn_list_item_bound(p, n_seqs, total_tokens, nkv, last_tile_only);
  LRX_CHECK_ARG(items != nullptr && items_bytes >= need, "attn: work list %zu B < required %zu B", items_bytes, need);
  // (a sequence's rows are addressed through one buffer descriptor: 32-bit byte offsets)
  LRX_CHECK_ARG((int64_t)max_seqlen * (nq + 2 * nkv) * head_dim * 2 < (1ll << 31), "attn: a sequence of %d rows x %d B exceeds a buffer descriptor", max_seqlen,
                (nq + 2 * nkv) * head_dim * 2);
  const float scale_log2 = (1.0f / sqrtf((float)head_dim)) * 1.4426950408889634f;
#define LRX_STREAM_CASE(DD, GG)                                                                                                                       \
  case GG: hipLaunchKernelGGL((k_attn_stream<DD, GG>), dim3((unsigned)p.n_wg), dim3(128 * GG), 0, s, (const __bf16*)qkv, (const int32_t*)items, \
                              (const i32x4*)((const char*)items + attn_list_hdr_bytes(p)), nq, nkv, (__bf16*)out, scale_log2, out_f16); break;
  if (head_dim == 64) {
    switch (p.grp) {
      LRX_STREAM_CASE(64, 1) LRX_STREAM_CASE(64, 2) LRX_STREAM_CASE(64, 3) LRX_STREAM_CASE(64, 4)
      LRX_STREAM_CASE(64, 5) LRX_STREAM_CASE(64, 6) LRX_STREAM_CASE(64, 7) LRX_STREAM_CASE(64, 8)
    }
  } else {
    switch (p.grp) { LRX_STREAM_CASE(128, 1) LRX_STREAM_CASE(128, 2) LRX_STREAM_CASE(128, 3) LRX_STREAM_CASE(128, 4) }
  }
#undef LRX_STREAM_CASE
  LRX_LAUNCH_CHECK();
  return LRX_OK;
}

extern "C" int lrx_attn_varlen_causal(const void* qkv, const int32_t* cu_seqlens, int32_t n_seqs, int32_t total_tokens,
                                      int32_t max_seqlen, int32_t num_q_heads, int32_t num_kv_heads, int32_t head_dim, void* out,
                                      int32_t last_tile_only, void* stream) {
  return lrx_attn_varlen_causal_ex(qkv, cu_seqlens, n_seqs, total_tokens, max_seqlen, num_q_heads, num_kv_heads, head_dim, out, last_tile_only, 0, stream);
}
int lrx_attn_varlen_causal_ex(const void* qkv, const int32_t* cu_seqlens, int32_t n_seqs, int32_t total_tokens, int32_t max_seqlen, int32_t num_q_heads,
                              int32_t num_kv_heads, int32_t head_dim, void* out, int32_t last_tile_only, int out_f16, void* stream) {
  int rc = attn_check_layout(num_q_heads, num_kv_heads, head_dim);
  if (rc) return rc;
  LRX_CHECK_ARG(max_seqlen > 0 || total_tokens == 0, "attn: max_seqlen must be > 0");
  if (total_tokens == 0 || n_seqs == 0) return LRX_OK;
  int grp = num_q_heads / num_kv_heads;
  hipStream_t s = (hipStream_t)stream;
  if (attn_uses_resident64(head_dim, max_seqlen, last_tile_only)) return launch_resident64(qkv, cu_seqlens, n_seqs, num_q_heads, num_kv_heads, out, s, out_f16);
  // No work list: the walker kernel (k_attn_varlen_causal) derives every item's (sequence, kv head, q tile) itself -- the variant for
  // callers without scratch memory; lrx_attn_varlen_causal_items is the faster launch.
  // Tiled kernel: GRP q heads per workgroup, nparts workgroups per kv head (heads beyond the group idle).  head_dim 64 takes up to 8
  // heads per workgroup; head_dim 128 needs ~190 VGPRs per wave, so at most 4 (more than 8 waves per workgroup would spill): groups
  // of 5-6 heads run as two workgroups of 3, 7-8 as two of 4 (the K/V tiles are staged twice, from L2), larger groups as ceil(grp/4).
#define LRX_ATTN_CASE(DD, GG, PARTS) \
  return launch_attn<DD, GG>(qkv, cu_seqlens, n_seqs, max_seqlen, num_q_heads, num_kv_heads, out, last_tile_only, s, PARTS, out_f16);
  if (head_dim == 64) {
    switch (grp) {
      case 1: LRX_ATTN_CASE(64, 1, 1)
      case 2: LRX_ATTN_CASE(64, 2, 1)
      case 3: LRX_ATTN_CASE(64, 3, 1)
      case 4: LRX_ATTN_CASE(64, 4, 1)
      case 5: LRX_ATTN_CASE(64, 5, 1)
      case 6: LRX_ATTN_CASE(64, 6, 1)
      case 7: LRX_ATTN_CASE(64, 7, 1)
      case 8: LRX_ATTN_CASE(64, 8, 1)
      default: LRX_ATTN_CASE(64, 8, (grp + 7) / 8)
    }
  }
  switch (grp) {
    case 1: LRX_ATTN_CASE(128, 1, 1)
    case 2: LRX_ATTN_CASE(128, 2, 1)
    case 3: LRX_ATTN_CASE(128, 3, 1)
    case 4: LRX_ATTN_CASE(128, 4, 1)
    case 5: case 6: LRX_ATTN_CASE(128, 3, 2)
    default: LRX_ATTN_CASE(128, 4, (grp + 3) / 4)
  }
#undef LRX_ATTN_CASE
}
