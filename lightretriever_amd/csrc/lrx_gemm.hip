// bf16 GEMM  C[M,N] = A[M,K] * B[N,K]^T  (nn.Linear layout, both operands K-contiguous) on gfx950 MFMA.
//
// Tile 256 x 256 x 64, 512 threads = 8 waves as 2 (M) x 4 (N).  Each operand tile is split into two 128-row HALF-TILES
// (16 KiB each, A0/A1/B0/B1); wave (wr, wc) owns rows wr*64..+63 of EACH A half and columns wc*32..+31 of EACH B half, so
// its 128 x 64 output is four 64 x 32 quadrants (A-half h) x (B-half h'), one quadrant = 16 MFMA 16x16x32 per K-tile.
//
// 4-phase ping-pong schedule (2 phases per K-tile, 2 K-tiles of LDS = 8 half-tile slots, 128 KiB):
//   phase = { ds_read the register sub-tiles of TWO quadrants ; issue LDS-DMA half-tiles (2 x global_load_lds dwordx4 per lane
//             each) ; counted s_waitcnt vmcnt(8) ; s_waitcnt lgkmcnt(0) ; s_barrier ; 32 MFMA under s_setprio(1) ; s_barrier }
//   P1(t): quadrants (A0,B0),(A0,B1): reads A0, B0, B1 (both B sub-tiles then stay in registers);  P2(t): (A1,B1),(A1,B0): reads A1.
//   A half-tile slot is refilled one phase after its last read (safe under the group stagger): P2(t) issues A0,B0,B1 of K-tile
//   t+2, P1(t+1) issues A1 of t+2; the 2 + 6 youngest DMA instructions stay in flight across every barrier (never vmcnt(0) in
//   steady state).  An 8-phase version (16 MFMA per barrier pair, one half-tile per phase, vmcnt(6)) measured 2.5-5 % slower on
//   all four encoder shapes: the barrier pairs, not the LDS reads, were the overhead.
//   The two wave groups (waves 0-3 / 4-7, one wave of each per SIMD) run offset by one barrier, so one group's MFMA
//   section overlaps the other group's LDS-read/issue section on every SIMD.
// LDS image per half-tile is lane-linear per LDS-DMA instruction; the bank swizzle (16-B chunk ^= (row>>1)&7) is applied
// to the per-lane SOURCE address and undone on the ds_read_b128 side -> conflict-free fragment reads.
//
// The MFMA takes the WEIGHT fragment as A and the ACTIVATION fragment as B, so a lane holds 4 consecutive output columns
// of one row: 8-byte stores, fused bias / residual, and lane-local SwiGLU pairs (gate/up interleaved in 16-row groups).
//
// Workgroup -> tile map: bijective XCD-chunked remap (blocks b and b+8 share an XCD L2) + 8-m-tile groups walked
// n-fastest inside an XCD chunk, so the 32 co-resident tiles of an XCD share 8 A panels and 4 B panels.
#include "lrx_common.h"
#include <float.h>
#include <stdlib.h>

#define GBM 256
#define GBN 256
#define GBK 64
#define HALF_BYTES 16384

enum { EPI_STORE = 0, EPI_RESID = 1, EPI_SWIGLU = 2, EPI_ROPE = 3, EPI_MAXAGG = 4, EPI_EMIT = 5, EPI_RESID32 = 6, EPI_SAMPLE = 7 };
// the two instantiations of the search: A = the shard's tiled fp16 shadow, B = fp16 queries, f16 MFMA
#define EPI_IS_SEARCH(E) ((E) == EPI_EMIT || (E) == EPI_SAMPLE)

// Fused QKV epilogue (EPI_ROPE, round 3): the rows of the q and k heads of Wqkv arrive in ROTARY-PAIR order (include/lrx.h: physical
// column 32 g + 16 i + t of a head = logical column i d/2 + 16 g + t), so the two accumulators a lane holds for a 16 x 32 block -- columns
// t and 16 + t of the block -- ARE a rotary pair (x_j, x_{j + d/2}): the rotation runs on the fp32 accumulators with the fp32 cos/sin
// table, one rounding, no partner read from the staged tile, and the result is stored as FP16 in the same pair order (q . k does not
// care about a permutation of the head dimension common to both).  v columns: plain fp16 store.
struct RopeArgs {
  const int32_t* positions;  // [M]
  const float* cos;          // [max_pos, d/2] fp32
  const float* sin;
  int rope_cols;             // columns [0, rope_cols) are q|k heads to rotate; the rest (v) is stored as is
  int head_dim;
  int ldc;                   // row stride of C in elements, 0 = N (a column slice of the q|k|v buffer: lrx_gemm_qkv_rope_slice)
};

__device__ __forceinline__ __bf16 f2h_bits(float v) {   // fp16 (saturating) in the kernel's 16-bit container type
  return __builtin_bit_cast(__bf16, (_Float16)fminf(fmaxf(v, -65504.f), 65504.f));
}
// q|k|v elements of the fused QKV epilogue that left fp16's range (|v| > 65504) or were NaN -- both become a finite 65504 in the store
// above, which is a silent change of the model's arithmetic: counted here, read by lrx_device_saturation_count (one atomic per wave
// that saw any, i.e. none on a healthy checkpoint).
__device__ unsigned int g_qkv_fp16_saturations = 0;
unsigned int lrx_gemm_saturations(int* ok) {                 // read only: lrx_device_saturation_count resets after BOTH counters were read
  unsigned int v = 0;
  *ok = hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_qkv_fp16_saturations), sizeof(v)) == hipSuccess;
  return v;
}
int lrx_gemm_saturations_reset() {
  const unsigned int z = 0;
  return hipMemcpyToSymbol(HIP_SYMBOL(g_qkv_fp16_saturations), &z, sizeof(z)) == hipSuccess;
}

// Search filter pass (EPI_EMIT): A = the shard's tiled FP16 shadow (lrx_shadow_off), B = fp16 queries (f16 MFMA); no C.  A score reaching thr[query] is appended
// to the query's candidate list (lrx_search.hip).  ss > 0: m-tile t of the launch is the t-th 256-row tile that is NOT in the sample
// (the sample = every ss-th tile).
// Sample pass (EPI_SAMPLE, round 6): m-tile t of the launch IS sample tile t (corpus tile t * ss); its scores go out as fp32 into the compact
// sample matrix (scores[query, ld_s], sample row 256 t + row) together with the maxima of its 16-row groups (gmax[query, 8 nblk_ld_s], 8 per
// 128-row block) -- what k_sample_threshold selects from (lrx_search.hip).
struct EmitArgs {
  const float* thr;          // [nq]
  unsigned long long* cand;  // [nq, cap]
  unsigned int* cnt;         // [nq * CNT_STRIDE]
  int ss;
  unsigned int cap;          // capacity of one candidate list
  float* scores;             // EPI_SAMPLE: [nq, ld_s]
  float* gmax;               //             [nq, 8 * nblk_ld_s]
  int64_t ld_s;
  int nblk_ld_s;
};

struct MaxAggArgs {
  const int32_t* row_seg;    // [M] output row (sequence index) each A row is reduced into, -1 = row takes no part
  float* out;                // [n_seqs, ldo] running maxima (pre-filled by the caller)
  int64_t ldo;
};

// RMSNorm folded around the GEMM (norm weight pre-multiplied into B's columns by the caller):
//   rscale  [M] fp32 or NULL: row m of the fp32 accumulator is multiplied by rscale[m] = rsqrt(mean(x_m^2) + eps) before bias /
//           RoPE / SwiGLU -- x . (gamma (*) W)^T * r  ==  (r x (*) gamma) . W^T, without materialising the normalised rows;
//   ss_part [tiles_n, M] fp32 or NULL (residual epilogue only): sum of squares of this tile's 256 columns of every OUTPUT row
//           (the bf16-rounded residual stream), one slot per n-tile, summed in fixed order by lrx_finalize_rscale: deterministic.
struct NormArgs {
  const float* rscale;
  float* ss_part;
  int group_m;               // m-tiles per group of the block -> tile map (rides along in this struct)
};

// max(*p, v) for floats with integer atomics: non-negative floats order like ints, negative ones inversely like uints
__device__ __forceinline__ void atomic_fmax_bits(float* p, float v) {
  const unsigned b = __float_as_uint(v);
  if (!(b & 0x80000000u)) __hip_atomic_fetch_max((int*)p, (int)b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else __hip_atomic_fetch_min((unsigned*)p, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

#define G_BAR()                            \
  do {                                     \
    __builtin_amdgcn_sched_barrier(0);     \
    __builtin_amdgcn_s_barrier();          \
    __builtin_amdgcn_sched_barrier(0);     \
  } while (0)
#define G_LSYNC()                                          \
  do {                                                     \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     \
    G_BAR();                                               \
  } while (0)



// bf16 x bf16 for the encoder GEMMs; the search filter's instantiation multiplies fp16 operands (same containers, same data movement)
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
// F16 (round 6, lrx_encoder_config.precise_stream = 2): the encoder GEMM's operands are FP16 too -- activations fp16(x * gamma), the attention output, the
// SwiGLU output, and the projection weights converted once at load (exact for |w| >= 6.1e-5): three more mantissa bits than bf16 at the
// same MFMA rate, worth 14-42 x in 1 - cos against the fp32 model (tools/exp/rounding_fp16_o_act.py); what the epilogue WRITES as the
// next GEMM's operand is fp16 then (saturating, counted like q|k|v).
template <int EPI, bool F16>
__device__ __forceinline__ f32x4 gemm_mfma(bf16x8 x, bf16x8 y, f32x4 c) {
  if constexpr (EPI_IS_SEARCH(EPI) || F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, x), __builtin_bit_cast(f16x8, y), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, c, 0, 0, 0);
}

// OUTH (EPI_RESID32): the operand this GEMM WRITES for the next projection is fp16 -- its own operands may still be bf16 (precise_stream = 3: only
// the QKV projection multiplies fp16, so the down-projection, bf16 x bf16, hands it an fp16 operand).
template <int EPI, bool F16 = false, bool OUTH = F16>
__global__ void __launch_bounds__(512, 2)
k_gemm_bf16_nt(const __bf16* __restrict__ A, const __bf16* __restrict__ B, __bf16* C, const __bf16* __restrict__ bias,
               const __bf16* resid, int M, int N, int K, int tiles_m, int tiles_n, RopeArgs rope, MaxAggArgs mx, NormArgs nrm, EmitArgs em) {
  // [buf 0/1][A0 | A1 | B0 | B1]; EPI_EMIT: + 2304 B behind the ring for the tile's thresholds and hit counters (set up at kernel start)
  __shared__ __attribute__((aligned(1024))) char smem[8 * HALF_BYTES + (EPI == EPI_EMIT ? 2304 : 0)];
  // ---- workgroup -> tile
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, qd = nwg >> 3, rm = nwg & 7;
  const int t_lin = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (bid >> 3);
  // m-tiles per group (a group sweeps all n-tiles): measured per shape on one box -- gate-up 6 (5.97 ms vs 6.28 at 8), down/o 4
  // (2.92 vs 3.02, 0.91 vs 0.93), qkv 8 (1.30 vs 1.34 at 6): the A group must share the 4-MiB L2 with the streaming B tiles
  const int GM = nrm.group_m;
  const int width = GM * tiles_n;
  const int group = t_lin / width, first_m = group * GM;
  const int gsz = min(tiles_m - first_m, GM);
  const int tin = t_lin - group * width;
  const int tm = first_m + tin % gsz, tn = tin / gsz;
  // (filter pass: the tm-th tile outside the strided sample)
  const int tmx = EPI == EPI_SAMPLE ? tm * em.ss : ((EPI == EPI_EMIT && em.ss > 1) ? (tm / (em.ss - 1)) * em.ss + 1 + tm % (em.ss - 1) : tm);
  const int m0 = tmx * GBM, n0 = tn * GBN;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  if constexpr (EPI == EPI_EMIT) {
    // the epilogue's thresholds of this tile's 256 queries and its zeroed hit counters, parked behind the ring NOW: their global loads and the
    // clearing barrier then cost the epilogue nothing (the K loop's barriers order these writes before the epilogue's reads)
    float* xthr = (float*)(smem + 8 * HALF_BYTES);
    unsigned int* xqc = (unsigned int*)(smem + 8 * HALF_BYTES + 1024);
    if (tid < GBN) { xthr[tid] = n0 + tid < N ? em.thr[n0 + tid] : FLT_MAX; xqc[tid] = 0u; }
    if (tid < 8) ((unsigned int*)(smem + 8 * HALF_BYTES + 2048))[tid] = 0u;
  }

  // ---- LDS-DMA sources: per half-tile 2 instructions per lane; lane l of instruction i fills 16-B slot
  //      s = (wave*2+i)*64 + l  ->  row = s>>3 (0..127), chunk position s&7 holding logical chunk (s&7) ^ ((row>>1)&7)
  const __bf16 *pA0[2], *pA1[2], *pB0[2], *pB1[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int s = (wave * 2 + i) * 64 + lane;
    int row = s >> 3, c = (s & 7) ^ ((row >> 1) & 7);
    if (EPI_IS_SEARCH(EPI)) {
      // tiled shadow: a half-tile (128 rows x 64) IS one 16-KiB fragment-major tile of the source -> copied linearly, 1 KiB per request
      // (the LDS image is then fragment-major too: laneoffA below); blocks past the last one re-read it, masked in the epilogue
      const int64_t lastb = (int64_t)(M - 1) >> 7;
      const int64_t b0 = min((int64_t)(m0 >> 7), lastb), b1 = min((int64_t)(m0 >> 7) + 1, lastb);
      pA0[i] = A + (b0 * (K / 64)) * 8192 + s * 8;
      pA1[i] = A + (b1 * (K / 64)) * 8192 + s * 8;
    } else {
      pA0[i] = A + (int64_t)min(m0 + row, M - 1) * K + c * 8;
      pA1[i] = A + (int64_t)min(m0 + 128 + row, M - 1) * K + c * 8;
    }
    pB0[i] = B + (int64_t)min(n0 + row, N - 1) * K + c * 8;
    pB1[i] = B + (int64_t)min(n0 + 128 + row, N - 1) * K + c * 8;
  }
  char* const dma_dst = smem + wave * 2048;  // + buf*65536 + slot*16384 + i*1024
#define G_KOFF(KT) ((KT) * GBK)
  const int a_ks = EPI_IS_SEARCH(EPI) ? 8192 : GBK;   // elements from one K-tile of an A row to the next
#define G_ISSUE_(P, SLOT, BUF, KOFF)                                                                                       \
  do {                                                                                                                     \
    __builtin_amdgcn_global_load_lds((gptr_t)(P[0] + (KOFF)), (lptr_t)(dma_dst + (BUF) * 65536 + (SLOT) * HALF_BYTES), 16, 0, 0);        \
    __builtin_amdgcn_global_load_lds((gptr_t)(P[1] + (KOFF)), (lptr_t)(dma_dst + (BUF) * 65536 + (SLOT) * HALF_BYTES + 1024), 16, 0, 0); \
  } while (0)
#define G_ISSUE(P, SLOT, BUF, KT) G_ISSUE_(P, SLOT, BUF, ((SLOT) < 2 ? (int64_t)(KT) * a_ks : (int64_t)G_KOFF(KT)))   /* slots 0/1 = A halves */

  // ---- fragment read offsets
  const int fr = lane & 15, fq = lane >> 4;
  const int xs = fr >> 1;
  int laneoff[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) laneoff[ks] = fr * 128 + (((ks * 4 + fq) ^ xs) << 4);
  int laneoffA[2];               // A-side fragment offsets: the swizzled row image, or (tiled shadow) the fragment-major tile [16-row group][ks][lane]
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) laneoffA[ks] = EPI_IS_SEARCH(EPI) ? ks * 1024 + lane * 16 : laneoff[ks];
  const int a_off = (wr * 64) * 128, b_off = 2 * HALF_BYTES + (wc * 32) * 128;

  f32x4 acc[2][2][4][2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int hp = 0; hp < 2; ++hp)
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) acc[h][hp][mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 a[4][2], b[2][2];

#define G_LDA(H)                                                                                     \
  _Pragma("unroll") for (int mi = 0; mi < 4; ++mi) _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)  \
      a[mi][ks] = *(const bf16x8*)(sbuf + (H) * HALF_BYTES + a_off + mi * 2048 + laneoffA[ks]);
#define G_LDB(HP)                                                                                    \
  _Pragma("unroll") for (int ni = 0; ni < 2; ++ni) _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)  \
      b[ni][ks] = *(const bf16x8*)(sbuf + (HP) * HALF_BYTES + b_off + ni * 2048 + laneoff[ks]);
#define G_MM(H, HP)                                                                                   \
  do {                                                                                                \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int mi = 0; mi < 4; ++mi) \
        _Pragma("unroll") for (int ni = 0; ni < 2; ++ni) acc[H][HP][mi][ni] =                        \
            gemm_mfma<EPI, F16>(b[ni][ks], a[mi][ks], acc[H][HP][mi][ni]); \
  } while (0)

  // folded RMSNorm: this lane's 8 row scales, requested now so they have long arrived when the epilogue multiplies
  float rsv[2][4];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
      rsv[h][mi] = (EPI != EPI_MAXAGG && EPI != EPI_RESID && !EPI_IS_SEARCH(EPI) && nrm.rscale != nullptr) ? nrm.rscale[min(m0 + h * 128 + wr * 64 + mi * 16 + fr, M - 1)] : 1.0f;

  int rposv[2][4];             // EPI_ROPE: positions of this lane's 8 rows (the table lookups of the epilogue depend on them)
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) rposv[h][mi] = EPI == EPI_ROPE ? rope.positions[min(m0 + h * 128 + wr * 64 + mi * 16 + fr, M - 1)] : 0;

  const int nk = K / GBK;
  // ---- prologue: K-tile 0 landed, K-tile 1 (issued in the steady-state order A0,B0,B1 then A1) stays in flight
  G_ISSUE(pA0, 0, 0, 0); G_ISSUE(pB1, 3, 0, 0); G_ISSUE(pA1, 1, 0, 0); G_ISSUE(pB0, 2, 0, 0);
  if (nk > 1) {
    G_ISSUE(pA0, 0, 1, 1); G_ISSUE(pB0, 2, 1, 1); G_ISSUE(pB1, 3, 1, 1); G_ISSUE(pA1, 1, 1, 1);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }

  G_BAR();
  if (wr == 1) G_BAR();  // stagger: group 1 runs one barrier behind group 0

  bf16x8 b2[2][2];
#define G_LDB2(HP)                                                                                   \
  _Pragma("unroll") for (int ni = 0; ni < 2; ++ni) _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)  \
      b2[ni][ks] = *(const bf16x8*)(sbuf + (HP) * HALF_BYTES + b_off + ni * 2048 + laneoff[ks]);
#define G_MM2(H, HP)                                                                                  \
  do {                                                                                                \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int mi = 0; mi < 4; ++mi) \
        _Pragma("unroll") for (int ni = 0; ni < 2; ++ni) acc[H][HP][mi][ni] =                        \
            gemm_mfma<EPI, F16>(b2[ni][ks], a[mi][ks], acc[H][HP][mi][ni]); \
  } while (0)
  // Lifetimes: A0,B0,B1 of K-tile t are read in P1(t) (B fragments stay in registers), A1 in P2(t).  Refill one phase after
  // the last read (stagger-safe): P2(t) issues A0,B0,B1 of t+2, P1(t) issues A1 of t+1 (t >= 1; K-tile 1's comes from the
  // prologue).  Counted waits: vmcnt(8) in both phases = the 2 + 6 youngest DMA instructions stay in flight.
  for (int t = 0; t < nk; ++t) {
    const int cur = t & 1;
    const char* sbuf = smem + cur * 65536;
    const bool n1 = t + 1 < nk, n2 = t + 2 < nk;
    // P1: quadrants (A0,B0) and (A0,B1)
    G_LDB(0) G_LDB2(1) G_LDA(0)
    if (t >= 1 && n1) G_ISSUE(pA1, 1, cur ^ 1, t + 1);          // A1(t+1): its slot (A1 of t-1) was last read in P2(t-1)
    if (n1 && n2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // A1(t) landed
    G_LSYNC();
    __builtin_amdgcn_s_setprio(1);
    G_MM(0, 0); G_MM2(0, 1);
    __builtin_amdgcn_s_setprio(0);
    G_BAR();
    // P2: quadrants (A1,B1) and (A1,B0)
    G_LDA(1)
    if (n2) {
      G_ISSUE(pA0, 0, cur, t + 2); G_ISSUE(pB0, 2, cur, t + 2); G_ISSUE(pB1, 3, cur, t + 2);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    G_LSYNC();
    __builtin_amdgcn_s_setprio(1);
    G_MM2(1, 1); G_MM(1, 0);
    __builtin_amdgcn_s_setprio(0);
    G_BAR();
  }
  if (wr == 0) G_BAR();

  // ---- epilogue.  The accumulator layout (lane = 1 row x 4 columns per 16x16 tile) would give 8-byte stores that touch 16
  //      partial lines per wave instruction (measured: 6-20 us per tile).  Instead the bf16 C tile is staged through the
  //      (now idle) 128 KiB of LDS -- 16-B chunk index XOR (row & 15): conflict-free ds_write_b64 and ds_read_b128 -- and
  //      written row-major, 16 B per lane: every wave instruction moves two full 512-B row segments.  The residual is
  //      read the same way and added to the bf16-rounded product (exactly the reference's bf16 `residual + linear(x)`).
  if constexpr (EPI == EPI_SAMPLE) {
    // ---- sample pass of the search: fp32 scores into the compact sample matrix [query, sample row] + the maxima of the 16-row groups.
    //      Lane (fr, fq) of wave (wr, wc) holds rows h*128 + wr*64 + mi*16 + fr and query columns hp*128 + wc*32 + ni*16 + fq*4 + r: stored
    //      straight from that layout a wave instruction writes four 64-B pieces of four queries' rows (measured: 610 us for the 125 MB of a
    //      1000-query sample of 1M rows).  Instead the tile goes through the idle LDS TRANSPOSED, one 128-query half at a time (128 KiB):
    //      LDS[query][256 rows] fp32, 16-B chunk index XOR ((query >> 2) & 3) << 2 -- the four fq lanes of a write land in four different
    //      16-bank windows, a read-out instruction takes one query's whole 1-KiB row run (one permuted chunk per lane) -- and leaves as full
    //      1-KiB runs; the 16 rows of a group are then the four lanes of a quad: two DPP steps give its maximum.
    const int64_t gstride = 8 * (int64_t)em.nblk_ld_s;
#pragma unroll
    for (int hp = 0; hp < 2; ++hp) {
      if (hp) __syncthreads();                                   // the first half has been read out
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
          const int row = h * 128 + wr * 64 + mi * 16 + fr;
          const bool mok = (int64_t)m0 + row < M;
#pragma unroll
          for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int cl = wc * 32 + ni * 16 + fq * 4 + r;     // query column inside this half; (cl >> 2) & 3 == fq
              *(float*)(smem + cl * 1024 + ((((row >> 2) ^ (fq << 2))) << 4) + (row & 3) * 4) = mok ? acc[h][hp][mi][ni][r] : -FLT_MAX;
            }
        }
      __syncthreads();
#pragma unroll 4
      for (int it = 0; it < 16; ++it) {
        const int cl = it * 8 + wave, col = n0 + hp * 128 + cl;  // (wave-uniform)
        const f32x4 v = *(const f32x4*)(smem + cl * 1024 + ((lane ^ (((cl >> 2) & 3) << 2)) << 4));   // rows 4 lane .. 4 lane + 3 of query `col`
        float mx = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
        mx = fmaxf(mx, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(mx), 0x4E, 0xF, 0xF, true)));    // quad_perm [2,3,0,1]
        mx = fmaxf(mx, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(mx), 0xB1, 0xF, 0xF, true)));    // quad_perm [1,0,3,2]
        if (col < N) {
          *(f32x4*)(em.scores + (int64_t)col * em.ld_s + (int64_t)tm * GBM + lane * 4) = v;
          if ((lane & 3) == 0) em.gmax[(int64_t)col * gstride + (int64_t)tm * 16 + (lane >> 2)] = mx;   // group (128-row block 2 tm + row / 128, 16-row group) = 16 tm + row / 16
        }
      }
    }
    return;
  }
  if constexpr (EPI == EPI_EMIT) {
    // ---- search filter epilogue: nothing is stored per (row, query).  Lane (fr, fq) of wave (wr, wc) holds rows h*128 + wr*64 +
    //      mi*16 + fr and query columns hp*128 + wc*32 + ni*16 + fq*4 + r.  Hits (~5e-3 of the scores) first go to a per-wave list in
    //      the idle LDS (an LDS atomic per 16x16 sub-tile that has any: ~100 cycles, against ~2 us for a global one, and this
    //      workgroup has the CU to itself).  A parked hit also takes its rank among the WORKGROUP's hits of its query (a second LDS
    //      atomic), so the global slots of a query are reserved by ONE atomic per (tile, query): 485 -> 428 us per launch where a fifth
    //      of the scores were hits (top-1000 of 100 k rows at sample stride 20; the stride rule of plan_chunk now keeps the hit rate
    //      near 2 %, lrx_search.hip).
    constexpr int WCAP = 512;                                   // hits a wave can park in LDS; more go straight to the global lists
    unsigned long long* wl = (unsigned long long*)(smem + wave * (WCAP * 12 + 64));
    unsigned int* wq = (unsigned int*)(wl + WCAP);
    unsigned int* wn = (unsigned int*)(smem + 8 * HALF_BYTES + 2048) + wave;   // the wave's hit count        (the three of them behind the ring:
    unsigned int* qc = (unsigned int*)(smem + 8 * HALF_BYTES + 1024);          // hits of the workgroup per query column; then their global base
    const float* xthr = (const float*)(smem + 8 * HALF_BYTES);                  // thresholds of queries n0 .. n0 + 255: written at kernel start)
    f32x4 t16[2][2];
#pragma unroll
    for (int hp = 0; hp < 2; ++hp)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) t16[hp][ni] = *(const f32x4*)(xthr + hp * 128 + wc * 32 + ni * 16 + fq * 4);
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        const int64_t m = (int64_t)m0 + h * 128 + wr * 64 + mi * 16 + fr;
        const bool mok = m < M;
#pragma unroll
        for (int hp = 0; hp < 2; ++hp)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) {
            const f32x4 v = acc[h][hp][mi][ni];
            unsigned int c = 0;
#pragma unroll
            for (int r = 0; r < 4; ++r) c += (mok && v[r] >= t16[hp][ni][r]) ? 1u : 0u;
            if (c) {
              unsigned int p = atomicAdd(wn, c);                // LDS
#pragma unroll
              for (int r = 0; r < 4; ++r)
                if (mok && v[r] >= t16[hp][ni][r]) {
                  const unsigned int col = hp * 128 + wc * 32 + ni * 16 + fq * 4 + r;
                  const unsigned long long w = sel_pack(f2key(v[r]), m);
                  if (p < WCAP) { wl[p] = w; wq[p] = col | (atomicAdd(&qc[col], 1u) << 8); }   // rank < 256 rows of the tile
                  else {                                        // list full (a tile of near-duplicates): straight to the query's list
                    const unsigned int gp = atomicAdd(&em.cnt[(n0 + col) * CNT_STRIDE], 1u);
                    if (gp < em.cap) em.cand[(int64_t)(n0 + col) * em.cap + gp] = w;
                  }
                  ++p;
                }
            }
          }
      }
    __syncthreads();                                             // every wave's list and the per-query counts are complete
    if (tid < GBN) {
      const unsigned int n = qc[tid];
      if (n) qc[tid] = atomicAdd(&em.cnt[(n0 + tid) * CNT_STRIDE], n);  // the count may pass cap: that IS the overflow signal downstream
    }
    __syncthreads();
    const unsigned int total = min(*wn, (unsigned int)WCAP);
    for (unsigned int i = lane; i < total; i += 64) {
      const unsigned long long w = wl[i];
      const unsigned int e = wq[i], col = e & 255u;
      const unsigned int gp = qc[col] + (e >> 8);
      if (gp < em.cap) em.cand[(int64_t)(n0 + col) * em.cap + gp] = w;
    }
    return;
  }
  if constexpr (EPI == EPI_RESID32) {
    // ---- precise residual stream: x32[m, n] += acc (fp32 in, fp32 out, ONE rounding: to fp32), a16[m, n] = bf16(x32 * gamma[n]) = the
    //      next projection's bf16 A operand (the RMSNorm weight rides on the operand, the weights stay exact; the row scale is applied to
    //      the consumer's accumulator as in the folded path), ss_part = sum of squares of the fp32 row.  `resid` = x32 (in/out), C = a16
    //      (may be NULL), `bias` = gamma (bf16 [N], NULL = 1).  The fp32 tile is staged through LDS in two 128-row passes (128 KiB each):
    //      16-B chunk index XOR ((row & 15) << 2) keeps the accumulator-layout writes and the row-major reads conflict-free; one wave
    //      instruction of the read-out covers one whole 1-KiB row segment (x32 read + write) and 512 B of a16.
    float* x32 = (float*)resid;
    const int nq4 = n0 + lane * 4;                       // this lane's four columns in every read-out iteration
    const bool nin = nq4 < N;
    float gm[4] = {1.f, 1.f, 1.f, 1.f};
    if (bias != nullptr && nin) {
      const bf16x4 g4 = *(const bf16x4*)(bias + nq4);
#pragma unroll
      for (int e = 0; e < 4; ++e) gm[e] = bf2f(g4[e]);
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      f32x4 rv[16];
#pragma unroll
      for (int it = 0; it < 16; ++it) {
        const int m = m0 + p * 128 + it * 8 + wave;
        rv[it] = (m < M && nin) ? *(const f32x4*)(x32 + (int64_t)m * N + nq4) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
      if (p) __syncthreads();                            // the previous pass has been read out
#pragma unroll
      for (int hp = 0; hp < 2; ++hp)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) {
            const int row = wr * 64 + mi * 16 + fr, chunk = hp * 32 + wc * 8 + ni * 4 + fq;
            *(f32x4*)(smem + row * 1024 + ((chunk ^ ((row & 15) << 2)) << 4)) = acc[p][hp][mi][ni];
          }
      __syncthreads();
#pragma unroll
      for (int it = 0; it < 16; ++it) {
        const int row = it * 8 + wave, m = m0 + p * 128 + row;
        f32x4 v = *(const f32x4*)(smem + row * 1024 + ((lane ^ ((row & 15) << 2)) << 4));
        const bool inb = m < M && nin;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += rv[it][e];
        if (inb) {
          *(f32x4*)(x32 + (int64_t)m * N + nq4) = v;
          if (C != nullptr) {
            bf16x4 a4;
            if constexpr (OUTH) {
              bool sat = false;
#pragma unroll
              for (int e = 0; e < 4; ++e) { const float t = v[e] * gm[e]; sat |= !(fabsf(t) <= 65504.f); a4[e] = f2h_bits(t); }
              if (__any(sat)) { if ((int)(threadIdx.x & 63) == __ffsll((long long)__ballot(1)) - 1) atomicAdd(&g_qkv_fp16_saturations, 1u); }
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e) a4[e] = f2bf(v[e] * gm[e]);
            }
            *(bf16x4*)(C + (int64_t)m * N + nq4) = a4;
          }
        }
        if (nrm.ss_part != nullptr) {
          // sum of squares of the row's 256 columns: fixed order inside the lane, then a fixed DPP tree over the wave (rotations
          // inside the 16-lane rows, row_bcast:15 / :31 across them): lane 63 holds the total, the same bits in every run
          float ssq = inb ? (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]) : 0.f;
          ssq += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(ssq), 0x128, 0xF, 0xF, true));   // row_ror:8
          ssq += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(ssq), 0x124, 0xF, 0xF, true));   // row_ror:4
          ssq += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(ssq), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
          ssq += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(ssq), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
          ssq += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(ssq), 0x142, 0xA, 0xF, false));  // row_bcast:15 -> rows 1, 3
          ssq += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(ssq), 0x143, 0xC, 0xF, false));  // row_bcast:31 -> rows 2, 3
          if (lane == 63 && m < M) nrm.ss_part[(int64_t)tn * M + m] = ssq;
        }
      }
    }
    return;
  }
  constexpr int CW = (EPI == EPI_SWIGLU) ? 128 : 256;  // output columns of this tile
  constexpr int CPR = CW / 8;                          // 16-B chunks per staged row
  constexpr int NIT = (256 * CPR) / 512;
  bf16x8 rv[EPI == EPI_RESID ? NIT : 1];
  if (EPI == EPI_RESID) {
    // all residual loads of this thread are issued before anything is staged: the operand fragments of the K loop are dead, so their
    // 64 registers hold the 16 loads while the accumulators are converted and written to LDS (the tile's 128 KiB of residual then
    // arrive under the staging + barrier instead of after them: the timeline showed 6.0 us of a 60 us tile waiting here)
    const int ldc_ = N, c0_ = n0;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int q = it * 512 + tid;
      const int m = min(m0 + q / CPR, M - 1), n = min(c0_ + (q % CPR) * 8, ldc_ - 8);
      rv[it] = *(const bf16x8*)(resid + (int64_t)m * N + n);
    }
  }
  // all LDS reads of the K loop are complete (lgkmcnt(0) precedes every barrier; the last barrier was passed by all waves)
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      const int row = h * 128 + wr * 64 + mi * 16 + fr;
      const float rs = rsv[h][mi];
#pragma unroll
      for (int hp = 0; hp < 2; ++hp) {
        if (EPI == EPI_SWIGLU) {
          // wave columns [nb, nb+16) = gate, [nb+16, nb+32) = up of output columns nb/2 .. nb/2+15
          bf16x4 o;
          bool sat = false;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            // silu(g) * u with the two raw transcendentals (v_exp_f32, v_rcp_f32: 1 ulp, far below the 16-bit rounding that follows);
            // the IEEE division sequence of `g / (1 + expf(-g))` cost ~2 us per 256x256 tile (timeline, tools/gemm_timeline.py)
            const float g = acc[h][hp][mi][0][r] * rs, u = acc[h][hp][mi][1][r] * rs;
            const float sg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(g * -1.4426950408889634f));
            const float t = g * sg * u;
            if constexpr (F16) { sat |= !(fabsf(t) <= 65504.f); o[r] = f2h_bits(t); }
            else o[r] = f2bf(t);
          }
          if constexpr (F16) { if (__any(sat)) { if ((int)(threadIdx.x & 63) == __ffsll((long long)__ballot(1)) - 1) atomicAdd(&g_qkv_fp16_saturations, 1u); } }
          const int col = hp * 64 + wc * 16 + fq * 4;  // within the 128-column output tile
          *(bf16x4*)(smem + row * (CW * 2) + ((((col >> 3) ^ (row & 15)) << 4) | ((col & 4) << 1))) = o;
        } else if (EPI == EPI_ROPE) {
          // the lane's rotary pair of this 32-column block: x1 = block columns fq*4 + r (logical j), x2 = 16 + fq*4 + r (logical j + d/2)
          const int cb = hp * 128 + wc * 32;                       // block's first column inside the tile
          const bool rot = n0 + cb < rope.rope_cols;               // (wave-uniform: q|k blocks rotate, v blocks do not)
          f32x4 x1 = acc[h][hp][mi][0], x2 = acc[h][hp][mi][1];
#pragma unroll
          for (int r = 0; r < 4; ++r) { x1[r] *= rs; x2[r] *= rs; }
          if (bias != nullptr) {
            const int nb = min(n0 + cb + fq * 4, N - 20);
            const bf16x4 b1 = *(const bf16x4*)(bias + nb), b2 = *(const bf16x4*)(bias + nb + 16);
#pragma unroll
            for (int r = 0; r < 4; ++r) { x1[r] += bf2f(b1[r]); x2[r] += bf2f(b2[r]); }
          }
          if (rot) {
            const int rhalf = rope.head_dim >> 1;
            const int j = (((n0 + cb) % rope.head_dim) >> 5) * 16 + fq * 4;
            const f32x4 cc = *(const f32x4*)(rope.cos + (int64_t)rposv[h][mi] * rhalf + j), sn = *(const f32x4*)(rope.sin + (int64_t)rposv[h][mi] * rhalf + j);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float a = x1[r], b = x2[r];
              x1[r] = a * cc[r] - b * sn[r];
              x2[r] = b * cc[r] + a * sn[r];
            }
          }
          bf16x4 o1, o2;
          bool sat = false;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            sat |= !(fabsf(x1[r]) <= 65504.f) | !(fabsf(x2[r]) <= 65504.f);    // (NaN fails the comparison too)
            o1[r] = f2h_bits(x1[r]);
            o2[r] = f2h_bits(x2[r]);
          }
          // one atomic per wave INSTRUCTION that saw any (the wave votes, its first lane reports): on a healthy checkpoint none at all, on
          // a broken one bounded contention on the counter instead of one atomic per lane
          if (__any(sat)) { if ((int)(threadIdx.x & 63) == __ffsll((long long)__ballot(1)) - 1) atomicAdd(&g_qkv_fp16_saturations, 1u); }
          const int c1 = cb + fq * 4, c2 = c1 + 16;
          *(bf16x4*)(smem + row * (CW * 2) + ((((c1 >> 3) ^ (row & 15)) << 4) | ((c1 & 4) << 1))) = o1;
          *(bf16x4*)(smem + row * (CW * 2) + ((((c2 >> 3) ^ (row & 15)) << 4) | ((c2 & 4) << 1))) = o2;
        } else {
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) {
            const int col = hp * 128 + wc * 32 + ni * 16 + fq * 4;
            f32x4 v = acc[h][hp][mi][ni];
            if (EPI == EPI_STORE) {
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] *= rs;
            }
            if (EPI == EPI_STORE && bias != nullptr) {
              const int n = min(n0 + col, N - 4);
              bf16x4 bv = *(const bf16x4*)(bias + n);
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] += bf2f(bv[r]);
            }
            if (EPI == EPI_MAXAGG && bias != nullptr) {    // N is arbitrary here: element-wise clamped reads
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] += bf2f(bias[min(n0 + col + r, N - 1)]);
            }
            bf16x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = f2bf(v[r]);
            *(bf16x4*)(smem + row * (CW * 2) + ((((col >> 3) ^ (row & 15)) << 4) | ((col & 4) << 1))) = o;
          }
        }
      }
    }
  __syncthreads();   // (also waits for the residual rows; measured in round 3: an LDS-only wait here, the loads left in flight, is 1 % slower on the O-projection)
  if (EPI == EPI_MAXAGG) {
    // Segmented column maximum of the staged bf16 logits tile (utils/max_linear_map.py:8-88 without the [B,S,V] tensor):
    // wave w owns columns [32w, 32w+32); a lane walks rows (lane>>3) + 8*step holding 4 columns.  Rows map to output rows
    // through row_seg (non-decreasing where >= 0); a lane flushes its 4 maxima with integer atomics whenever its segment
    // changes, and at the end the 8 row-lanes of a column group are combined first when the whole wave ended in one segment
    // (the common case: a 256-row tile inside one document -> 256 atomics per tile).
    const int colq = wave * 32 + (lane & 7) * 4;
    const int choff = colq >> 3, hb = (colq & 4) << 1;
    const float NEG = -__builtin_inff();
    float m4[4] = {NEG, NEG, NEG, NEG};
    int cur = -1;
    auto flush = [&](int seg) {
      if (seg < 0) return;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (n0 + colq + r < N && m4[r] != NEG) atomic_fmax_bits(mx.out + (int64_t)seg * mx.ldo + n0 + colq + r, m4[r]);
    };
    for (int step = 0; step < 32; ++step) {
      const int row = step * 8 + (lane >> 3);
      const int m = m0 + row;
      const int seg = m < M ? mx.row_seg[m] : -1;
      if (seg >= 0) {
        if (seg != cur) {
          flush(cur);
          cur = seg;
#pragma unroll
          for (int r = 0; r < 4; ++r) m4[r] = NEG;
        }
        const bf16x4 v = *(const bf16x4*)(smem + row * 512 + (((choff ^ (row & 15)) << 4) | hb));
#pragma unroll
        for (int r = 0; r < 4; ++r) m4[r] = fmaxf(m4[r], bf2f(v[r]));
      }
    }
    int cmax = cur;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) cmax = max(cmax, __shfl_xor(cmax, o));
    const bool uniform = __all(cur == cmax || cur == -1);
    if (uniform) {
      if (cur == -1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) m4[r] = NEG;
      }
#pragma unroll
      for (int o = 8; o <= 32; o <<= 1)
#pragma unroll
        for (int r = 0; r < 4; ++r) m4[r] = fmaxf(m4[r], __shfl_xor(m4[r], o));
      if ((lane >> 3) == 0) flush(cmax);
    } else {
      flush(cur);
    }
    return;
  }
  const int ncols = (EPI == EPI_SWIGLU) ? (N >> 1) : N;                 // columns of C this launch produces
  const int ldc = (EPI == EPI_ROPE && rope.ldc > 0) ? rope.ldc : ncols;  // ... inside rows of this stride
  const int c0 = (EPI == EPI_SWIGLU) ? (n0 >> 1) : n0;
  {
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int q = it * 512 + tid;
    const int row = q / CPR, ch = q % CPR;
    const int m = m0 + row, n = c0 + ch * 8;
    const bool inb = m < M && n < ncols;
    if (!(EPI == EPI_RESID && nrm.ss_part != nullptr) && !inb) continue;   // (with ss_part every lane stays for the row reduction)
    bf16x8 v = *(const bf16x8*)(smem + row * (CW * 2) + ((ch ^ (row & 15)) << 4));
    if (EPI == EPI_RESID) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = f2bf(bf2f(v[e]) + bf2f(rv[it][e]));
      if (nrm.ss_part != nullptr) {
        // sum of squares of the bf16 values just produced: 8 elements in order, then a fixed xor tree over the row's 32 chunks
        float ssq = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float f = bf2f(v[e]); ssq += f * f; }
        if (!inb) ssq = 0.f;
        // all-reduce over the row's 32 lanes: rotations inside the 16-lane DPP rows (row_ror 8, 4, then the two quad permutes), one
        // cross-row exchange -- the same order in every lane and every run (five ds_bpermute per chunk cost ~4 us per tile)
        ssq += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(ssq), 0x128, 0xF, 0xF, true));   // row_ror:8
        ssq += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(ssq), 0x124, 0xF, 0xF, true));   // row_ror:4
        ssq += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(ssq), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
        ssq += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(ssq), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
        ssq += __shfl_xor(ssq, 16, 64);
        if ((tid & 31) == 0 && m < M) nrm.ss_part[(int64_t)tn * M + m] = ssq;
        if (!inb) continue;
      }
    }
    *(bf16x8*)(C + (int64_t)m * ldc + n) = v;
  }
  }
}

// m-tiles per group of the block -> tile map, per epilogue class (measured, see the kernel); LRX_GEMM_GM overrides it for sweeps
static int gemm_group_m(int epilogue, int K) {
  static const int env = lrx_dev_knob("LRX_GEMM_GM", 0);   // (dev builds only; thread-safe one-time read)
  if (env > 0) return env;
  // 1B shapes (K = 2048): gate-up 6, o/down 4, qkv 8; 8B shapes (K = 4096): within 2 % for 2..8, gate-up best at 8 (1596 TFLOP/s).
  // Round 4 sweep (tools/exp/gm_sweep.sh, M = 131 072): qkv 1.31-1.32 ms at 0 / 4 / 8 (1.34-1.59 elsewhere); down (K = 8192) 2.94 at 2 / 4, 3.03 at 6 / 8;
  // o (N = K = 2048) 0.882-0.887 at 3 / 6 / 12 against 0.901-0.906 at 2 / 4 / 8; 8B o (N = K = 4096, fp32 stream) 1.798 at 6 against 1.82-1.84 at 2 / 4 / 8,
  // 8B down (K = 14336) 5.36-5.37 at 2 / 4 against 5.49-5.51 at 6 / 8 (tools/exp/gm_sweep_8b.sh) -> the residual GEMMs take 6 for K <= 4096, 4 above.
  return epilogue == EPI_SWIGLU ? (K >= 4096 ? 8 : 6) : (epilogue == EPI_RESID ? (K <= 4096 ? 6 : 4) : 8);
}

extern "C" int lrx_gemm_bf16_nt(const void* A, const void* B, void* C, const void* bias, const void* resid, int32_t M, int32_t N,
                                int32_t K, int32_t epilogue, void* stream) {
  return lrx_gemm_bf16_nt_fused(A, B, C, bias, resid, M, N, K, epilogue, nullptr, nullptr, stream);
}

extern "C" int lrx_gemm_bf16_nt_fused(const void* A, const void* B, void* C, const void* bias, const void* resid, int32_t M, int32_t N,
                                      int32_t K, int32_t epilogue, const float* rscale, float* ss_part, void* stream) {
  return lrx_gemm_nt_fused_ex(A, B, C, bias, resid, M, N, K, epilogue, rscale, ss_part, 0, stream);
}

// (f16 != 0: fp16 operands, SwiGLU epilogue only -- the gate-up projection of precise_stream = 2; the output is fp16 then)
int lrx_gemm_nt_fused_ex(const void* A, const void* B, void* C, const void* bias, const void* resid, int32_t M, int32_t N, int32_t K, int32_t epilogue,
                         const float* rscale, float* ss_part, int f16, void* stream) {
  LRX_CHECK_ARG(M >= 0 && N > 0 && K > 0, "gemm: bad shape M=%d N=%d K=%d", M, N, K);
  LRX_CHECK_ARG(!f16 || epilogue == EPI_SWIGLU, "gemm: fp16 operands are served for the SwiGLU epilogue (epilogue %d)", epilogue);
  LRX_CHECK_ARG(K % GBK == 0, "gemm: K=%d must be a multiple of %d", K, GBK);
  LRX_CHECK_ARG(N % 8 == 0, "gemm: N=%d must be a multiple of 8", N);
  LRX_CHECK_ARG(epilogue >= 0 && epilogue <= 2, "gemm: unknown epilogue %d", epilogue);
  LRX_CHECK_ARG(epilogue != EPI_SWIGLU || N % 32 == 0, "gemm: SwiGLU epilogue needs N %% 32 == 0 (N=%d)", N);
  LRX_CHECK_ARG(epilogue != EPI_RESID || resid != nullptr, "gemm: residual epilogue without resid");
  LRX_CHECK_ARG(ss_part == nullptr || epilogue == EPI_RESID, "gemm: ss_part belongs to the residual epilogue");
  LRX_CHECK_ARG(rscale == nullptr || epilogue != EPI_RESID, "gemm: rscale applies to the store / SwiGLU epilogues");
  if (M == 0) return LRX_OK;
  int tiles_m = (int)lrx_cdiv(M, GBM), tiles_n = (int)lrx_cdiv(N, GBN);
  dim3 grid(tiles_m * tiles_n), block(512);
  hipStream_t s = (hipStream_t)stream;
  const __bf16 *a = (const __bf16*)A, *b = (const __bf16*)B, *bi = (const __bf16*)bias, *re = (const __bf16*)resid;
  __bf16* c = (__bf16*)C;
  RopeArgs none = {nullptr, nullptr, nullptr, 0, 64, 0};
  MaxAggArgs nomx = {nullptr, nullptr, 0};
  NormArgs nrm = {rscale, ss_part, gemm_group_m(epilogue, K)};
  switch (epilogue) {
    case EPI_STORE: hipLaunchKernelGGL(k_gemm_bf16_nt<EPI_STORE>, grid, block, 0, s, a, b, c, bi, re, M, N, K, tiles_m, tiles_n, none, nomx, nrm, EmitArgs{nullptr, nullptr, nullptr, 0, 0u, nullptr, nullptr, 0, 0}); break;
    case EPI_RESID: hipLaunchKernelGGL(k_gemm_bf16_nt<EPI_RESID>, grid, block, 0, s, a, b, c, bi, re, M, N, K, tiles_m, tiles_n, none, nomx, nrm, EmitArgs{nullptr, nullptr, nullptr, 0, 0u, nullptr, nullptr, 0, 0}); break;
    default:
      if (f16) hipLaunchKernelGGL((k_gemm_bf16_nt<EPI_SWIGLU, true>), grid, block, 0, s, a, b, c, bi, re, M, N, K, tiles_m, tiles_n, none, nomx, nrm, EmitArgs{nullptr, nullptr, nullptr, 0, 0u, nullptr, nullptr, 0, 0});
      else hipLaunchKernelGGL(k_gemm_bf16_nt<EPI_SWIGLU>, grid, block, 0, s, a, b, c, bi, re, M, N, K, tiles_m, tiles_n, none, nomx, nrm, EmitArgs{nullptr, nullptr, nullptr, 0, 0u, nullptr, nullptr, 0, 0});
      break;
  }
  LRX_LAUNCH_CHECK();
  return LRX_OK;
}

extern "C" int lrx_gemm_qkv_rope(const void* A, const void* Wqkv, void* C, const void* bias, const int32_t* positions, const float* cos,
                                 const float* sin, int32_t M, int32_t K, int32_t num_q_heads, int32_t num_kv_heads, int32_t head_dim,
                                 void* stream) {
  return lrx_gemm_qkv_rope_fused(A, Wqkv, C, bias, positions, cos, sin, M, K, num_q_heads, num_kv_heads, head_dim, nullptr, stream);
}

extern "C" int lrx_gemm_qkv_rope_fused(const void* A, const void* Wqkv, void* C, const void* bias, const int32_t* positions, const float* cos,
                                       const float* sin, int32_t M, int32_t K, int32_t num_q_heads, int32_t num_kv_heads, int32_t head_dim,
                                       const float* rscale, void* stream) {
  return lrx_gemm_qkv_rope_slice(A, Wqkv, C, bias, positions, cos, sin, M, K, num_q_heads, num_kv_heads, head_dim, rscale, 0,
                                 num_q_heads + 2 * num_kv_heads, stream);
}

// Heads [head0, head0 + n_heads) of the fused q|k|v projection (heads numbered q_0 .. q_{nq-1}, k_0 .. k_{nkv-1}, v_0 .. v_{nkv-1}): the same
// kernel over that row slice of Wqkv / bias, writing that column slice of C (whose rows keep the full q|k|v width).  The encoder's FINAL
// layer needs k|v of every token but q of the pooled (last) tokens only: lrx_encode_packed runs the k|v slice over all rows and the q slice
// over the gathered last rows (two thirds of that layer's projection FLOPs are not computed).
extern "C" int lrx_gemm_qkv_rope_slice(const void* A, const void* Wqkv, void* C, const void* bias, const int32_t* positions, const float* cos,
                                       const float* sin, int32_t M, int32_t K, int32_t num_q_heads, int32_t num_kv_heads, int32_t head_dim,
                                       const float* rscale, int32_t head0, int32_t n_heads, void* stream) {
  return lrx_gemm_qkv_rope_slice_ex(A, Wqkv, C, bias, positions, cos, sin, M, K, num_q_heads, num_kv_heads, head_dim, rscale, head0, n_heads, 0, stream);
}

// (f16 != 0: A and Wqkv hold fp16 values -- precise_stream = 2; the bias stays bf16, the output is fp16 either way)
int lrx_gemm_qkv_rope_slice_ex(const void* A, const void* Wqkv, void* C, const void* bias, const int32_t* positions, const float* cos, const float* sin,
                               int32_t M, int32_t K, int32_t num_q_heads, int32_t num_kv_heads, int32_t head_dim, const float* rscale, int32_t head0,
                               int32_t n_heads, int f16, void* stream) {
  const int n_all = num_q_heads + 2 * num_kv_heads;
  LRX_CHECK_ARG(M >= 0 && K > 0 && K % GBK == 0, "gemm_qkv_rope: bad shape M=%d K=%d", M, K);
  LRX_CHECK_ARG(head_dim == 64 || head_dim == 128, "gemm_qkv_rope: head_dim=%d unsupported", head_dim);
  LRX_CHECK_ARG(positions && cos && sin, "gemm_qkv_rope: null rope inputs");
  LRX_CHECK_ARG(head0 >= 0 && n_heads > 0 && head0 + n_heads <= n_all, "gemm_qkv_rope: head slice [%d, %d) outside 0..%d", head0, head0 + n_heads, n_all);
  if (M == 0) return LRX_OK;
  const int N = n_heads * head_dim, col0 = head0 * head_dim;
  const int rope_cols = (num_q_heads + num_kv_heads) * head_dim - col0;          // rotating columns of the slice (<= 0: v heads only)
  int tiles_m = (int)lrx_cdiv(M, GBM), tiles_n = (int)lrx_cdiv(N, GBN);
  RopeArgs rope = {positions, cos, sin, rope_cols < 0 ? 0 : (rope_cols > N ? N : rope_cols), head_dim, n_all * head_dim};
  if (f16)
    hipLaunchKernelGGL((k_gemm_bf16_nt<EPI_ROPE, true>), dim3(tiles_m * tiles_n), dim3(512), 0, (hipStream_t)stream, (const __bf16*)A,
                       (const __bf16*)Wqkv + (int64_t)col0 * K, (__bf16*)C + col0, bias ? (const __bf16*)bias + col0 : (const __bf16*)nullptr,
                       (const __bf16*)nullptr, M, N, K, tiles_m, tiles_n, rope, MaxAggArgs{nullptr, nullptr, 0}, NormArgs{rscale, nullptr, 8},
                       EmitArgs{nullptr, nullptr, nullptr, 0, 0u, nullptr, nullptr, 0, 0});
  else
  hipLaunchKernelGGL(k_gemm_bf16_nt<EPI_ROPE>, dim3(tiles_m * tiles_n), dim3(512), 0, (hipStream_t)stream, (const __bf16*)A,
                     (const __bf16*)Wqkv + (int64_t)col0 * K, (__bf16*)C + col0, bias ? (const __bf16*)bias + col0 : (const __bf16*)nullptr,
                     (const __bf16*)nullptr, M, N, K, tiles_m, tiles_n, rope, MaxAggArgs{nullptr, nullptr, 0}, NormArgs{rscale, nullptr, 8},
                     EmitArgs{nullptr, nullptr, nullptr, 0, 0u, nullptr, nullptr, 0, 0});
  LRX_LAUNCH_CHECK();
  return LRX_OK;
}

// x32[M, N] (fp32, in place) += A[M, K] . B[N, K]^T; a16_out[M, N] (bf16, optional) = bf16(x32 * gamma[n]) (gamma bf16 [N], NULL = 1);
// ss_part (optional) [ceil(N / 256), M]: sum of squares of the new fp32 row per 256-column tile.  The precise residual stream of deep
// backbones (lrx_encoder_config.precise_stream).
extern "C" int lrx_gemm_bf16_nt_resid32(const void* A, const void* B, float* x32, void* a16_out, const void* gamma, int32_t M, int32_t N, int32_t K,
                                        float* ss_part, void* stream) {
  return lrx_gemm_nt_resid32_ex(A, B, x32, a16_out, gamma, M, N, K, ss_part, 0, 0, stream);
}

// (f16 != 0: A and B hold fp16 values -- precise_stream = 2; out_f16 != 0: a16_out = fp16(x32 * gamma) -- always with f16, and for the GEMM that feeds the
// QKV projection under precise_stream = 3; gamma stays bf16)
int lrx_gemm_nt_resid32_ex(const void* A, const void* B, float* x32, void* a16_out, const void* gamma, int32_t M, int32_t N, int32_t K, float* ss_part,
                           int f16, int out_f16, void* stream) {
  LRX_CHECK_ARG(!f16 || out_f16 || a16_out == nullptr, "gemm_resid32: fp16 operands write an fp16 operand");
  LRX_CHECK_ARG(M >= 0 && N > 0 && K > 0 && K % GBK == 0 && N % 8 == 0, "gemm_resid32: bad shape M=%d N=%d K=%d", M, N, K);
  LRX_CHECK_ARG(x32 != nullptr, "gemm_resid32: null residual stream");
  if (M == 0) return LRX_OK;
  int tiles_m = (int)lrx_cdiv(M, GBM), tiles_n = (int)lrx_cdiv(N, GBN);
  RopeArgs none = {nullptr, nullptr, nullptr, 0, 64, 0};
  if (f16)
    hipLaunchKernelGGL((k_gemm_bf16_nt<EPI_RESID32, true>), dim3(tiles_m * tiles_n), dim3(512), 0, (hipStream_t)stream, (const __bf16*)A, (const __bf16*)B,
                       (__bf16*)a16_out, (const __bf16*)gamma, (const __bf16*)x32, M, N, K, tiles_m, tiles_n, none, MaxAggArgs{nullptr, nullptr, 0},
                       NormArgs{nullptr, ss_part, gemm_group_m(EPI_RESID, K)}, EmitArgs{nullptr, nullptr, nullptr, 0, 0u, nullptr, nullptr, 0, 0});
  else if (out_f16 && a16_out != nullptr)
    hipLaunchKernelGGL((k_gemm_bf16_nt<EPI_RESID32, false, true>), dim3(tiles_m * tiles_n), dim3(512), 0, (hipStream_t)stream, (const __bf16*)A, (const __bf16*)B,
                       (__bf16*)a16_out, (const __bf16*)gamma, (const __bf16*)x32, M, N, K, tiles_m, tiles_n, none, MaxAggArgs{nullptr, nullptr, 0},
                       NormArgs{nullptr, ss_part, gemm_group_m(EPI_RESID, K)}, EmitArgs{nullptr, nullptr, nullptr, 0, 0u, nullptr, nullptr, 0, 0});
  else
  hipLaunchKernelGGL(k_gemm_bf16_nt<EPI_RESID32>, dim3(tiles_m * tiles_n), dim3(512), 0, (hipStream_t)stream, (const __bf16*)A, (const __bf16*)B,
                     (__bf16*)a16_out, (const __bf16*)gamma, (const __bf16*)x32, M, N, K, tiles_m, tiles_n, none, MaxAggArgs{nullptr, nullptr, 0},
                     NormArgs{nullptr, ss_part, gemm_group_m(EPI_RESID, K)}, EmitArgs{nullptr, nullptr, nullptr, 0, 0u, nullptr, nullptr, 0, 0});
  LRX_LAUNCH_CHECK();
  return LRX_OK;
}

#ifndef LRX_MAXAGG_GM
#define LRX_MAXAGG_GM 8
#endif
// out[row_seg[m], n] = max(out[row_seg[m], n], bf16(A[m,:] . B[n,:] + bias[n])) over the rows with row_seg >= 0 (lrx_sparse.hip)
int lrx_gemm_max_aggregate_launch(const void* A, const void* B, const void* bias, const int32_t* row_seg, float* out, int64_t ldo, int M, int N,
                                  int K, hipStream_t stream) {
  LRX_CHECK_ARG(M >= 0 && N > 0 && K > 0 && K % GBK == 0, "max_aggregate: bad shape M=%d N=%d K=%d (K must be a multiple of %d)", M, N, K, GBK);
  LRX_CHECK_ARG(row_seg && out && ldo >= N, "max_aggregate: bad output spec");
  if (M == 0) return LRX_OK;
  int tiles_m = (int)lrx_cdiv(M, GBM), tiles_n = (int)lrx_cdiv(N, GBN);
  RopeArgs none = {nullptr, nullptr, nullptr, 0, 64, 0};
  MaxAggArgs mx = {row_seg, out, ldo};
  static const int maxagg_gm = lrx_dev_knob("LRX_MAXAGG_GM", 0) > 0 ? lrx_dev_knob("LRX_MAXAGG_GM", 0) : LRX_MAXAGG_GM;   // (dev builds: tools/exp/maxagg_gm_sweep.sh)
  hipLaunchKernelGGL(k_gemm_bf16_nt<EPI_MAXAGG>, dim3(tiles_m * tiles_n), dim3(512), 0, stream, (const __bf16*)A, (const __bf16*)B, (__bf16*)nullptr,
                     (const __bf16*)bias, (const __bf16*)nullptr, M, N, K, tiles_m, tiles_n, none, mx, NormArgs{nullptr, nullptr, maxagg_gm},
                     EmitArgs{nullptr, nullptr, nullptr, 0, 0u, nullptr, nullptr, 0, 0});
  LRX_LAUNCH_CHECK();
  return LRX_OK;
}

// The filter pass of the bounded search for more than 128 queries (lrx_search.hip): the shard's tiled fp16 shadow against the fp16 queries
// on the GEMM kernel -- its 256 x 256 tile stages a query k-slice once per 256 rows and its 4-phase K loop keeps the LDS-DMA ahead of
// the MFMAs with one workgroup per CU; the 128-row filter kernel re-stages the 32-KiB query slice for every 128 rows and is bound
// by that L2 -> LDS traffic at 16 query tiles (1.19 ms for 256 queries over 1M x 2048; HBM floor 0.75).
// Round 6 -- wide chunks: up to LRX_EMIT_MAX_QUERIES queries = up to four 256-query n-tiles per streamed 256-row A tile.  The block -> tile
// map walks groups of GM m-tiles x all n-tiles with GM x n-tiles ~ 32 = the workgroups one XCD runs at a time: the n-tiles of an A tile
// run side by side on one XCD and find it in that XCD's L2, so the shadow is read from HBM once per pass whatever the query count
// (1000 queries over 1M x 2048: 4.1 GB instead of 4 x 4.1 GB; the pass is then MFMA-bound like the encoder's GEMMs).
int lrx_gemm_filter_emit_launch(const void* Xb, const void* q16, int64_t n_rows, int nq, int dim, int ss, int64_t n_tiles,
                                const float* thr, unsigned long long* cand, unsigned int* cnt, unsigned int cap, hipStream_t stream) {
  LRX_CHECK_ARG(dim > 0 && dim % GBK == 0 && nq > 0 && nq <= LRX_EMIT_MAX_QUERIES && n_rows < (1ll << 31), "filter_emit: bad shape rows=%lld nq=%d dim=%d",
                (long long)n_rows, nq, dim);
  if (n_tiles <= 0) return LRX_OK;
  const int tiles_n = (nq + GBN - 1) / GBN;
  LRX_CHECK_ARG(n_tiles * tiles_n < (1ll << 31), "filter_emit: %lld tiles", (long long)(n_tiles * tiles_n));
  static const int gm_env = lrx_dev_knob("LRX_EMIT_GM", 0);
  const int gm = gm_env > 0 ? gm_env : (tiles_n == 1 ? 8 : (32 + tiles_n - 1) / tiles_n);
  RopeArgs none = {nullptr, nullptr, nullptr, 0, 64, 0};
  hipLaunchKernelGGL(k_gemm_bf16_nt<EPI_EMIT>, dim3((unsigned)(n_tiles * tiles_n)), dim3(512), 0, stream, (const __bf16*)Xb, (const __bf16*)q16, (__bf16*)nullptr,
                     (const __bf16*)nullptr, (const __bf16*)nullptr, (int)n_rows, nq, dim, (int)n_tiles, tiles_n, none, MaxAggArgs{nullptr, nullptr, 0},
                     NormArgs{nullptr, nullptr, gm}, EmitArgs{thr, cand, cnt, ss, cap, nullptr, nullptr, 0, 0});
  LRX_LAUNCH_CHECK();
  return LRX_OK;
}

// The SAMPLE pass of the same search on the same kernel (round 6): every ss-th 256-row tile of the shadow against all queries, scores and
// 16-row-group maxima stored for k_sample_threshold.  (Rounds 2-5 ran it on the 128-row register-streaming kernel, which at 16 query tiles
// spends 170 us on ONE block per workgroup -- 0.68 of the 4.0 ms of a 1000-query search went there.)
int lrx_gemm_filter_sample_launch(const void* Xb, const void* q16, int64_t n_rows, int nq, int dim, int ss, int64_t n_tiles, float* scores, float* gmax,
                                  int64_t ld_s, int nblk_ld_s, hipStream_t stream) {
  LRX_CHECK_ARG(dim > 0 && dim % GBK == 0 && nq > 0 && nq <= LRX_EMIT_MAX_QUERIES && n_rows < (1ll << 31) && ss >= 1 && scores && gmax,
                "filter_sample: bad shape rows=%lld nq=%d dim=%d", (long long)n_rows, nq, dim);
  if (n_tiles <= 0) return LRX_OK;
  const int tiles_n = (nq + GBN - 1) / GBN;
  RopeArgs none = {nullptr, nullptr, nullptr, 0, 64, 0};
  hipLaunchKernelGGL(k_gemm_bf16_nt<EPI_SAMPLE>, dim3((unsigned)(n_tiles * tiles_n)), dim3(512), 0, stream, (const __bf16*)Xb, (const __bf16*)q16, (__bf16*)nullptr,
                     (const __bf16*)nullptr, (const __bf16*)nullptr, (int)n_rows, nq, dim, (int)n_tiles, tiles_n, none, MaxAggArgs{nullptr, nullptr, 0},
                     NormArgs{nullptr, nullptr, tiles_n == 1 ? 8 : (32 + tiles_n - 1) / tiles_n}, EmitArgs{nullptr, nullptr, nullptr, ss, 0u, scores, gmax, ld_s, nblk_ld_s});
  LRX_LAUNCH_CHECK();
  return LRX_OK;
}
