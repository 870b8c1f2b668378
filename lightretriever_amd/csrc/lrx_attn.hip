// Varlen causal GQA attention forward for packed sequences (S <= max_positions), gfx950 MFMA 32x32x16 bf16.
//
// Work item = (sequence, 64-row q tile, kv head).  Workgroup = 2*GRP waves: wave w handles q head (kvh*GRP + w%GRP)
// and the 32-row half (w / GRP) of the q tile, so the K/V tile staged in LDS is shared by all GRP q heads of the group.
//
// Per wave, per 32-key sub-tile:
//   S^T[key, q] = K . Q^T      (A = K rows from LDS via ds_read_b128, B = Q fragments kept in registers)
//     -> the score column of one q row lives in ONE lane (16 regs) + its lane^32 partner: row max / row sum are
//        15 VALU ops + one cross-half shuffle, no LDS.
//   P^T (bf16) is exactly the B-operand layout of the next MFMA (accumulator tile as operand), so
//   O^T[d, q] += V^T . P^T     (A = V^T fragments read from the row-major V tile with ds_read_b64_tr_b16)
//   online softmax state (m, l) is one scalar per lane.
//
// K/V tiles (64 keys) are staged with 16-byte global_load_lds into two LDS stages; the K image is XOR-swizzled for
// conflict-free ds_read_b128 row reads, the V image for conflict-free transposed reads (swizzle applied on the
// per-lane source address, undone on the read).
#include "lrx_common.h"

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;

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

template <int D, int GRP>
__global__ void __launch_bounds__(128 * GRP)
k_attn_varlen_causal(const __bf16* __restrict__ qkv, const int32_t* __restrict__ cu, int nqt, int nq, int nkv,
                     __bf16* __restrict__ out, float scale_log2, int last_tile_only) {
  using G = AttnGeom<D>;
  constexpr int NW = 2 * GRP;
  constexpr int KS = D / 16;  // k-steps of the QK^T product
  constexpr int DT = D / 32;  // 32-row tiles of O^T
  __shared__ __attribute__((aligned(1024))) char smem[4 * G::TILE_BYTES];  // [stage][K|V]

  const int b = last_tile_only ? blockIdx.x : blockIdx.x / nqt;
  const int hk = blockIdx.y;
  const int s0 = cu[b], len = cu[b + 1] - s0;
  // last_tile_only: one q tile per sequence, the one holding its last token (all the pooled path needs of the last layer)
  const int qt = last_tile_only ? ((len - 1) >> 6) : nqt - 1 - (blockIdx.x - b * nqt);  // else: heavy (late) q tiles first
  const int qtile0 = qt * 64;
  if (qtile0 >= len) return;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int hq = hk * GRP + (wave % GRP);
  const int q0 = qtile0 + (wave / GRP) * 32;
  const bool active = q0 < len;
  const int64_t RS = (int64_t)(nq + 2 * nkv) * D;
  const __bf16* kbase = qkv + (int64_t)s0 * RS + (int64_t)(nq + hk) * D;
  const __bf16* vbase = kbase + (int64_t)nkv * D;

  // ---- Q fragments (B operand): Q[q0 + r][16 ks + 8 h .. +7]
  bf16x8 qf[KS];
  {
    const int qrow = min(q0 + r, len - 1);
    const __bf16* qp = qkv + ((int64_t)s0 + qrow) * RS + (int64_t)hq * D + h * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = *(const bf16x8*)(qp + ks * 16);
  }

  // ---- staging: instruction j (0..INSTS-1) of a tile fills LDS bytes [j*1024, j*1024+1024): slot s = j*64 + lane,
  //      row = s / CH, chunk position cs = s % CH, holding logical chunk cs ^ x(row)
  auto stage = [&](int st, int kt) {
    char* sK = smem + st * (2 * G::TILE_BYTES);
    char* sV = sK + G::TILE_BYTES;
    for (int j = wave; j < G::INSTS; j += NW) {
      int s = j * 64 + lane;
      int row = s / G::CH, cs = s % G::CH;
      int grow = min(kt * 64 + row, len - 1);
      const __bf16* kp = kbase + (int64_t)grow * RS + ((cs ^ G::xk(row)) << 3);
      const __bf16* vp = vbase + (int64_t)grow * RS + ((cs ^ G::xv(row)) << 3);
      __builtin_amdgcn_global_load_lds((gptr_t)kp, (lptr_t)(sK + j * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)vp, (lptr_t)(sV + j * 1024), 16, 0, 0);
    }
  };

  // ---- lane-constant LDS read offsets
  int koff[KS];  // K row read: row (sub*32 + r), logical chunk 2ks + h
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) koff[ks] = r * G::ROW_BYTES + (((2 * ks + h) ^ G::xk(r)) << 4);
  // V transposed read: 16-lane group g = lane>>4, i = lane&15, qd = i>>2, p = i&3; block row = key_base + qd,
  // columns dt*32 + 16*(g&1) + 4p .. +3  ->  logical chunk dt*4 + 2*(g&1) + (p>>1), byte 8*(p&1) inside it
  int voff[DT];
  {
    const int g = lane >> 4, i = lane & 15, qd = i >> 2, p = i & 3;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
      voff[dt] = qd * G::ROW_BYTES + (((dt * 4 + 2 * (g & 1) + (p >> 1)) ^ G::xv(qd)) << 4) + 8 * (p & 1);
  }

  f32x16 o[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int t = 0; t < 16; ++t) o[dt][t] = 0.f;
  float m = -1e30f, l = 0.f;

  const int nkt = qt + 1;
  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int kt = 0; kt < nkt; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nkt) stage(cur ^ 1, kt + 1);
    if (active) {
      const char* sK = smem + cur * (2 * G::TILE_BYTES);
      const char* sV = sK + G::TILE_BYTES;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int key0 = kt * 64 + j * 32;
        if (key0 > q0) continue;  // wave-uniform: sub-tile entirely above the diagonal
        // ---- S^T = K Q^T
        f32x16 s;
#pragma unroll
        for (int t = 0; t < 16; ++t) s[t] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          bf16x8 kf = *(const bf16x8*)(sK + j * 32 * G::ROW_BYTES + koff[ks]);
          s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s, 0, 0, 0);
        }
        // ---- scale, causal mask (diagonal sub-tile only), online softmax
        const bool diag = (key0 == q0);
        float mloc = -1e30f;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
          float v = s[t] * scale_log2;
          if (diag) {
            int key = (t & 3) + 8 * (t >> 2) + 4 * h;  // relative to key0 ; q relative to q0 is r
            v = key > r ? -1e30f : v;
          }
          s[t] = v;
          mloc = fmaxf(mloc, v);
        }
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
        const float mnew = fmaxf(m, mloc);
        const float alpha = __builtin_amdgcn_exp2f(m - mnew);
        m = mnew;
        float psum = 0.f;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
          float p = __builtin_amdgcn_exp2f(s[t] - mnew);
          s[t] = p;
          psum += p;
        }
        l = l * alpha + psum;
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
          for (int jj = 0; jj < 8; ++jj) pf[sp][jj] = f2bf(s[8 * sp + jj]);
        // ---- O^T += V^T P^T
#pragma unroll
        for (int sp = 0; sp < 2; ++sp) {
          const char* vrow = sV + (j * 32 + 16 * sp + 4 * h) * G::ROW_BYTES;
#pragma unroll
          for (int dt = 0; dt < DT; ++dt) {
            s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(vrow + voff[dt]));
            s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(vrow + 8 * G::ROW_BYTES + voff[dt]));
            union { struct { s16x4 a, b; } s; bf16x8 v; } u;
            u.s.a = lo; u.s.b = hi;
            o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(u.v, pf[sp], o[dt], 0, 0, 0);
          }
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  if (!active) return;
  const float ltot = l + __shfl_xor(l, 32, 64);
  const float inv = 1.0f / ltot;
  const int q = q0 + r;
  if (q < len) {
    __bf16* op = out + ((int64_t)s0 + q) * ((int64_t)nq * D) + (int64_t)hq * D + 4 * h;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        bf16x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = f2bf(o[dt][4 * g4 + e] * inv);
        *(bf16x4*)(op + dt * 32 + 8 * g4) = v;
      }
  }
}

template <int D, int GRP>
static int launch_attn(const void* qkv, const int32_t* cu, int n_seqs, int max_seqlen, int nq, int nkv, void* out, int last_tile_only,
                       hipStream_t s) {
  int nqt = (int)lrx_cdiv(max_seqlen, 64);
  float scale_log2 = (1.0f / sqrtf((float)D)) * 1.4426950408889634f;
  hipLaunchKernelGGL((k_attn_varlen_causal<D, GRP>), dim3(last_tile_only ? n_seqs : n_seqs * nqt, nkv), dim3(128 * GRP), 0, s, (const __bf16*)qkv,
                     cu, nqt, nq, nkv, (__bf16*)out, scale_log2, last_tile_only);
  LRX_LAUNCH_CHECK();
  return LRX_OK;
}

extern "C" int lrx_attn_varlen_causal(const void* qkv, const int32_t* cu_seqlens, int32_t n_seqs, int32_t total_tokens,
                                      int32_t max_seqlen, int32_t num_q_heads, int32_t num_kv_heads, int32_t head_dim, void* out,
                                      int32_t last_tile_only, void* stream) {
  LRX_CHECK_ARG(head_dim == 64 || head_dim == 128, "attn: head_dim=%d unsupported (64 or 128)", head_dim);
  LRX_CHECK_ARG(num_kv_heads > 0 && num_q_heads % num_kv_heads == 0, "attn: nq=%d not a multiple of nkv=%d", num_q_heads, num_kv_heads);
  LRX_CHECK_ARG(max_seqlen > 0 || total_tokens == 0, "attn: max_seqlen must be > 0");
  if (total_tokens == 0 || n_seqs == 0) return LRX_OK;
  int grp = num_q_heads / num_kv_heads;
  hipStream_t s = (hipStream_t)stream;
#define LRX_ATTN_CASE(DD, GG) \
  if (head_dim == DD && grp == GG) return launch_attn<DD, GG>(qkv, cu_seqlens, n_seqs, max_seqlen, num_q_heads, num_kv_heads, out, last_tile_only, s);
  LRX_ATTN_CASE(64, 1) LRX_ATTN_CASE(64, 2) LRX_ATTN_CASE(64, 4) LRX_ATTN_CASE(64, 6) LRX_ATTN_CASE(64, 7) LRX_ATTN_CASE(64, 8)
  LRX_ATTN_CASE(128, 1) LRX_ATTN_CASE(128, 2) LRX_ATTN_CASE(128, 4) LRX_ATTN_CASE(128, 6) LRX_ATTN_CASE(128, 7) LRX_ATTN_CASE(128, 8)
#undef LRX_ATTN_CASE
  lrx_set_error("attn: GQA group size %d unsupported", grp);
  return LRX_ERR_INVALID;
}
