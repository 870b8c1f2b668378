// NEGATIVE RESULT kept for the record (not built into liblrx): measured 980-1050 TFLOP/s on the four encoder GEMM shapes against
// 1190-1440 for the 256x256 kernel of lightretriever_amd/csrc/lrx_gemm.hip -- with 24-KiB stages the LDS carries 576 B per MFMA
// (12 fragment reads per 32 MFMAs + 1.5x the DMA writes of the big tile) and the plain read-then-MFMA loop leaves the matrix pipe idle
// ~48 % of the time even with two workgroups per CU.  Results were bit-compatible (6 parity cases passed).
// Second GEMM shape: C[M,N] = A[M,K] . B[N,K]^T with 256x128 output tiles, 4 waves, K-slices of 32 and a 3-stage LDS ring
// (72 KiB) so that TWO workgroups share a CU: one workgroup's prologue / epilogue (a K = 2048 tile spends ~15 % of its time
// there in the 256x256 one-workgroup-per-CU kernel of lrx_gemm.hip) overlaps the other's MFMA main loop.
//   wave (wr, wc) of a 2x2 grid owns 128 rows x 64 columns: acc[8][4] f32x4 = 128 VGPRs, weight fragment as MFMA-A so a lane
//   owns one row x 4 consecutive columns (same convention as lrx_gemm.hip);
//   LDS stage = A 256 rows x 64 B + B 128 rows x 64 B, 16-B chunk c of row r stored at position c ^ ((r >> 2) & 3): the 16
//   rows a quarter-wave reads land in 16 distinct 16-B bank groups (conflict-free ds_read_b128);
//   epilogue staged through LDS like lrx_gemm.hip (row-major bf16 tile, 16-B stores, fused residual / SwiGLU).
#include "lrx_common.h"

#define G2_BM 256
#define G2_BN 128
#define G2_BK 32
#define G2_NST 3
#define G2_STAGE ((G2_BM + G2_BN) * G2_BK * 2)   // 24 KiB

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

enum { G2_STORE = 0, G2_RESID = 1, G2_SWIGLU = 2 };

template <int EPI>
__global__ void __launch_bounds__(256, 2)
k_gemm2_bf16_nt(const __bf16* __restrict__ A, const __bf16* __restrict__ B, __bf16* C, const __bf16* __restrict__ bias, const __bf16* resid,
                int M, int N, int K, int tiles_m, int tiles_n) {
  __shared__ __attribute__((aligned(1024))) char smem[G2_NST * G2_STAGE];
  // ---- workgroup -> tile: XCD-chunked, groups of 8 m-tiles sweeping n (as lrx_gemm.hip)
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, qd = nwg >> 3, rm = nwg & 7;
  const int t_lin = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (bid >> 3);
  const int GM = 8;
  const int width = GM * tiles_n;
  const int group = t_lin / width, first_m = group * GM;
  const int gsz = min(tiles_m - first_m, GM);
  const int tin = t_lin - group * width;
  const int tm = first_m + tin % gsz, tn = tin / gsz;
  const int m0 = tm * G2_BM, n0 = tn * G2_BN;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;

  // ---- LDS-DMA: a 1-KiB instruction covers 16 rows x 64 B; slot s = row*4 + pos, pos holds chunk pos ^ ((row >> 2) & 3).
  //      A: 16 instructions per stage (4 per wave), B: 8 (2 per wave).
  const __bf16 *pA[4], *pB[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int s = (wave * 4 + i) * 64 + lane;
    const int row = s >> 2, c = (s & 3) ^ ((row >> 2) & 3);
    pA[i] = A + (int64_t)min(m0 + row, M - 1) * K + c * 8;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int s = (wave * 2 + i) * 64 + lane;
    const int row = s >> 2, c = (s & 3) ^ ((row >> 2) & 3);
    pB[i] = B + (int64_t)min(n0 + row, N - 1) * K + c * 8;
  }
  auto issue = [&](int st, int kt) {
    char* sA = smem + st * G2_STAGE;
    char* sB = sA + G2_BM * G2_BK * 2;
#pragma unroll
    for (int i = 0; i < 4; ++i) __builtin_amdgcn_global_load_lds((gptr_t)(pA[i] + kt * G2_BK), (lptr_t)(sA + (wave * 4 + i) * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < 2; ++i) __builtin_amdgcn_global_load_lds((gptr_t)(pB[i] + kt * G2_BK), (lptr_t)(sB + (wave * 2 + i) * 1024), 16, 0, 0);
  };

  const int fr = lane & 15, fq = lane >> 4;
  // fragment (row r = base + fr, chunk fq): byte r*64 + ((fq ^ ((r >> 2) & 3)) << 4); base is a multiple of 16 -> (r >> 2) & 3 = (fr >> 2) & 3
  const int foff = fr * 64 + ((fq ^ ((fr >> 2) & 3)) << 4);
  const int a_off = (wr * 128) * 64 + foff;
  const int b_off = G2_BM * G2_BK * 2 + (wc * 64) * 64 + foff;

  f32x4 acc[8][4];
#pragma unroll
  for (int mi = 0; mi < 8; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = K / G2_BK;
  issue(0, 0);
  if (nk > 1) issue(1, 1);
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // the younger stage's 6 instructions may stay in flight
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + 2 < nk) issue((kt + 2) % G2_NST, kt + 2);
    const char* sb = smem + (kt % G2_NST) * G2_STAGE;
    bf16x8 a[8], b[4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) b[ni] = *(const bf16x8*)(sb + b_off + ni * 1024);
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) a[mi] = *(const bf16x8*)(sb + a_off + mi * 1024);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[ni], a[mi], acc[mi][ni], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  }
  __syncthreads();

  // ---- epilogue: bf16 tile staged row-major in LDS (16-B chunk index XOR (row & 15)), then 16-B row-major stores
  constexpr int CW = (EPI == G2_SWIGLU) ? 64 : 128;   // output columns of this tile
  constexpr int CPR = CW / 8;
#pragma unroll
  for (int mi = 0; mi < 8; ++mi) {
    const int row = wr * 128 + mi * 16 + fr;
    if (EPI == G2_SWIGLU) {
      // wave columns [nb, nb+16) = gate, [nb+16, nb+32) = up of output columns nb/2 .. nb/2+15 (16-row interleaved weights)
#pragma unroll
      for (int np = 0; np < 2; ++np) {
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float g = acc[mi][2 * np][r], u = acc[mi][2 * np + 1][r];
          o[r] = f2bf(g / (1.0f + __expf(-g)) * u);
        }
        const int col = wc * 32 + np * 16 + fq * 4;
        *(bf16x4*)(smem + row * (CW * 2) + ((((col >> 3) ^ (row & 7)) << 4) | ((col & 4) << 1))) = o;
      }
    } else {
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        const int col = wc * 64 + ni * 16 + fq * 4;
        f32x4 v = acc[mi][ni];
        if (EPI == G2_STORE && bias != nullptr) {
          const bf16x4 bv = *(const bf16x4*)(bias + min(n0 + col, N - 4));
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += bf2f(bv[r]);
        }
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = f2bf(v[r]);
        *(bf16x4*)(smem + row * (CW * 2) + ((((col >> 3) ^ (row & 15)) << 4) | ((col & 4) << 1))) = o;
      }
    }
  }
  __syncthreads();
  constexpr int XM = (EPI == G2_SWIGLU) ? 7 : 15;     // chunk swizzle mask (8 chunks per row for SwiGLU, 16 otherwise)
  const int ldc = (EPI == G2_SWIGLU) ? (N >> 1) : N;
  const int c0 = (EPI == G2_SWIGLU) ? (n0 >> 1) : n0;
  constexpr int NIT = (256 * CPR) / 256;
  bf16x8 rv[EPI == G2_RESID ? NIT : 1];
  if (EPI == G2_RESID) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int q = it * 256 + tid;
      const int m = min(m0 + q / CPR, M - 1), n = min(c0 + (q % CPR) * 8, N - 8);
      rv[it] = *(const bf16x8*)(resid + (int64_t)m * N + n);
    }
  }
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int q = it * 256 + tid;
    const int row = q / CPR, ch = q % CPR;
    const int m = m0 + row, n = c0 + ch * 8;
    if (m >= M || n >= ldc) continue;
    bf16x8 v = *(const bf16x8*)(smem + row * (CW * 2) + ((ch ^ (row & XM)) << 4));
    if (EPI == G2_RESID) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = f2bf(bf2f(v[e]) + bf2f(rv[it][e]));
    }
    *(bf16x8*)(C + (int64_t)m * ldc + n) = v;
  }
}

// used by lrx_gemm_bf16_nt when the 256x128 shape is selected (lrx_set_gemm_shape)
int lrx_gemm2_launch(const void* A, const void* B, void* C, const void* bias, const void* resid, int M, int N, int K, int epilogue, hipStream_t s) {
  LRX_CHECK_ARG(K % G2_BK == 0 && N % 8 == 0, "gemm2: K=%d must be a multiple of %d and N=%d of 8", K, G2_BK, N);
  if (M == 0) return LRX_OK;
  const int tiles_m = (int)lrx_cdiv(M, G2_BM), tiles_n = (int)lrx_cdiv(N, G2_BN);
  dim3 grid(tiles_m * tiles_n), block(256);
  const __bf16 *a = (const __bf16*)A, *b = (const __bf16*)B, *bi = (const __bf16*)bias, *re = (const __bf16*)resid;
  __bf16* c = (__bf16*)C;
  switch (epilogue) {
    case G2_STORE: hipLaunchKernelGGL(k_gemm2_bf16_nt<G2_STORE>, grid, block, 0, s, a, b, c, bi, re, M, N, K, tiles_m, tiles_n); break;
    case G2_RESID: hipLaunchKernelGGL(k_gemm2_bf16_nt<G2_RESID>, grid, block, 0, s, a, b, c, bi, re, M, N, K, tiles_m, tiles_n); break;
    default: hipLaunchKernelGGL(k_gemm2_bf16_nt<G2_SWIGLU>, grid, block, 0, s, a, b, c, bi, re, M, N, K, tiles_m, tiles_n); break;
  }
  LRX_LAUNCH_CHECK();
  return LRX_OK;
}
