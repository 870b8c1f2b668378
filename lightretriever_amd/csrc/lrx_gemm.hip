// bf16 GEMM  C[M,N] = A[M,K] * B[N,K]^T  (nn.Linear layout, both operands K-contiguous) on gfx950 MFMA.
//
// Tile 256 x 256 x 64, 512 threads = 8 waves as 2 (M) x 4 (N), wave tile 128 x 64 = 8 x 4 MFMA 16x16x32 tiles
// (128 fp32 accumulators per lane).  Operands are staged global -> LDS with 16-byte global_load_lds (LDS-DMA, no
// VGPR round trip); the LDS image is lane-linear per wave instruction, so the bank-conflict swizzle is applied to
// the per-lane SOURCE address and undone on the ds_read_b128 side (16-B chunk index ^= (row >> 1) & 7, which makes
// every 16-lane ds_read_b128 group hit 16 distinct 16-B slots of the 256-B bank row).  Two LDS stages (128 KiB):
// tile t+1 streams in while tile t feeds the matrix cores.
//
// The MFMA is issued with the WEIGHT fragment as the A operand and the ACTIVATION fragment as the B operand, so a
// lane ends up holding 4 consecutive output columns of one output row -> 8-byte stores and lane-local SwiGLU pairs.
//
// Workgroup -> tile map: bijective XCD-chunked remap (blocks b and b+8 share an XCD's L2) and, inside an XCD's
// chunk, 8-m-tile groups walked n-fastest so the 32 co-resident tiles of an XCD share 8 A panels and 4 B panels.
#include "lrx_common.h"

#define GBM 256
#define GBN 256
#define GBK 64
#define G_TILE_BYTES (GBM * GBK * 2)  // 32 KiB per operand per stage

enum { EPI_STORE = 0, EPI_RESID = 1, EPI_SWIGLU = 2 };

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int EPI>
__global__ void __launch_bounds__(512, 2)
k_gemm_bf16_nt(const __bf16* __restrict__ A, const __bf16* __restrict__ B, __bf16* C, const __bf16* __restrict__ bias,
               const __bf16* resid, int M, int N, int K, int tiles_m, int tiles_n) {
  __shared__ __attribute__((aligned(1024))) char smem[4 * G_TILE_BYTES];  // [stage][A|B]

  // ---- workgroup -> tile
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int xcd = bid & 7, qd = nwg >> 3, rm = nwg & 7;
  const int t = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (bid >> 3);
  const int GM = 8;
  const int width = GM * tiles_n;
  const int group = t / width, first_m = group * GM;
  const int gsz = min(tiles_m - first_m, GM);
  const int tin = t - group * width;
  const int tm = first_m + tin % gsz, tn = tin / gsz;
  const int m0 = tm * GBM, n0 = tn * GBN;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;

  // ---- staging sources: 4 LDS-DMA instructions per operand per wave per tile; lane l of instruction i fills slot
  //      s = (wave*4+i)*64 + l  ->  row = s>>3, swizzled chunk position cs = s&7 holding logical chunk cs ^ ((row>>1)&7)
  const __bf16* pa[4];
  const __bf16* pb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int s = (wave * 4 + i) * 64 + lane;
    int row = s >> 3, c = (s & 7) ^ ((row >> 1) & 7);
    int ga = min(m0 + row, M - 1), gb = min(n0 + row, N - 1);
    pa[i] = A + (int64_t)ga * K + c * 8;
    pb[i] = B + (int64_t)gb * K + c * 8;
  }
  auto stage = [&](int st, int k0) {
    char* sA = smem + st * (2 * G_TILE_BYTES);
    char* sB = sA + G_TILE_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      __builtin_amdgcn_global_load_lds((gptr_t)(pa[i] + k0), (lptr_t)(sA + (wave * 4 + i) * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(pb[i] + k0), (lptr_t)(sB + (wave * 4 + i) * 1024), 16, 0, 0);
    }
  };

  // ---- fragment read offsets (bytes inside an operand tile), lane part
  const int fr = lane & 15, fq = lane >> 4;
  const int xs = fr >> 1;  // (row>>1)&7 for row = 16*k + fr
  int laneoff[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) laneoff[ks] = fr * 128 + (((ks * 4 + fq) ^ xs) << 4);
  const int a_base = (wm * 128) * 128, b_base = (wn * 64) * 128;

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = K / GBK;
  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) stage(cur ^ 1, (kt + 1) * GBK);
    const char* sA = smem + cur * (2 * G_TILE_BYTES) + a_base;
    const char* sB = smem + cur * (2 * G_TILE_BYTES) + G_TILE_BYTES + b_base;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[8], bfr[4];
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) bfr[ni] = *(const bf16x8*)(sB + ni * 2048 + laneoff[ks]);
#pragma unroll
      for (int mi = 0; mi < 8; ++mi) af[mi] = *(const bf16x8*)(sA + mi * 2048 + laneoff[ks]);
#pragma unroll
      for (int mi = 0; mi < 8; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[ni], af[mi], acc[mi][ni], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // ---- epilogue: lane holds row m = .. + fr, columns n = .. + fq*4 + {0,1,2,3}
  const int mrow0 = m0 + wm * 128 + fr;
  const int ncol0 = n0 + wn * 64 + fq * 4;
  if (EPI == EPI_SWIGLU) {
    const int ldc = N >> 1;
    const int oc0 = ((n0 + wn * 64) >> 1) + fq * 4;
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) {
      int m = mrow0 + mi * 16;
      if (m >= M) continue;
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        int n = ncol0 + ni * 16;  // gate column in the interleaved layout; its up partner is n + 32
        if (n >= N) continue;
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float g = acc[mi][ni][r], u = acc[mi][ni + 2][r];
          o[r] = f2bf(g / (1.0f + __expf(-g)) * u);
        }
        *(bf16x4*)(C + (int64_t)m * ldc + oc0 + ni * 16) = o;
      }
    }
  } else {
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) {
      int m = mrow0 + mi * 16;
      if (m >= M) continue;
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        int n = ncol0 + ni * 16;
        if (n >= N) continue;
        f32x4 v = acc[mi][ni];
        if (EPI == EPI_STORE && bias != nullptr) {
          bf16x4 bv = *(const bf16x4*)(bias + n);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += bf2f(bv[r]);
        }
        if (EPI == EPI_RESID) {
          bf16x4 rv = *(const bf16x4*)(resid + (int64_t)m * N + n);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += bf2f(rv[r]);
        }
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = f2bf(v[r]);
        *(bf16x4*)(C + (int64_t)m * N + n) = o;
      }
    }
  }
}

extern "C" int lrx_gemm_bf16_nt(const void* A, const void* B, void* C, const void* bias, const void* resid, int32_t M, int32_t N,
                                int32_t K, int32_t epilogue, void* stream) {
  LRX_CHECK_ARG(M >= 0 && N > 0 && K > 0, "gemm: bad shape M=%d N=%d K=%d", M, N, K);
  LRX_CHECK_ARG(K % GBK == 0, "gemm: K=%d must be a multiple of %d", K, GBK);
  LRX_CHECK_ARG(N % 4 == 0, "gemm: N=%d must be a multiple of 4", N);
  LRX_CHECK_ARG(epilogue >= 0 && epilogue <= 2, "gemm: unknown epilogue %d", epilogue);
  LRX_CHECK_ARG(epilogue != EPI_SWIGLU || N % 64 == 0, "gemm: SwiGLU epilogue needs N %% 64 == 0 (N=%d)", N);
  LRX_CHECK_ARG(epilogue != EPI_RESID || resid != nullptr, "gemm: residual epilogue without resid");
  if (M == 0) return LRX_OK;
  int tiles_m = (int)lrx_cdiv(M, GBM), tiles_n = (int)lrx_cdiv(N, GBN);
  dim3 grid(tiles_m * tiles_n), block(512);
  hipStream_t s = (hipStream_t)stream;
  const __bf16 *a = (const __bf16*)A, *b = (const __bf16*)B, *bi = (const __bf16*)bias, *re = (const __bf16*)resid;
  __bf16* c = (__bf16*)C;
  switch (epilogue) {
    case EPI_STORE: hipLaunchKernelGGL(k_gemm_bf16_nt<EPI_STORE>, grid, block, 0, s, a, b, c, bi, re, M, N, K, tiles_m, tiles_n); break;
    case EPI_RESID: hipLaunchKernelGGL(k_gemm_bf16_nt<EPI_RESID>, grid, block, 0, s, a, b, c, bi, re, M, N, K, tiles_m, tiles_n); break;
    default: hipLaunchKernelGGL(k_gemm_bf16_nt<EPI_SWIGLU>, grid, block, 0, s, a, b, c, bi, re, M, N, K, tiles_m, tiles_n); break;
  }
  LRX_LAUNCH_CHECK();
  return LRX_OK;
}
