// Experiment: what does the power cap leave of the bf16 MFMA peak, per MFMA shape and per amount of LDS fragment traffic?
// Every CU runs 8 waves (2 per SIMD, 128 fp32 accumulators per lane, like k_gemm_bf16_nt); per "K-tile" a wave issues the
// MFMAs of a 128 x 64 x 64 wave tile (64 x v_mfma_f32_16x16x32_bf16 or 32 x v_mfma_f32_32x32x16_bf16) and LDSR ds_read_b128
// fragment reads of random bf16 data (0 = operands stay in registers; 24 = what the 256x256 tile with 2x4 waves reads;
// 16 = what 128x128 per-wave tiles would read).  No global memory traffic, no barriers: the rate that comes out is the
// power / clock ceiling for that instruction mix.  Each variant runs ~0.4 s so that the power controller has settled.
// build: hipcc --offload-arch=gfx950 -O3 tools/exp/mfma_power.hip -o tools/exp/mfma_power
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;

template <int SHAPE, int LDSR>
__global__ void __launch_bounds__(512, 2) k_mfma(const uint4* __restrict__ seed, float* __restrict__ out, int iters) {
  __shared__ __attribute__((aligned(1024))) uint4 lds[8192];   // 128 KiB: one workgroup per CU
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 8192; i += 512) lds[i] = seed[i];
  __syncthreads();
  // conflict-free fragment addresses (row = lane&15, 16-B chunk XOR-swizzled), one base per wave
  const int fr = lane & 15, fq = lane >> 4;
  const char* base = (const char*)lds + (tid >> 6) * 16384 + fr * 128 + ((fq ^ (fr >> 1)) << 4);
  bf16x8 a[4][2], b[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      a[i][k] = *(const bf16x8*)(base + i * 2048 + k * 64);
      b[i][k] = *(const bf16x8*)(base + 8192 + i * 2048 + k * 64);
    }
  if (SHAPE == 0) {
    f32x4 acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
      const char* p = base + (it & 1) * 64;
#pragma unroll
      for (int g = 0; g < 8; ++g) {          // 8 groups of 8 MFMAs; LDSR/8 reads per group
#pragma unroll
        for (int r = 0; r < LDSR / 8; ++r) {
          const int idx = g * (LDSR / 8) + r;
          bf16x8 v = *(const bf16x8*)(p + (idx & 7) * 2048 + (idx >> 3) * 32);
          if (idx & 1) a[(idx >> 1) & 3][(idx >> 3) & 1] = v; else b[(idx >> 1) & 3][(idx >> 3) & 1] = v;
        }
#pragma unroll
        for (int m = 0; m < 8; ++m) {
          const int i = m & 3, j = g, k = m >> 2;
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j & 3][k], a[i][k], acc[i][j], 0, 0, 0);
        }
      }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[blockIdx.x * 512 + tid] = s;
  } else {
    f32x16 acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
      const char* p = base + (it & 1) * 64;
#pragma unroll
      for (int g = 0; g < 8; ++g) {          // 8 groups of 4 MFMAs (same flops as 8 of the small shape)
#pragma unroll
        for (int r = 0; r < LDSR / 8; ++r) {
          const int idx = g * (LDSR / 8) + r;
          bf16x8 v = *(const bf16x8*)(p + (idx & 7) * 2048 + (idx >> 3) * 32);
          if (idx & 1) a[(idx >> 1) & 3][(idx >> 3) & 1] = v; else b[(idx >> 1) & 3][(idx >> 3) & 1] = v;
        }
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const int j = (g & 1) * 4 + m;
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[m][g & 1], a[(m + g) & 3][(g >> 1) & 1], acc[j], 0, 0, 0);
        }
      }
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) s += acc[j][e];
    out[blockIdx.x * 512 + tid] = s;
  }
}

template <int SHAPE, int LDSR>
static void run(const uint4* seed, float* out, const char* name) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int nb = 256;
  // flops per wave per iteration: 64 x 16x16x32 = 64 * 16384
  const double fl_it = 64.0 * 16384.0 * 8 * nb;
  int iters = 20000;
  float ms = 0.f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k_mfma<SHAPE, LDSR>), dim3(nb), dim3(512), 0, 0, seed, out, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    if (rep == 0) iters = (int)(iters * 400.0 / ms);      // ~0.4 s per timed run
    else printf("%-44s run %d: %7.1f ms  %7.1f TFLOP/s  (= %.3f GHz x full MFMA rate)\n", name, rep, ms, fl_it * iters / ms / 1e9,
                fl_it * iters / ms / 1e9 / 1041.7);
  }
  fflush(stdout);
}

int main() {
  uint4* seed; float* out;
  hipMalloc(&seed, 8192 * 16);
  hipMalloc(&out, 256 * 512 * 4);
  uint16_t* h = (uint16_t*)malloc(8192 * 16);
  srand(1);
  for (int i = 0; i < 8192 * 8; ++i) {        // random bf16 in +-[0.5, 2): sign, exponent 126..127, random mantissa
    h[i] = (uint16_t)(((rand() & 1) << 15) | ((126 + (rand() & 1)) << 7) | (rand() & 127));
  }
  hipMemcpy(seed, h, 8192 * 16, hipMemcpyHostToDevice);
  run<0, 0>(seed, out, "16x16x32, operands in registers");
  run<1, 0>(seed, out, "32x32x16, operands in registers");
  run<0, 24>(seed, out, "16x16x32 + 24 ds_read_b128 per 64 MFMAs");
  run<1, 24>(seed, out, "32x32x16 + 24 ds_read_b128 per 32 MFMAs");
  run<0, 16>(seed, out, "16x16x32 + 16 ds_read_b128 per 64 MFMAs");
  run<1, 16>(seed, out, "32x32x16 + 16 ds_read_b128 per 32 MFMAs");
  run<0, 8>(seed, out, "16x16x32 +  8 ds_read_b128 per 64 MFMAs");
  return 0;
}
