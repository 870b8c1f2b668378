// Experiment: would a 4-wave GEMM workgroup with 128 x 128 per-wave tiles (256 accumulators per lane, one wave per SIMD, 16 fragment
// ds_read_b128 per 64 MFMAs) beat the product's 8-wave / 128 x 64 shape (128 accumulators, two waves per SIMD, 24 reads per 64 MFMAs) under
// the power cap?  Both shapes as MOCK kernels with the instruction mix of the real 256 x 256 x 64 K loop -- per K-tile 64 KiB of LDS-DMA
// from an L2-resident source into a 2-stage 128-KiB ring, the fragment reads of random bf16 data, 512 MFMA 16x16x32 per workgroup, the
// barriers and counted vmcnt waits of a stage hand-over -- but no tile addressing, no epilogue, garbage results: only the rate matters.
// Compiler-scheduled straight-line code, two barriers per K-tile: not the product's staggered 4-phase schedule (which is ~15 % faster than the
// 8-wave mock) -- the question is how far apart the two SHAPES are.
// build: hipcc --offload-arch=gfx950 -O3 tools/exp/gemm_mock.hip -o tools/exp/gemm_mock
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// WAVES = 8: wave tile 128 x 64 (MT = 8 row fragments of 16, NT = 4 column fragments), 4: 128 x 128 (MT = 8, NT = 8)
template <int WAVES>
__global__ void __launch_bounds__(64 * WAVES, WAVES == 8 ? 2 : 1) k_mock(const char* __restrict__ src, float* __restrict__ out, int nk) {
  constexpr int MT = 8, NT = WAVES == 8 ? 4 : 8;
  constexpr int DMA = 65536 / (WAVES * 64 * 16);            // 16-B LDS-DMA instructions per lane per K-tile (8 or 16)
  __shared__ __attribute__((aligned(1024))) char smem[131072];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const char* base = src + (size_t)(blockIdx.x & 63) * (1 << 20) + tid * 16;     // 64 L2-resident 1-MiB regions
  auto issue = [&](int stage, int kt) {
#pragma unroll
    for (int i = 0; i < DMA; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(base + ((size_t)((kt & 15) * DMA + i)) * (WAVES * 1024)), (lptr_t)(smem + stage * 65536 + (i * WAVES + wave) * 1024), 16, 0, 0);
  };
  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // fragment addresses: conflict-free linear 1-KiB reads
  const char* fa = smem + (wave & 1) * 16384 + lane * 16;
  const char* fb = smem + 32768 + (wave >> 1) * (NT * 2048) % 32768 + lane * 16;
  issue(0, 0);
  if (nk > 1) issue(1, 1);
  for (int kt = 0; kt < nk; ++kt) {
    const char* sa = fa + (kt & 1) * 65536;
    const char* sb = fb + (kt & 1) * 65536;
    if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                           // stage kt landed for everybody
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 a[MT], b[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) a[i] = *(const bf16x8*)(sa + (i * 2 + ks) * 1024);
#pragma unroll
      for (int j = 0; j < NT; ++j) b[j] = *(const bf16x8*)(sb + (j * 2 + ks) * 1024);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                           // everybody has read stage kt: refill it
    if (kt + 2 < nk) issue(kt & 1, kt + 2);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  out[(size_t)blockIdx.x * 64 * WAVES + tid] = s;
}

template <int WAVES>
static void run(const char* src, float* out, const char* name) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int nb = 256;
  int nk = 20000;
  float ms = 0.f;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k_mock<WAVES>), dim3(nb), dim3(64 * WAVES), 0, 0, src, out, nk);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double fl = 2.0 * 256 * 256 * 64 * (double)nk * nb;
    if (rep == 0) nk = (int)(nk * 400.0 / ms);
    else printf("%-52s run %d: %7.1f ms  %7.1f TFLOP/s  (%.2f us per K-tile)\n", name, rep, ms, fl / ms / 1e9, ms * 1e3 / nk);
  }
  fflush(stdout);
}

int main() {
  char* src; float* out;
  (void)hipMalloc(&src, 64u << 20);
  (void)hipMalloc(&out, 256 * 512 * 4);
  uint16_t* h = (uint16_t*)malloc(64u << 20);
  srand(1);
  for (size_t i = 0; i < (64u << 20) / 2; ++i) h[i] = (uint16_t)(((rand() & 1) << 15) | ((126 + (rand() & 1)) << 7) | (rand() & 127));
  (void)hipMemcpy(src, h, 64u << 20, hipMemcpyHostToDevice);
  run<8>(src, out, "8 waves, 128 x 64 per wave (24 reads / 64 MFMAs)");
  run<4>(src, out, "4 waves, 128 x 128 per wave (16 reads / 64 MFMAs)");
  run<8>(src, out, "8 waves again");
  return 0;
}
