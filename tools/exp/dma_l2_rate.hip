// Experiment: L2 -> LDS rate of global_load_lds per CU when the source is L2-resident (each workgroup re-reads a small region).
// Decides whether a GEMM shape that needs ~65 GB/s of staging per CU (256x128 tiles, two workgroups per CU) is feasible.
// build: hipcc --offload-arch=gfx950 -O3 tools/exp/dma_l2_rate.hip -o tools/exp/dma_l2_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int NST, int WAVES, int STAGE_KB>
__global__ void __launch_bounds__(64 * WAVES) k_l2(const char* __restrict__ src, int region_kb, int iters, float* sink) {
  __shared__ __attribute__((aligned(1024))) char smem[NST * STAGE_KB * 1024];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int IPW = STAGE_KB / WAVES;                  // 1-KiB instructions per wave per stage
  const char* base = src + (size_t)(blockIdx.x % 64) * region_kb * 1024;    // 64 distinct regions, shared by workgroups -> L2 hits
  auto issue = [&](int st, int it) {
    const int off = (it % (region_kb / STAGE_KB)) * STAGE_KB;      // off + STAGE_KB <= region_kb: never leaves this workgroup's region
#pragma unroll
    for (int i = 0; i < IPW; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(base + (size_t)(off + wave * IPW + i) * 1024 + lane * 16),
                                       (lptr_t)(smem + st * STAGE_KB * 1024 + (wave * IPW + i) * 1024), 16, 0, 0);
  };
  float acc = 0.f;
  for (int s = 0; s < NST - 1; ++s) issue(s, s);
  for (int it = 0; it < iters; ++it) {
    if (NST > 2 && it + NST - 2 < iters) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * IPW) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (it + NST - 1 < iters) issue((it + NST - 1) % NST, it + NST - 1);
    acc += *(const float*)(smem + (it % NST) * STAGE_KB * 1024 + tid * 16);
  }
  if (acc == 123.456f) sink[0] = acc;
}

template <int NST, int WAVES, int STAGE_KB>
static void run(const char* src, float* sink, int blocks_per_cu, const char* name) {
  const int iters = 2000, region_kb = 2048;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  dim3 grid(256 * blocks_per_cu), block(64 * WAVES);
  hipLaunchKernelGGL((k_l2<NST, WAVES, STAGE_KB>), grid, block, 0, 0, src, region_kb, iters, sink);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k_l2<NST, WAVES, STAGE_KB>), grid, block, 0, 0, src, region_kb, iters, sink);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double bytes = (double)grid.x * iters * STAGE_KB * 1024;
  printf("%-44s %.3f ms  %.1f GB/s per CU  (%.1f TB/s chip)\n", name, ms, bytes / ms / 1e6 / 256, bytes / ms / 1e9);
}

int main() {
  char* src; float* sink;
  hipMalloc(&src, (size_t)64 * 2048 * 1024); hipMalloc(&sink, 4);
  hipMemset(src, 0, (size_t)64 * 2048 * 1024);
  run<2, 8, 64>(src, sink, 1, "1 wg/CU, 8 waves, 64 KiB stages x2 (current GEMM)");
  run<3, 4, 24>(src, sink, 2, "2 wg/CU, 4 waves, 24 KiB stages x3");
  run<2, 4, 24>(src, sink, 2, "2 wg/CU, 4 waves, 24 KiB stages x2");
  run<3, 4, 24>(src, sink, 1, "1 wg/CU, 4 waves, 24 KiB stages x3");
  run<4, 4, 16>(src, sink, 2, "2 wg/CU, 4 waves, 16 KiB stages x4");
  return 0;
}
