// Experiment: is the GEMM epilogue's store phase bound per CU or by the chip-wide write path?  Every workgroup writes one
// 256 x 256 bf16 C tile (128 KiB) exactly like the epilogue (16 B per lane, 2 rows x 512 B per wave instruction, row stride N*2),
// once with all 256 CUs writing at the same time and once with fewer workgroups.  Reports the time of the store phase per workgroup.
// build: hipcc --offload-arch=gfx950 -O3 tools/exp/store_burst.hip -o tools/exp/store_burst
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__global__ void __launch_bounds__(512) k_store(uint4* __restrict__ C, int N, int tiles_n, int reps, long long* __restrict__ cyc) {
  const int tid = threadIdx.x;
  long long t0 = 0, t1 = 0;
  for (int r = 0; r < reps; ++r) {
    const int tile = blockIdx.x + r * gridDim.x;
    const int tm = tile / tiles_n, tn = tile % tiles_n;
    __syncthreads();
    if (r == 1) t0 = wall_clock64();
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const int q = it * 512 + tid;            // 8192 chunks of 16 B
      const int row = q >> 5, ch = q & 31;
      C[((size_t)(tm * 256 + row) * N + tn * 256) / 8 + ch] = make_uint4(q, r, tile, 7);
    }
    __builtin_amdgcn_s_waitcnt(0);             // all stores of this wave acknowledged
    __syncthreads();
  }
  t1 = wall_clock64();
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
  const int M = 131072, N = 16384, tiles_n = N / 256;
  uint4* C; long long* cyc;
  hipMalloc(&C, (size_t)M * N * 2);
  hipMalloc(&cyc, 4096 * sizeof(long long));
  long long h[4096];
  for (int nb : {256, 128, 64, 32, 8}) {
    const int reps = 9;                          // rep 0 warms up, reps 1..8 timed
    hipLaunchKernelGGL(k_store, dim3(nb), dim3(512), 0, 0, C, N, tiles_n, reps, cyc);
    hipDeviceSynchronize();
    hipMemcpy(h, cyc, nb * sizeof(long long), hipMemcpyDeviceToHost);
    double avg = 0; for (int i = 0; i < nb; ++i) avg += h[i]; avg /= nb;
    const double us = avg / 100.0 / 8.0;         // wall_clock64: 100 MHz; 8 timed tiles
    printf("%3d workgroups writing at once: %.2f us per 128-KiB tile  -> %.1f GB/s per CU, %.2f TB/s total\n", nb, us, 131072 / us / 1e3, nb * 131072 / us / 1e6);
  }
  return 0;
}
