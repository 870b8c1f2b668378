// Experiment: achievable HBM->LDS streaming rate of global_load_lds for different request shapes (no compute).
//   pattern 0: each wave instruction reads 8 rows x 128 B (8 KiB row stride)      -- what the search kernels do
//   pattern 1: each wave instruction reads 4 rows x 256 B
//   pattern 2: each wave instruction reads 1 row  x 1 KiB contiguous
//   pattern 3: plain contiguous block sweep (each workgroup owns a contiguous region)
// build: hipcc --offload-arch=gfx950 -O3 tools/exp/stream_patterns.hip -o gpurun_out/stream_patterns ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int PAT, int NST>
__global__ void __launch_bounds__(256) k_stream(const float* __restrict__ X, long N, int D, float* __restrict__ sink) {
  __shared__ __attribute__((aligned(1024))) char smem[NST * 16384];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long n0 = (long)blockIdx.x * 128;       // 128 rows per workgroup, 16 KiB per stage (like the search kernel)
  const int nk = D / 32;                        // 64 stages of 128 rows x 128 B
  auto issue = [&](int st, int kt) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int s = (wave * 4 + i) * 64 + lane;   // 16-B slot 0..1023 of the 16 KiB stage
      const float* p;
      if (PAT == 0) { int row = s >> 3, c = s & 7; p = X + (n0 + row) * D + kt * 32 + c * 4; }
      else if (PAT == 1) { int row = s >> 4, c = s & 15; int k2 = kt >> 1, half = kt & 1; p = X + (n0 + half * 64 + row) * D + k2 * 64 + c * 4; }
      else if (PAT == 2) { int row = s >> 6, c = s & 63; int k8 = kt >> 3, sub = kt & 7; p = X + (n0 + sub * 16 + row) * D + k8 * 256 + c * 4; }
      else { p = X + n0 * D + (long)kt * 4096 + s * 4; }
      __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)(smem + st * 16384 + (wave * 4 + i) * 1024), 16, 0, 0);
    }
  };
  float acc = 0.f;
  for (int st = 0; st < NST - 1; ++st) issue(st, st);
  for (int kt = 0; kt < nk; ++kt) {
    if (NST > 2 && kt + NST - 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * 4) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + NST - 1 < nk) issue((kt + NST - 1) % NST, kt + NST - 1);
    acc += *(const float*)(smem + (kt % NST) * 16384 + tid * 16);
  }
  if (acc == 123.456f) sink[0] = acc;
}

template <int PAT, int NST>
static void run(const float* X, long N, int D, float* sink, const char* name) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  dim3 grid((unsigned)(N / 128)), block(256);
  hipLaunchKernelGGL((k_stream<PAT, NST>), grid, block, 0, 0, X, N, D, sink);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k_stream<PAT, NST>), grid, block, 0, 0, X, N, D, sink);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  printf("%-34s NST=%d  %.3f ms  %.0f GB/s\n", name, NST, ms, (double)N * D * 4 / ms / 1e6);
}

int main() {
  const long N = 1000064; const int D = 2048;   // multiple of 128 rows
  float *X, *sink;
  hipMalloc(&X, (size_t)N * D * 4); hipMalloc(&sink, 4);
  hipMemset(X, 0, (size_t)N * D * 4);
  run<0, 2>(X, N, D, sink, "8 rows x 128 B per instruction");
  run<0, 3>(X, N, D, sink, "8 rows x 128 B per instruction");
  run<0, 4>(X, N, D, sink, "8 rows x 128 B per instruction");
  run<1, 2>(X, N, D, sink, "4 rows x 256 B per instruction");
  run<1, 3>(X, N, D, sink, "4 rows x 256 B per instruction");
  run<2, 2>(X, N, D, sink, "1 row x 1 KiB per instruction");
  run<2, 3>(X, N, D, sink, "1 row x 1 KiB per instruction");
  run<2, 4>(X, N, D, sink, "1 row x 1 KiB per instruction");
  run<3, 2>(X, N, D, sink, "contiguous 16 KiB per stage");
  run<3, 3>(X, N, D, sink, "contiguous 16 KiB per stage");
  run<3, 4>(X, N, D, sink, "contiguous 16 KiB per stage");
  return 0;
}
