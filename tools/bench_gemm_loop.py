#!/usr/bin/env python3
"""Back-to-back gate-up GEMMs (M=131072, N=16384, K=2048, SwiGLU epilogue) for LOOPS launches: a steady load for tools/power_probe.sh."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lightretriever_amd import ops

M, N, K = 131072, 16384, 2048
g = torch.Generator(device="cuda").manual_seed(0)
A = torch.randn(M, K, generator=g, device="cuda").to(torch.bfloat16)
B = (torch.randn(N, K, generator=g, device="cuda") * 0.02).to(torch.bfloat16)
out = torch.empty(M, N // 2, dtype=torch.bfloat16, device="cuda")
for _ in range(3):
    ops.gemm_bf16_nt(A, B, epilogue=2, out=out)
torch.cuda.synchronize()
loops = int(os.environ.get("LOOPS", 400))
for rep in range(int(os.environ.get("REPS", 4))):
    t0 = time.perf_counter()
    for _ in range(loops):
        ops.gemm_bf16_nt(A, B, epilogue=2, out=out)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"rep {rep}: {loops} launches in {dt:.2f} s = {2.0 * M * N * K * loops / dt / 1e12:.1f} TFLOP/s", flush=True)
