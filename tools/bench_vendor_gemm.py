#!/usr/bin/env python3
"""Reference point only: rocBLAS/hipBLASLt bf16 GEMM (torch.matmul) on the encoder's shapes, same random data as tools/bench_gemm.py."""
import statistics, torch
M = 131072
g = torch.Generator(device="cuda").manual_seed(0)
for name, N, K in [("qkv", 3072, 2048), ("o", 2048, 2048), ("gate_up", 16384, 2048), ("down", 2048, 8192)]:
    A = torch.randn(M, K, generator=g, device="cuda").to(torch.bfloat16)
    B = (torch.randn(N, K, generator=g, device="cuda") * 0.02).to(torch.bfloat16)
    for _ in range(3):
        torch.matmul(A, B.t())
    ts = []
    for _ in range(12):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); torch.matmul(A, B.t()); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    med = statistics.median(ts)
    print(f"vendor {name:8s} M={M} N={N} K={K}: median {med:.3f} ms = {2.0*M*N*K/med/1e9:.1f} TF/s (plain GEMM, no fused epilogue)", flush=True)
