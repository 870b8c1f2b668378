#!/usr/bin/env python3
"""Dev micro-benchmark of lrx_gemm_bf16_nt on the encoder's GEMM shapes (random bf16 operands, HIP-event timing)."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lightretriever_amd import ops

def main():
    M = int(os.environ.get("M", 131072))
    H, I, QKV = int(os.environ.get("H", 2048)), int(os.environ.get("I", 8192)), int(os.environ.get("QKV", 3072))
    shapes = [("qkv", QKV, H, 0), ("o", H, H if "QD" not in os.environ else int(os.environ["QD"]), 1), ("gate_up", 2 * I, H, 2), ("down", H, I, 1)]
    if os.environ.get("LRX_GEMM_V1"):
        shapes = [s for s in shapes if s[3] != 2]
    g = torch.Generator(device="cuda").manual_seed(0)
    for name, N, K, epi in shapes:
        A = (torch.randn(M, K, generator=g, device="cuda") * 1.0).to(torch.bfloat16)
        B = (torch.randn(N, K, generator=g, device="cuda") * 0.02).to(torch.bfloat16)
        out = torch.empty(M, N // 2 if epi == 2 else N, dtype=torch.bfloat16, device="cuda")
        resid = out if epi == 1 else None
        for _ in range(3):
            ops.gemm_bf16_nt(A, B, resid=resid, epilogue=epi, out=out)
        ts = []
        for _ in range(12):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); ops.gemm_bf16_nt(A, B, resid=resid, epilogue=epi, out=out); e1.record()
            torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        med, mn = statistics.median(ts), min(ts)
        fl = 2.0 * M * N * K
        print(f"{name:8s} M={M} N={N} K={K} epi={epi}: median {med:.3f} ms = {fl/med/1e9:.1f} TF/s   min {mn:.3f} ms = {fl/mn/1e9:.1f} TF/s", flush=True)

if __name__ == "__main__":
    main()
