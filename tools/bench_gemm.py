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

def rope_variant():
    """the fused QKV + RoPE projection (rotary-pair weights, fp32 table, fp16 out) next to the plain store on the same shape, and the precise
    residual GEMM (fp32 stream) next to the bf16 one"""
    from lightretriever_amd import EncoderConfig, rope_tables
    M = int(os.environ.get("M", 131072))
    g = torch.Generator(device="cuda").manual_seed(0)
    for tag, cfg in (("1B", EncoderConfig.llama32_1b()), ("8B", EncoderConfig.llama31_8b())):
        H, d, nq, nkv, I = cfg.hidden_size, cfg.head_dim, cfg.num_q_heads, cfg.num_kv_heads, cfg.intermediate_size
        Mx = M if tag == "1B" else M // 2
        N = (nq + 2 * nkv) * d
        cos, sin = (t.cuda() for t in rope_tables(cfg))
        A = torch.randn(Mx, H, generator=g, device="cuda").to(torch.bfloat16)
        W = (torch.randn(N, H, generator=g, device="cuda") * 0.02).to(torch.bfloat16)
        pos = (torch.arange(Mx, device="cuda") % 512).to(torch.int32)
        rs = torch.rand(Mx, generator=g, device="cuda") + 0.5
        out = torch.empty(Mx, N, dtype=torch.bfloat16, device="cuda")
        def t(fn, n=12):
            for _ in range(3): fn()
            ts = []
            for _ in range(n):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
            return statistics.median(ts)
        fl = 2.0 * Mx * N * H
        a = t(lambda: ops.gemm_bf16_nt(A, W, epilogue=0, out=out))
        b = t(lambda: ops.gemm_qkv_rope(A, W, pos, cos, sin, nq, nkv, d, rscale=rs))
        print(f"{tag} qkv M={Mx} N={N} K={H}: plain store {a:.3f} ms = {fl/a/1e9:.0f} TF/s | fused rope {b:.3f} ms = {fl/b/1e9:.0f} TF/s", flush=True)
        for name, K in (("o", nq * d), ("down", I)):
            Ak = torch.randn(Mx, K, generator=g, device="cuda").to(torch.bfloat16)
            Wk = (torch.randn(H, K, generator=g, device="cuda") * 0.02).to(torch.bfloat16)
            x16 = torch.randn(Mx, H, generator=g, device="cuda").to(torch.bfloat16)
            x32 = torch.randn(Mx, H, generator=g, device="cuda")
            gam = (1 + 0.1 * torch.randn(H, generator=g, device="cuda")).to(torch.bfloat16)
            flk = 2.0 * Mx * H * K
            a = t(lambda: ops.gemm_bf16_nt_fused(Ak, Wk, resid=x16, epilogue=1, want_ss=True))
            b = t(lambda: ops.gemm_resid32(Ak, Wk, x32, gamma=gam, want_ss=True))
            print(f"{tag} {name} M={Mx} N={H} K={K}: bf16 stream {a:.3f} ms = {flk/a/1e9:.0f} TF/s | fp32 stream {b:.3f} ms = {flk/b/1e9:.0f} TF/s", flush=True)


if __name__ == "__main__":
    if os.environ.get("VARIANTS"):
        rope_variant()
    else:
        main()
