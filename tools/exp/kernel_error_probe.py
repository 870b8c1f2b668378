#!/usr/bin/env python3
"""Relative RMS error of each encoder kernel against an fp64 evaluation of the SAME inputs, at Llama-3.1-8B layer shapes: which kernel
injects more than its output rounding (bf16: 2^-9/sqrt(3) = 1.1e-3, fp16: 1.4e-4, fp32: ~1e-7)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from lightretriever_amd import EncoderConfig, LrxEncoder, ops

def rel(got, want):
    return ((got.double() - want.double()).norm() / want.double().norm()).item()

def main():
    cfg = EncoderConfig.llama31_8b()
    cfg.num_layers = 2
    cfg.precise_stream = True
    enc = LrxEncoder.random_init(cfg, seed=5)
    L = enc.layers[1]
    H, d, nq, nkv, I = cfg.hidden_size, cfg.head_dim, cfg.num_q_heads, cfg.num_kv_heads, cfg.intermediate_size
    g = torch.Generator(device="cuda").manual_seed(1)
    S = 512
    x32 = torch.randn(S, H, generator=g, device="cuda") * 1.5
    a16 = (x32 * L["ln1"].float()).to(torch.bfloat16)
    rs = torch.rsqrt(x32.pow(2).mean(-1) + cfg.rms_eps)
    pos = torch.arange(S, dtype=torch.int32, device="cuda")
    perm = ops.rotary_pair_order(nq, nkv, d).cuda()
    qkv_p = ops.gemm_qkv_rope(a16, L["wqkv_c"], pos, enc.rope_cos, enc.rope_sin, nq, nkv, d, rscale=rs)
    qkv = torch.empty_like(qkv_p); qkv[:, perm] = qkv_p
    t = (a16.double() @ L["wqkv"].double().T) * rs.double()[:, None]
    qk = t[:, :(nq + nkv) * d].reshape(S, nq + nkv, d)
    c, s_ = enc.rope_cos[:S].double()[:, None, :], enc.rope_sin[:S].double()[:, None, :]
    x1, x2 = qk[..., :d // 2], qk[..., d // 2:]
    want = torch.cat([torch.cat([x1 * c - x2 * s_, x2 * c + x1 * s_], -1).reshape(S, -1), t[:, (nq + nkv) * d:]], 1)
    print("qkv+rope (fp16 out)        rel rms %.2e" % rel(qkv, want))
    # attention on the product's own fp16 qkv (logical order for the reference; the kernel takes the pair order)
    cu = torch.tensor([0, S], dtype=torch.int32, device="cuda")
    o = ops.attn_varlen_causal(qkv_p, cu, S, nq, nkv, d)
    q = qkv[:, :nq * d].double().view(S, nq, d); k = qkv[:, nq * d:(nq + nkv) * d].double().view(S, nkv, d); v = qkv[:, (nq + nkv) * d:].double().view(S, nkv, d)
    kk, vv = k.repeat_interleave(nq // nkv, 1), v.repeat_interleave(nq // nkv, 1)
    sc = torch.einsum("qhd,khd->hqk", q, kk) * d ** -0.5
    sc = sc.masked_fill(~torch.ones(S, S, dtype=torch.bool, device="cuda").tril(), float("-inf"))
    ow = torch.einsum("hqk,khd->qhd", torch.softmax(sc, -1), vv).reshape(S, nq * d)
    print("attention (bf16 out)       rel rms %.2e   logits std %.2f max %.1f" % (rel(o, ow), sc[sc > -1e30].std().item(), sc[sc > -1e30].abs().max().item()))
    # o-proj residual (fp32 stream)
    x_in = x32.clone()
    a_next, ss = ops.gemm_resid32(o, L["wo"], x_in, gamma=L["ln2"], want_ss=True)
    xw = x32.double() + o.double() @ L["wo"].double().T
    print("o-proj + residual (fp32)   rel rms %.2e   a16 %.2e" % (rel(x_in, xw), rel(a_next, xw * L["ln2"].double())))
    rsB = ops.finalize_rscale(ss, H, cfg.rms_eps)
    print("row scale                  rel rms %.2e" % rel(rsB, torch.rsqrt(xw.pow(2).mean(-1) + cfg.rms_eps)))
    act, _ = ops.gemm_bf16_nt_fused(a_next, L["wgu_c"], epilogue=2, rscale=rsB)
    gu = (a_next.double() @ L["wgu"].double().T) * rsB.double()[:, None]
    gu = gu.view(S, I // 16, 2, 16)
    gg, uu = gu[:, :, 0].reshape(S, I), gu[:, :, 1].reshape(S, I)
    print("gate-up swiglu (bf16 out)  rel rms %.2e" % rel(act, torch.nn.functional.silu(gg) * uu))
    x2_ = x_in.clone()
    ops.gemm_resid32(act, L["wdown"], x2_, want_a16=False)
    print("down + residual (fp32)     rel rms %.2e" % rel(x2_, x_in.double() + act.double() @ L["wdown"].double().T))

main()
