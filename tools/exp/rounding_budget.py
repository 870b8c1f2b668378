#!/usr/bin/env python3
"""Where does the 1 - cos(lrx, HF fp32) of a deep backbone come from?  A torch fp32 restatement of the folded lrx pipeline with every
bf16 rounding point behind a switch, run at a released backbone's REAL dims and depth on the GPU (random-init weights, the documents of
tests/test_gpu_encoder.py::test_full_depth_hf_parity), each variant compared with the all-fp32 forward (= HF fp32) and with the product.

    python tools/exp/rounding_budget.py [preset=llama31_8b]

Rounding points (product = all on):
  lin     residual GEMM output rounded to bf16 BEFORE the residual is added (HF's arithmetic: bf16 linear output + bf16 residual)
  stream  the residual stream itself stored as bf16 (off: kept at >= 16 mantissa bits -- fp32 or a hi + lo pair of bf16 arrays; the GEMM
          A operand is still bf16(x))
  qkv     q|k|v rounded to bf16 before RoPE (the staged tile), and after it
  rope2   off = rotate from the fp32 accumulator and round once
  p, o    softmax probabilities / attention output rounded to bf16
  act     SwiGLU output rounded to bf16
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from lightretriever_amd import EncoderConfig, LrxEncoder
from lightretriever_amd.encoder import rope_tables

bf = torch.bfloat16


def r16(t, on=True):
    if on == "f16":                                                         # fp16 instead of bf16 (11 significant bits)
        return t.to(torch.float16).float()
    if on == "hilo":                                                        # bf16 hi + bf16 lo (16 significant bits): two GEMM passes
        hi = t.to(bf).float()
        return hi + (t - hi).to(bf).float()
    return t.to(bf).float() if on else t


_TABLES = {}


def tables(cfg, enc, S, cs16):
    """cos/sin [S, d/2]: the product's bf16-valued table, or the fp32 values HF's fp32 model uses"""
    if cs16:
        return bf16_tables(enc, S)
    if "f32" not in _TABLES:
        import math
        from lightretriever_amd import encoder as E
        d = cfg.head_dim
        inv = 1.0 / (torch.tensor(cfg.rope_theta, dtype=torch.float32) ** (torch.arange(0, d, 2, dtype=torch.int64).float() / d))
        if cfg.rope_type == "llama3":
            old = cfg.rope_original_max_position
            low_wl, high_wl = old / cfg.rope_low_freq_factor, old / cfg.rope_high_freq_factor
            wavelen = 2 * math.pi / inv
            inv_l = torch.where(wavelen > low_wl, inv / cfg.rope_factor, inv)
            smooth = (old / wavelen - cfg.rope_low_freq_factor) / (cfg.rope_high_freq_factor - cfg.rope_low_freq_factor)
            smoothed = (1 - smooth) * inv_l / cfg.rope_factor + smooth * inv_l
            medium = ~(wavelen < high_wl) & ~(wavelen > low_wl)
            inv = torch.where(medium, smoothed, inv_l)
        fr = torch.arange(cfg.max_positions, dtype=torch.float32)[:, None] * inv[None, :].float()
        _TABLES["f32"] = (fr.cos().cuda(), fr.sin().cuda())
    return _TABLES["f32"][0][:S], _TABLES["f32"][1][:S]


def folded(L, key, ln):
    """bf16(W * gamma): what the folded-norm pipeline multiplies with (cached on the layer dict)"""
    k = key + "_f"
    if k not in L:
        L[k] = (L[key].float() * L[ln].float()[None, :]).to(bf)
    return L[k]


def bf16_tables(enc, S):
    if "bf" not in _TABLES:
        _TABLES["bf"] = (enc.rope_cos.to(bf).float(), enc.rope_sin.to(bf).float())
    return _TABLES["bf"][0][:S], _TABLES["bf"][1][:S]


def forward(enc, cfg, ids, fl):
    """one document, all-fp32 torch arithmetic with the selected roundings; returns the pooled, final-normed row (fp32)"""
    H, d, nq, nkv = cfg.hidden_size, cfg.head_dim, cfg.num_q_heads, cfg.num_kv_heads
    S = ids.numel()
    cos, sin = tables(cfg, enc, S, fl.get("cs16", True))
    x = enc.embed[ids.long()].float()                                       # exact bf16 values
    grp = nq // nkv
    causal = torch.ones(S, S, dtype=torch.bool, device=x.device).tril()
    for L in enc.layers:
        a = r16(x)                                                           # GEMM A operand (always bf16)
        rs = torch.rsqrt((a if fl["stream"] else x).pow(2).mean(-1, keepdim=True) + cfg.rms_eps)
        if fl.get("fold", True):
            t = (a @ folded(L, "wqkv", "ln1").float().T) * rs
        elif fl.get("a_gamma"):                                              # the precise stream's operand: bf16(x * gamma) (or hi + lo / unrounded), row scale on the accumulator
            if "stats" in fl:
                fl["stats"]["a"] = max(fl["stats"].get("a", 0.0), (x * L["ln1"].float()).abs().max().item(), (x * L["ln2"].float()).abs().max().item())
            t = (r16(x * L["ln1"].float(), fl.get("a_qkv", True)) @ L["wqkv"].float().T) * rs
        else:                                                                # exact weights and gamma (what HF fp32 multiplies)
            t = ((a if fl.get("a16", True) else x) * rs * L["ln1"].float()) @ L["wqkv"].float().T
        if L["bqkv"] is not None:
            t = t + L["bqkv"].float()
        t = r16(t, fl["qkv"] and fl["rope2"])
        q, k, v = t[:, :nq * d].view(S, nq, d), t[:, nq * d:(nq + nkv) * d].view(S, nkv, d), t[:, (nq + nkv) * d:].view(S, nkv, d)

        def rope(u):
            u1, u2 = u[..., :d // 2], u[..., d // 2:]
            c, s_ = cos[:, None, :], sin[:, None, :]
            return torch.cat([u1 * c - u2 * s_, u2 * c + u1 * s_], -1)
        q, k, v = r16(rope(q), fl["qkv"]), r16(rope(k), fl["qkv"]), r16(v, fl["qkv"])
        kk, vv = k.repeat_interleave(grp, 1), v.repeat_interleave(grp, 1)
        sc = torch.einsum("qhd,khd->hqk", q, kk) * (d ** -0.5)
        sc = sc.masked_fill(~causal, float("-inf"))
        m = sc.max(-1, keepdim=True).values
        p = torch.exp(sc - m)
        l = p.sum(-1, keepdim=True)                                         # (the kernels sum the fp32 p, then round p for the MFMA)
        o = torch.einsum("hqk,khd->qhd", r16(p, fl["p"]), vv) / l.permute(1, 0, 2)
        o = r16(o.reshape(S, nq * d), fl["o"])
        lin = r16(o @ L["wo"].float().T, fl["lin"])
        x = r16(x + lin, fl["stream"])
        a = r16(x)
        rs = torch.rsqrt((a if fl["stream"] else x).pow(2).mean(-1, keepdim=True) + cfg.rms_eps)
        if fl.get("fold", True):
            gu = (a @ folded(L, "wgu", "ln2").float().T) * rs
        elif fl.get("a_gamma"):
            gu = (r16(x * L["ln2"].float(), fl.get("a_gu", True)) @ L["wgu"].float().T) * rs
        else:
            gu = ((a if fl.get("a16", True) else x) * rs * L["ln2"].float()) @ L["wgu"].float().T
        I = cfg.intermediate_size
        gu = gu.view(S, I // 16, 2, 16)
        g, u = gu[:, :, 0].reshape(S, I), gu[:, :, 1].reshape(S, I)
        act = torch.nn.functional.silu(g) * u
        if "stats" in fl:                                                    # operand ranges (is fp16 wide enough?)
            fl["stats"]["act"] = max(fl["stats"].get("act", 0.0), act.abs().max().item())
            fl["stats"]["o"] = max(fl["stats"].get("o", 0.0), o.abs().max().item())
        act = r16(act, fl["act"])
        lin = r16(act @ L["wdown"].float().T, fl["lin"])
        x = r16(x + lin, fl["stream"])
    h = x[-1]
    h = h * torch.rsqrt(h.pow(2).mean() + cfg.rms_eps) * enc.final_norm.float()
    return h


def main():
    preset = sys.argv[1] if len(sys.argv) > 1 else "llama31_8b"
    cfg = getattr(EncoderConfig, preset)()
    enc = LrxEncoder.random_init(cfg, seed=5)
    g = torch.Generator().manual_seed(2024)
    lens = [512, 1, 129, 300, 64, 511, 17]
    ids = torch.randint(1000, 127000, (sum(lens),), generator=g, dtype=torch.int64).to(torch.int32).cuda()
    cu = np.concatenate([[0], np.cumsum(lens)])
    out = enc.encode_packed(ids, torch.tensor(cu, dtype=torch.int32).cuda(), 512)
    ALL = dict(lin=True, stream=True, qkv=True, rope2=True, p=True, o=True, act=True)
    NONE = {k: False for k in ALL}
    ALL = dict(ALL, fold=True, cs16=True)
    NONE = dict(NONE, fold=True, cs16=True)
    F1 = dict(ALL, rope2=False, cs16=False)
    variants = {
        "HF-fp32-like (exact W, gamma, fp32 cos/sin), A ops bf16": dict(NONE, fold=False, cs16=False),
        "fp32 + folded weights": dict(NONE, cs16=False),
        "fp32 + bf16 cos/sin": dict(NONE, fold=False),
        "F1: product with fp32 cos/sin + single-rounded RoPE": F1,
        "F1 + F4 (q,k,v,p in fp16)": dict(F1, qkv="f16", p="f16"),
        "F1 + F4 + F2 (single rounding of x + linear)": dict(F1, qkv="f16", p="f16", lin=False),
        "F1 + F4 + F2 + F3 (hi+lo stream)": dict(F1, qkv="f16", p="f16", lin=False, stream=False),
        "P (fp32 stream, A = bf16(x gamma), exact weights)": dict(ALL, fold=False, lin=False, stream=False),
        "P + F1": dict(F1, fold=False, lin=False, stream=False),
        "P + F1 + F4": dict(F1, fold=False, lin=False, stream=False, qkv="f16", p="f16"),
        "P + F1 + F4 - o": dict(F1, fold=False, lin=False, stream=False, qkv="f16", p="f16", o=False),
        "P + F1 + F4 - act": dict(F1, fold=False, lin=False, stream=False, qkv="f16", p="f16", act=False),
        "P + F1 + F4 - o - act": dict(F1, fold=False, lin=False, stream=False, qkv="f16", p="f16", o=False, act=False),
        "P + F4": dict(ALL, fold=False, lin=False, stream=False, qkv="f16", p="f16"),
        "F4 only": dict(ALL, qkv="f16", p="f16"),
        "F1 + F2": dict(F1, lin=False),
        "F1 + F2 + F3": dict(F1, lin=False, stream=False),
        "fp32 (no rounding but the GEMM A operands)": NONE,
        "product emulation (all on)": ALL,
        "- lin (single rounding of x + linear)": dict(ALL, lin=False),
        "- lin - stream (16-bit-mantissa residual stream)": dict(ALL, lin=False, stream=False),
        "- lin - stream - rope2": dict(ALL, lin=False, stream=False, rope2=False),
        "- lin - stream - rope2 - p - o": dict(ALL, lin=False, stream=False, rope2=False, p=False, o=False),
        "- lin - stream - rope2 - act": dict(ALL, lin=False, stream=False, rope2=False, act=False),
        "- lin - stream - rope2 - qkv": dict(ALL, lin=False, stream=False, rope2=False, qkv=False),
        "only stream + lin (HF-like stream, rest fp32)": dict(NONE, lin=True, stream=True),
        "only qkv/rope": dict(NONE, qkv=True, rope2=True),
        "only p, o": dict(NONE, p=True, o=True),
        "only act": dict(NONE, act=True),
    }
    # reference: everything fp32, A operands NOT rounded either (= HF fp32 of the same weights)
    def ref_forward(doc):
        global r16
        keep = r16
        r16 = lambda t, on=True: t                                           # noqa: E731
        try:
            return forward(enc, cfg, doc, dict(NONE, fold=False, cs16=False, a16=False))
        finally:
            r16 = keep
    with torch.no_grad():
        docs = [ids[cu[b]:cu[b + 1]] for b in range(len(lens))]
        ref = torch.stack([torch.nn.functional.normalize(ref_forward(dc), dim=-1) for dc in docs])
        print("%-55s max 1-cos vs fp32   (vs product)" % preset)
        print("%-55s %.3e   per doc %s" % ("PRODUCT lrx_encode_packed (precise_stream=%s)" % enc.precise, (1 - (ref * out).sum(-1)).max().item(),
                                            " ".join("%.1e" % v for v in (1 - (ref * out).sum(-1)).tolist())))
        for name, fl in variants.items():
            e = torch.stack([torch.nn.functional.normalize(forward(enc, cfg, dc, fl), dim=-1) for dc in docs])
            print("%-55s %.3e   (%.3e)  per doc %s" % (name, (1 - (ref * e).sum(-1)).max().item(), (1 - (out * e).sum(-1)).max().item(),
                                                        " ".join("%.1e" % v for v in (1 - (ref * e).sum(-1)).tolist())), flush=True)


if __name__ == "__main__":
    main()
