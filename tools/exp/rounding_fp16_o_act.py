#!/usr/bin/env python3
"""What would fp16 (instead of bf16) operands of the two residual GEMMs buy?  The torch restatement of tools/exp/rounding_budget.py on
TRAINED-LIKE weights at a backbone's real depth, the attention output `o` and / or the SwiGLU output `act` rounded to fp16, bf16 or not
at all, each against the all-fp32 forward; plus the largest |o| and |act| seen (fp16 tops out at 65504).

    python tools/exp/rounding_fp16_o_act.py [preset=llama31_8b] [docs=32] [seed=0]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

import rounding_budget as rb
from lightretriever_amd import EncoderConfig, LrxEncoder
from lightretriever_amd.synth import sink_token
from tools.parity_margin import document_lengths, documents


def main():
    preset = sys.argv[1] if len(sys.argv) > 1 else "llama31_8b"
    n_docs = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    cfg = getattr(EncoderConfig, preset)()
    enc = LrxEncoder.random_init(cfg, seed=seed, profile="trained_like")
    lens = document_lengths(n_docs, seed)
    ids, cu = documents(cfg, lens, seed, first_token=sink_token(cfg))
    out = enc.encode_packed(ids, cu, 512)
    cuh = cu.cpu().numpy()
    base = dict(lin=False, stream=False, qkv="f16", rope2=False, p="f16", o=True, act=True, fold=False, cs16=False)
    none = dict(lin=False, stream=False, qkv=False, rope2=False, p=False, o=False, act=False, fold=False, cs16=False, a16=False)
    pa = dict(base, a_gamma=True)                                            # A = bf16(x * gamma), the product's operand exactly
    variants = {
        "product arithmetic, A = bf16(x gamma)": pa,
        "  A of QKV as bf16 hi + lo": dict(pa, a_qkv="hilo"),
        "  A of gate-up as bf16 hi + lo": dict(pa, a_gu="hilo"),
        "  both A as hi + lo": dict(pa, a_qkv="hilo", a_gu="hilo"),
        "  both A hi + lo, o / act fp16": dict(pa, a_qkv="hilo", a_gu="hilo", o="f16", act="f16"),
        "  A of QKV unrounded": dict(pa, a_qkv=False),
        "  A of QKV fp16": dict(pa, a_qkv="f16"),
        "  every GEMM operand fp16 (A qkv, A gate-up, o, act)": dict(pa, a_qkv="f16", a_gu="f16", o="f16", act="f16"),
        "product arithmetic (o, act bf16)": base,
        "o fp16": dict(base, o="f16"),
        "act fp16": dict(base, act="f16"),
        "o fp16, act fp16": dict(base, o="f16", act="f16"),
        "o, act unrounded": dict(base, o=False, act=False),
    }
    with torch.no_grad():
        docs = [ids[cuh[b]:cuh[b + 1]] for b in range(n_docs)]
        keep = rb.r16
        rb.r16 = lambda t, on=True: t                                        # noqa: E731
        stats = {}
        ref = torch.stack([torch.nn.functional.normalize(rb.forward(enc, cfg, dc, dict(none, stats=stats, a_gamma=True)), dim=-1) for dc in docs])
        rb.r16 = keep
        print("%s trained-like seed %d, %d documents: max |o| %.1f, max |act| %.1f, max |x gamma| %.1f" % (preset, seed, n_docs, stats["o"], stats["act"], stats["a"]))
        g = (1 - (ref * out).sum(-1)).cpu().numpy()
        print("%-36s max %.3e  p50 %.3e  mean %.3e" % ("PRODUCT", g.max(), np.median(g), g.mean()))
        for name, fl in variants.items():
            e = torch.stack([torch.nn.functional.normalize(rb.forward(enc, cfg, dc, fl), dim=-1) for dc in docs])
            g = (1 - (ref * e).sum(-1)).cpu().numpy()
            print("%-36s max %.3e  p50 %.3e  mean %.3e" % (name, g.max(), np.median(g), g.mean()), flush=True)


if __name__ == "__main__":
    main()
