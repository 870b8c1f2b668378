#!/usr/bin/env python3
"""Cosine gap of lrx_encode_packed against the HF transformers model -- fp32 AND bf16 -- on the same GPU, for a released backbone at its
real dims and depth, with Gaussian or TRAINED-LIKE synthetic weights (lightretriever_amd/synth.py).  Library of
tests/test_gpu_trained_like.py and a CLI:

    python tools/parity_margin.py [--presets llama32_1b,qwen25_1_5b,...] [--profile trained_like|gaussian] [--seeds 0,1,2] [--docs 64]
                                  [--out gpurun_out/parity.jsonl] [--streams default|both]

The forward being matched: finetune/modeling_hybrid.py:248-278 (HF LlamaModel / Qwen2Model -> lasttoken pooling -> normalise); the
reference runs it in bf16 under autocast (inference/exact_search_base.py:211), hence the second column."""
import argparse
import dataclasses
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def hf_model_for(cfg, dtype=torch.float32):
    """The HF transformers model of an EncoderConfig, on the GPU, uninitialised weights (the caller loads a state dict)."""
    common = dict(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size, num_hidden_layers=cfg.num_layers,
                  num_attention_heads=cfg.num_q_heads, num_key_value_heads=cfg.num_kv_heads, rms_norm_eps=cfg.rms_eps, attn_implementation="sdpa")
    if cfg.qkv_bias:
        from transformers import Qwen2Config, Qwen2Model
        hf_cfg = Qwen2Config(max_position_embeddings=32768, rope_parameters={"rope_type": "default", "rope_theta": cfg.rope_theta},
                             use_sliding_window=False, **common)
        cls = Qwen2Model
    else:
        from transformers import LlamaConfig, LlamaModel
        hf_cfg = LlamaConfig(head_dim=cfg.head_dim, max_position_embeddings=131072,
                             rope_parameters={"rope_type": "llama3", "rope_theta": cfg.rope_theta, "factor": cfg.rope_factor,
                                              "low_freq_factor": cfg.rope_low_freq_factor, "high_freq_factor": cfg.rope_high_freq_factor,
                                              "original_max_position_embeddings": cfg.rope_original_max_position}, **common)
        cls = LlamaModel
    try:                                             # skip the random initialisation of up to 8 G parameters (the caller loads every one)
        from transformers.initialization import no_init_weights
    except ImportError:
        from contextlib import nullcontext as no_init_weights
    with torch.device("cuda"), no_init_weights():
        return cls(hf_cfg).to(dtype).eval()


def document_lengths(n_docs: int, seed: int, max_len: int = 512):
    """mixed lengths incl. 1, 2, max_len - 1, max_len"""
    fixed = [max_len, 1, 2, max_len - 1, 129, 300, 64, 17][:n_docs]
    rng = np.random.default_rng(1000 + seed)
    return fixed + rng.integers(3, max_len - 1, size=max(0, n_docs - len(fixed))).tolist()


def documents(cfg, lens, seed: int, first_token=None):
    g = torch.Generator().manual_seed(2024 + seed)
    ids = torch.randint(1000, cfg.vocab_size - 1000, (sum(lens),), generator=g, dtype=torch.int64).to(torch.int32)
    cu = np.concatenate([[0], np.cumsum(lens)])
    if first_token is not None:
        ids[torch.from_numpy(cu[:-1])] = int(first_token)
    return ids.cuda(), torch.tensor(cu, dtype=torch.int32).cuda()


@torch.no_grad()
def hf_pooled(hf, ids, cu):
    """final-norm hidden state of every document's last token, fp32 [B, H] (one document per forward: no padding, no mask)"""
    rows = []
    for b in range(cu.numel() - 1):
        rows.append(hf(input_ids=ids[cu[b]:cu[b + 1]].long()[None], use_cache=False).last_hidden_state[0, -1].float())
    return torch.stack(rows)


def gaps(a, b):
    """1 - cos per row (fp64 on the host side of the comparison)"""
    a, b = a.double(), b.double()
    return 1.0 - (a * b).sum(-1) / (a.norm(dim=-1) * b.norm(dim=-1))


def summary(g):
    g = g.cpu().numpy()
    return {"max": float(g.max()), "p999": float(np.quantile(g, 0.999)), "p99": float(np.quantile(g, 0.99)), "p50": float(np.median(g)), "argmax": int(g.argmax()),
            "over_1e-3": int((g > 1e-3).sum()), "n": int(g.size)}


def measure(preset: str, seed: int = 0, profile: str = "trained_like", n_docs: int = 64, mrl: int = 256, other_stream: bool = False, synth=None,
            operand_dtype=None) -> dict:
    """One backbone x one weight seed: lrx (default stream mode) and HF bf16 against HF fp32, full embedding and the MRL slice."""
    from lightretriever_amd import EncoderConfig, LrxEncoder, _lib
    from lightretriever_amd.synth import sink_token
    cfg = getattr(EncoderConfig, preset)()
    if operand_dtype:
        cfg = dataclasses.replace(cfg, operand_dtype=operand_dtype)
    enc = LrxEncoder.random_init(cfg, seed=seed, profile=profile, **(synth or {}))
    lens = document_lengths(n_docs, seed)
    ids, cu = documents(cfg, lens, seed, first_token=sink_token(cfg) if profile == "trained_like" else None)
    _lib.lib().lrx_device_saturation_count(1)
    out = enc.encode_packed(ids, cu, 512)
    assert torch.equal(out, enc.encode_packed(ids, cu, 512))
    out_mrl = enc.encode_packed(ids, cu, 512, out_dim=mrl)
    torch.cuda.synchronize()
    rec = {"preset": preset, "layers": cfg.num_layers, "profile": profile, "seed": seed, "docs": n_docs,
           "stream": "precise_fp32" if enc.precise else "bf16_folded_norm", "operands": enc.operand_mode,
           "fp16_saturations": int(_lib.lib().lrx_device_saturation_count(1))}
    if getattr(enc, "synth_stats", None):
        rec["weights"] = enc.synth_stats["summary"]
    sd = enc.hf_state_dict()
    other = None
    if other_stream:
        del enc
        torch.cuda.empty_cache()
        enc_b = LrxEncoder(dataclasses.replace(cfg, precise_stream=rec["stream"] != "precise_fp32", operand_dtype=None), sd)
        other = enc_b.encode_packed(ids, cu, 512).clone()
        del enc_b
    else:
        del enc
    torch.cuda.empty_cache()
    hf = hf_model_for(cfg)
    missing, unexpected = hf.load_state_dict({k: v.float() for k, v in sd.items()}, strict=False)
    assert not unexpected and all("rotary" in m for m in missing), (missing, unexpected)
    h32 = hf_pooled(hf, ids, cu)
    hf = hf.to(torch.bfloat16)                      # what the reference executes (bf16 weights and activations)
    h16 = hf_pooled(hf, ids, cu)
    del hf, sd
    torch.cuda.empty_cache()
    rec["lrx_vs_fp32"] = summary(gaps(out, h32))
    rec["hfbf16_vs_fp32"] = summary(gaps(h16, h32))
    rec["lrx_vs_hfbf16"] = summary(gaps(out, h16))
    rec["lrx_vs_fp32_mrl"] = summary(gaps(out_mrl, h32[:, :mrl]))
    rec["hfbf16_vs_fp32_mrl"] = summary(gaps(h16[:, :mrl], h32[:, :mrl]))
    if other is not None:
        rec["other_stream_vs_fp32"] = summary(gaps(other, h32))
    rec["worst_doc_len"] = int(lens[rec["lrx_vs_fp32"]["argmax"]])
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--presets", default="llama32_1b,qwen25_1_5b")
    ap.add_argument("--profile", default="trained_like")
    ap.add_argument("--seeds", default="0")
    ap.add_argument("--docs", type=int, default=64)
    ap.add_argument("--streams", default="default")
    ap.add_argument("--out", default="")
    ap.add_argument("--synth", default="", help="JSON dict of synth.trained_like_state_dict keyword overrides (experiments)")
    ap.add_argument("--operands", default="", help="bf16 | fp16 | fp16_qkv: GEMM operands of the fp32 stream (default: the encoder's rule)")
    a = ap.parse_args()
    for preset in a.presets.split(","):
        for seed in [int(s) for s in a.seeds.split(",")]:
            rec = measure(preset, seed, a.profile, a.docs, other_stream=a.streams == "both", synth=json.loads(a.synth) if a.synth else None,
                          operand_dtype=a.operands or None)
            if a.synth:
                rec["synth_overrides"] = json.loads(a.synth)
            line = json.dumps(rec)
            print(line, flush=True)
            if a.out:
                with open(a.out, "a") as f:
                    f.write(line + "\n")


if __name__ == "__main__":
    main()
