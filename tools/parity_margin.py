#!/usr/bin/env python3
"""Prints the cosine gap of lrx_encode_packed vs HF transformers fp32 AND vs HF bf16 on the same GPU at Llama-3.2-1B dims
(the numbers behind tests/test_gpu_encoder.py::test_full_size_llama32_1b_properties_and_hf_parity)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lightretriever_amd import EncoderConfig, LrxEncoder

def main():
    cfg = EncoderConfig.llama32_1b()
    enc = LrxEncoder.random_init(cfg, seed=0)
    g = torch.Generator().manual_seed(1234)
    lens = [512, 512, 300, 64, 1, 512, 17, 129, 512, 400]
    ids = torch.randint(1000, 127000, (sum(lens),), generator=g, dtype=torch.int64).to(torch.int32).cuda()
    cu = torch.tensor(np.concatenate([[0], np.cumsum(lens)]), dtype=torch.int32).cuda()
    out = enc.encode_packed(ids, cu, 512)
    from transformers import LlamaConfig, LlamaModel
    hf_cfg = LlamaConfig(vocab_size=cfg.vocab_size, hidden_size=2048, intermediate_size=8192, num_hidden_layers=16, num_attention_heads=32,
                         num_key_value_heads=8, head_dim=64, rms_norm_eps=1e-5, max_position_embeddings=131072,
                         rope_parameters={"rope_type": "llama3", "rope_theta": 500000.0, "factor": 32.0, "low_freq_factor": 1.0,
                                          "high_freq_factor": 4.0, "original_max_position_embeddings": 8192}, attn_implementation="sdpa")
    with torch.device("cuda"):
        hf = LlamaModel(hf_cfg).float().eval()
    sd = enc.hf_state_dict()
    hf.load_state_dict({k: v.float() for k, v in sd.items()}, strict=False)
    def run(model, dt):
        refs = []
        with torch.no_grad():
            for b in range(len(lens)):
                h = model(input_ids=ids[cu[b]:cu[b + 1]].long()[None], use_cache=False).last_hidden_state[0, -1]
                refs.append(torch.nn.functional.normalize(h.float(), dim=-1))
        return torch.stack(refs)
    r32 = run(hf, torch.float32)
    hf16 = hf.to(torch.bfloat16)
    r16 = run(hf16, torch.bfloat16)
    print("1-cos(lrx, HF fp32): max %.3e mean %.3e" % ((1 - (r32 * out).sum(-1)).max().item(), (1 - (r32 * out).sum(-1)).mean().item()))
    print("1-cos(HF bf16, HF fp32): max %.3e mean %.3e" % ((1 - (r32 * r16).sum(-1)).max().item(), (1 - (r32 * r16).sum(-1)).mean().item()))
    print("1-cos(lrx, HF bf16): max %.3e" % ((1 - (r16 * out).sum(-1)).max().item()))

if __name__ == "__main__":
    main()
