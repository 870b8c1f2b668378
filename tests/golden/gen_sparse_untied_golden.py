#!/usr/bin/env python3
"""Golden vector for an UNTIED CausalLM head (tie_word_embeddings=false: Llama-3.1-8B, Qwen2.5-7B): the REAL reference's
HybridModel.encode_passage(encode_sparse=True) projects with lm_head (finetune/modeling_hybrid.py:72-86 get_lm_head), not with the
embedding matrix.  Same tiny model and batch as gen_sparse_goldens.py, plus a head drawn from its own seed.

Build container only (needs /root/reference).  Usage: PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_sparse_untied_golden.py"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_goldens as G  # noqa: E402  (installs the import shim)

import numpy as np  # noqa: E402
import torch  # noqa: E402
from transformers import LlamaConfig, LlamaForCausalLM  # noqa: E402

torch.set_grad_enabled(False)
SEP, HEAD_SEED = 7, 91


def head_weights(V, H):
    """bf16-representable N(0, 0.05) head, regenerated from the seed by the test."""
    w = np.random.default_rng(HEAD_SEED).standard_normal((V, H)).astype(np.float32) * np.float32(0.05)
    return torch.from_numpy(w).to(torch.bfloat16).float().numpy()


def main():
    tok_dir = os.path.join(HERE, "tok")
    from oracle.lrx_oracle import random_weights
    rope_l3 = {"rope_type": "llama3", "rope_theta": 500000.0, "factor": 32.0, "low_freq_factor": 1.0,
               "high_freq_factor": 4.0, "original_max_position_embeddings": 64}
    V = 290
    cfg = LlamaConfig(vocab_size=V, hidden_size=256, intermediate_size=512, num_hidden_layers=3, num_attention_heads=4,
                      num_key_value_heads=2, head_dim=64, rms_norm_eps=1e-5, rope_parameters=rope_l3,
                      max_position_embeddings=512, tie_word_embeddings=False, attn_implementation="eager")
    lm = LlamaForCausalLM(cfg).eval()
    ocfg = G.hf_to_cfg(cfg, LlamaForCausalLM)
    wnp = random_weights(ocfg, seed=5, std=0.05, bf16=True)          # == the llama_small_d64 fixture weights
    missing, unexpected = lm.model.load_state_dict({k: torch.from_numpy(v) for k, v in wnp.items()}, strict=False)
    assert not unexpected
    lm.lm_head.weight.copy_(torch.from_numpy(head_weights(V, 256)))
    assert lm.lm_head.weight.data_ptr() != lm.model.embed_tokens.weight.data_ptr()      # untied head

    rng = np.random.default_rng(77)
    lens = [40, 3, 2, 17, 33, 1, 40, 25]
    ids, mask = G.ragged_batch(rng, len(lens), 40, V, lens)
    ids[ids == SEP] = SEP + 1
    ids[0, 5] = SEP
    ids[3, 2] = SEP
    tid, tmask = torch.from_numpy(ids), torch.from_numpy(mask)
    hm = G.make_hybrid(lm, tok_dir, pooling_strategy="lasttoken", score_function="cos_sim", hybrid_use_dense_vector=True,
                       hybrid_use_token_id_vector=True, sparse_use_max_aggregation=True, sparse_use_relu=True, sparse_use_log_saturation=True,
                       add_sep_token=True)
    hm.sep_token_id = SEP
    psg = {"input_ids": tid, "attention_mask": tmask, "unique_token_ids": None}
    out = hm.encode_passage(psg)
    g = {"input_ids": ids, "attention_mask": mask, "sep_token_id": np.int64(SEP), "weight_seed": np.int64(5), "head_seed": np.int64(HEAD_SEED),
         "sparse_reps": out["sparse_reps"].float().numpy(), "dense_reps": out["dense_reps"].float().numpy()}
    with torch.autocast("cpu"):
        g["sparse_reps_autocast"] = hm.encode_passage(psg)["sparse_reps"].float().numpy()
    np.savez_compressed(os.path.join(HERE, "sparse_untied.npz"), **g)
    tied = np.load(os.path.join(HERE, "sparse.npz"))["sparse_reps"]
    print("nonzeros per doc:", [(r > 0).sum() for r in g["sparse_reps"]], " max |untied - tied golden|:", float(np.abs(g["sparse_reps"] - tied).max()))


if __name__ == "__main__":
    main()
