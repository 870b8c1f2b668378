#!/usr/bin/env python3
"""Golden vectors for the sparse document-vector row, made by running the REAL reference:
  sparse_pooling.{get_sparse_attention_mask, aggregate, top_k_sampling, top_p_sampling},
  HybridModel.encode_passage(encode_sparse=True) (fp32 and under torch.autocast), and
  SparseConverterMixin.convert_sparse_reps_to_json_pt (the reference's own torch restatement of its Rust converter;
  the Rust crate `sparse_emb_util` itself is not in the reference tree -> converter parity is pinned on the _pt variant only).

Build container only (needs /root/reference).  Usage: PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_sparse_goldens.py"""
import json
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_goldens as G  # noqa: E402  (installs the import shim)

import numpy as np  # noqa: E402
import torch  # noqa: E402
from transformers import LlamaConfig, LlamaForCausalLM  # noqa: E402
from lightretriever.finetune.sparse_pooling import get_sparse_attention_mask, aggregate, top_k_sampling, top_p_sampling  # noqa: E402

torch.set_grad_enabled(False)
SEP = 7


def main():
    tok_dir = os.path.join(HERE, "tok")
    from oracle.lrx_oracle import random_weights
    rope_l3 = {"rope_type": "llama3", "rope_theta": 500000.0, "factor": 32.0, "low_freq_factor": 1.0,
               "high_freq_factor": 4.0, "original_max_position_embeddings": 64}
    V = 290
    cfg = LlamaConfig(vocab_size=V, hidden_size=256, intermediate_size=512, num_hidden_layers=3, num_attention_heads=4,
                      num_key_value_heads=2, head_dim=64, rms_norm_eps=1e-5, rope_parameters=rope_l3,
                      max_position_embeddings=512, tie_word_embeddings=True, attn_implementation="eager")
    lm = LlamaForCausalLM(cfg).eval()
    ocfg = G.hf_to_cfg(cfg, LlamaForCausalLM)
    wnp = random_weights(ocfg, seed=5, std=0.05, bf16=True)          # == the llama_small_d64 fixture weights
    missing, unexpected = lm.model.load_state_dict({k: torch.from_numpy(v) for k, v in wnp.items()}, strict=False)
    assert not unexpected
    assert lm.lm_head.weight.data_ptr() == lm.model.embed_tokens.weight.data_ptr()      # tied head

    rng = np.random.default_rng(77)
    lens = [40, 3, 2, 17, 33, 1, 40, 25]
    S = 40
    ids, mask = G.ragged_batch(rng, len(lens), S, V, lens)
    ids[ids == SEP] = SEP + 1
    ids[0, 5] = SEP                      # prompt(5) + sep + text
    ids[3, 2] = SEP
    ids[3, 9] = SEP                      # two seps: the first one counts
    ids[4, 32] = SEP                     # sep only at the last valid position
    ids[6, 39] = SEP                     # sep at the last padded column
    tid, tmask = torch.from_numpy(ids), torch.from_numpy(mask)

    g = {"input_ids": ids, "attention_mask": mask, "sep_token_id": np.int64(SEP), "weight_seed": np.int64(5)}
    g["mask_plain"] = get_sparse_attention_mask(tid, tmask, SEP, remove_prompt=False).numpy()
    g["mask_noprompt"] = get_sparse_attention_mask(tid, tmask, SEP, remove_prompt=True).numpy()
    # the `all first-sep positions == last column` quirk of get_prompt_mask (sparse_pooling.py:51-53)
    ids_q = ids[[0, 6]].copy()
    ids_q[0, 5] = SEP + 1
    ids_q[0, 39] = SEP
    g["quirk_ids"] = ids_q
    g["quirk_mask"] = get_sparse_attention_mask(torch.from_numpy(ids_q), tmask[[0, 6]], SEP, remove_prompt=True).numpy()

    hidden = lm.model(input_ids=tid, attention_mask=tmask, return_dict=True, use_cache=False).last_hidden_state
    g["agg_plain"] = aggregate(hidden, lm.lm_head, torch.from_numpy(g["mask_plain"]), True).numpy()
    g["agg_noprompt"] = aggregate(hidden, lm.lm_head, torch.from_numpy(g["mask_noprompt"]), True).numpy()

    def hybrid(**kw):
        hm = G.make_hybrid(lm, tok_dir, pooling_strategy="lasttoken", score_function="cos_sim", hybrid_use_dense_vector=True,
                           hybrid_use_token_id_vector=True, sparse_use_max_aggregation=True, **kw)
        hm.sep_token_id = SEP
        return hm

    psg = {"input_ids": tid, "attention_mask": tmask, "unique_token_ids": None}
    hm = hybrid(sparse_use_relu=True, sparse_use_log_saturation=True, add_sep_token=True)
    out = hm.encode_passage(psg)
    g["sparse_reps"] = out["sparse_reps"].float().numpy()
    g["dense_reps"] = out["dense_reps"].float().numpy()
    with torch.autocast("cpu"):
        g["sparse_reps_autocast"] = hm.encode_passage(psg)["sparse_reps"].float().numpy()
    g["sparse_reps_raw"] = hybrid(add_sep_token=False).encode_passage(psg)["sparse_reps"].float().numpy()   # no relu/log: finfo.min rows
    g["sparse_reps_top16"] = hybrid(sparse_use_relu=True, sparse_use_log_saturation=True, add_sep_token=True, sparse_top_k_psg=16,
                                    sparse_min_tokens_to_keep=8).encode_passage(psg)["sparse_reps"].float().numpy()
    g["sparse_reps_top3_min8"] = hybrid(sparse_use_relu=True, sparse_use_log_saturation=True, add_sep_token=True, sparse_top_k_psg=3,
                                        sparse_min_tokens_to_keep=8).encode_passage(psg)["sparse_reps"].float().numpy()
    g["sparse_reps_topp"] = hybrid(sparse_use_relu=True, sparse_use_log_saturation=True, add_sep_token=True, sparse_top_p_psg=0.3,
                                   sparse_min_tokens_to_keep=8).encode_passage(psg)["sparse_reps"].float().numpy()
    # stand-alone sampling functions on a matrix with exact ties at the threshold
    t = torch.tensor([[0.5, 2.0, 2.0, 0.0, 1.0, 2.0, 3.0, 0.25], [0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0]])
    g["tie_in"] = t.numpy()
    g["tie_top2"] = top_k_sampling(t, 2, min_tokens_to_keep=1).numpy()
    g["tie_topp"] = top_p_sampling(t, 0.5, min_tokens_to_keep=1).numpy()
    np.savez_compressed(os.path.join(HERE, "sparse.npz"), **g)

    conv = {"quant100": hm.convert_sparse_reps_to_json_pt(torch.from_numpy(g["sparse_reps"]), 100, False),
            "quant100_top16": hm.convert_sparse_reps_to_json_pt(torch.from_numpy(g["sparse_reps_top16"]), 100, False),
            "quant7_halves": hm.convert_sparse_reps_to_json_pt(torch.tensor([[0.5 / 7, 1.5 / 7, 2.5 / 7, -3.0, 0.0, 0.07]]), 7, False)}
    with open(os.path.join(HERE, "sparse_json.json"), "w") as f:
        json.dump(conv, f)
    nz = [(r > 0).sum() for r in g["sparse_reps"]]
    print("nonzeros per doc:", nz, " empty rows:", [i for i, n in enumerate(nz) if n == 0])
    print("top16 nonzeros:", [(r > 0).sum() for r in g["sparse_reps_top16"]], "topp:", [(r > 0).sum() for r in g["sparse_reps_topp"]])
    print("max |fp32 - autocast|:", np.abs(g["sparse_reps"] - g["sparse_reps_autocast"]).max())


if __name__ == "__main__":
    main()
