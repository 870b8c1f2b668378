#!/usr/bin/env python3
"""Golden vectors for the LM-encoded SPARSE QUERY vector (`--hybrid_use_sparse_vector`: the `spr` / `den_spr` query modes of
retriever/hybrid_search.py:160-180), made by running the REAL reference (round 6):
  HybridModel.encode_query with encode_sparse (finetune/modeling_hybrid.py:404-438: LM forward -> LM head -> max aggregation over the sparse
  attention mask -> get_sparse_emb(is_query=True): relu, log1p, top-p / top-k with the *_qry ratios) next to the dense vector of the same
  forward, on the model and batch of gen_sparse_goldens.py (llama_small_d64 weights, seed 5; [SEP] cases included), and
  SparseConverterMixin.convert_sparse_reps_to_pseudo_text_pt -- the reference's torch restatement of the Rust converter call_batch_encode
  applies to query vectors (inference/exact_search_base.py:231-236).

Build container only (needs /root/reference).  Usage: PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_sparse_query_goldens.py"""
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
    sp = np.load(os.path.join(HERE, "sparse.npz"))                    # the batch of gen_sparse_goldens.py (prompt + [SEP] + text rows included)
    ids, mask = sp["input_ids"], sp["attention_mask"]
    tid, tmask = torch.from_numpy(ids), torch.from_numpy(mask)

    def hybrid(**kw):
        hm = G.make_hybrid(lm, tok_dir, pooling_strategy="lasttoken", score_function="cos_sim", hybrid_use_dense_vector=True,
                           hybrid_use_sparse_vector=True, sparse_use_max_aggregation=True, sparse_use_relu=True, sparse_use_log_saturation=True,
                           add_sep_token=True, **kw)
        hm.sep_token_id = SEP
        return hm

    qry = {"input_ids": tid, "attention_mask": tmask, "unique_token_ids": None}
    g = {"input_ids": ids, "attention_mask": mask, "sep_token_id": np.int64(SEP), "weight_seed": np.int64(5)}
    hm = hybrid()
    out = hm.encode_query(qry)
    assert set(out) == {"dense_reps", "sparse_reps"}
    g["sparse_reps"], g["dense_reps"] = out["sparse_reps"].float().numpy(), out["dense_reps"].float().numpy()
    np.testing.assert_array_equal(g["sparse_reps"], sp["sparse_reps"])            # no *_qry ratio set: the passage vector of the same tokens
    g["sparse_reps_top8_qry"] = hybrid(sparse_top_k_qry=8, sparse_min_tokens_to_keep=4, sparse_top_k_psg=16).encode_query(qry)["sparse_reps"].float().numpy()
    g["sparse_reps_topp_qry"] = hybrid(sparse_top_p_qry=0.4, sparse_min_tokens_to_keep=8, sparse_top_p_psg=0.9).encode_query(qry)["sparse_reps"].float().numpy()
    g["sparse_only"] = hybrid().encode_query(qry, encode_dense=False)["sparse_reps"].float().numpy()
    np.savez_compressed(os.path.join(HERE, "sparse_query.npz"), **g)
    txt = {"quant100_row1": hm.convert_sparse_reps_to_pseudo_text_pt(torch.from_numpy(g["sparse_reps"][1:2]), 100, False),     # (one full row: 158 tokens x their weights)
           "quant100_top8": hm.convert_sparse_reps_to_pseudo_text_pt(torch.from_numpy(g["sparse_reps_top8_qry"]), 100, False),
           "quant7_halves": hm.convert_sparse_reps_to_pseudo_text_pt(torch.tensor([[0.5 / 7, 1.5 / 7, 2.5 / 7, -3.0, 0.0, 0.07], [0.0, 0.0, 0.0, 0.0, 0.0, 0.0]]), 7, False)}
    with open(os.path.join(HERE, "sparse_query_text.json"), "w") as f:
        json.dump(txt, f)
    print("nonzeros per query:", [(r > 0).sum() for r in g["sparse_reps"]], "top8:", [(r > 0).sum() for r in g["sparse_reps_top8_qry"]],
          "topp:", [(r > 0).sum() for r in g["sparse_reps_topp_qry"]])
    print("pseudo text of row 1:", txt["quant100_row1"][0][:120], "| empty vector ->", repr(txt["quant7_halves"][1]))


if __name__ == "__main__":
    main()
