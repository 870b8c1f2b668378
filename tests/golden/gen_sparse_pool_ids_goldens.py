#!/usr/bin/env python3
"""Golden vectors for `--sparse_pool_from_original_input_ids_psg / _qry` (finetune/modeling_hybrid.py:175-180: the aggregated logits keep only the
vocabulary entries of the sequence's own tokens under the sparse attention mask -- get_unique_token_ids + get_scores_with_indices,
finetune/sparse_pooling.py:147-179 -- before relu / log1p / top-k: a sparse vector without expansion terms), made by running the REAL
reference: HybridModel.encode_passage and encode_query with the flag on, on the model and batch of gen_sparse_goldens.py (llama_small_d64
weights, seed 5; prompt + [SEP] rows included), plain and with a top-k ratio; the flag of the OTHER side is shown not to leak.

Build container only (needs /root/reference).  Usage: PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_sparse_pool_ids_goldens.py"""
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
    cfg = LlamaConfig(vocab_size=290, hidden_size=256, intermediate_size=512, num_hidden_layers=3, num_attention_heads=4,
                      num_key_value_heads=2, head_dim=64, rms_norm_eps=1e-5, rope_parameters=rope_l3,
                      max_position_embeddings=512, tie_word_embeddings=True, attn_implementation="eager")
    lm = LlamaForCausalLM(cfg).eval()
    wnp = random_weights(G.hf_to_cfg(cfg, LlamaForCausalLM), seed=5, std=0.05, bf16=True)
    missing, unexpected = lm.model.load_state_dict({k: torch.from_numpy(v) for k, v in wnp.items()}, strict=False)
    assert not unexpected
    sp = np.load(os.path.join(HERE, "sparse.npz"))
    ids, mask = sp["input_ids"], sp["attention_mask"]
    batch = {"input_ids": torch.from_numpy(ids), "attention_mask": torch.from_numpy(mask), "unique_token_ids": None}

    def hybrid(**kw):
        hm = G.make_hybrid(lm, tok_dir, pooling_strategy="lasttoken", score_function="cos_sim", hybrid_use_dense_vector=True,
                           hybrid_use_sparse_vector=True, sparse_use_max_aggregation=True, sparse_use_relu=True,
                           sparse_use_log_saturation=True, add_sep_token=True, **kw)
        hm.sep_token_id = SEP
        return hm

    g = {"input_ids": ids, "attention_mask": mask, "sep_token_id": np.int64(SEP), "weight_seed": np.int64(5)}
    g["psg"] = hybrid(sparse_pool_from_original_input_ids_psg=True).encode_passage(batch)["sparse_reps"].float().numpy()
    g["psg_top4"] = hybrid(sparse_pool_from_original_input_ids_psg=True, sparse_top_k_psg=4,
                           sparse_min_tokens_to_keep=2).encode_passage(batch)["sparse_reps"].float().numpy()
    g["qry"] = hybrid(sparse_pool_from_original_input_ids_qry=True).encode_query(batch)["sparse_reps"].float().numpy()
    g["qry_top4"] = hybrid(sparse_pool_from_original_input_ids_qry=True, sparse_top_k_qry=4,
                           sparse_min_tokens_to_keep=2).encode_query(batch)["sparse_reps"].float().numpy()
    # each flag is its own side's: the passage flag leaves queries alone and the other way round
    np.testing.assert_array_equal(hybrid(sparse_pool_from_original_input_ids_psg=True).encode_query(batch)["sparse_reps"].float().numpy(), sp["sparse_reps"])
    np.testing.assert_array_equal(hybrid(sparse_pool_from_original_input_ids_qry=True).encode_passage(batch)["sparse_reps"].float().numpy(), sp["sparse_reps"])
    np.testing.assert_array_equal(g["psg"], g["qry"])
    np.savez_compressed(os.path.join(HERE, "sparse_pool_ids.npz"), **g)
    print("nonzeros per row: all terms", [(r > 0).sum() for r in sp["sparse_reps"]], "own tokens only", [(r > 0).sum() for r in g["psg"]],
          "top4", [(r > 0).sum() for r in g["psg_top4"]])


if __name__ == "__main__":
    main()
