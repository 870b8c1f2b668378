#!/usr/bin/env python3
"""Golden vectors for the pooling strategies other than 'lasttoken', made by running the REAL reference (round 6):

  * `pooling()` itself (finetune/dense_pooling.py:12-82) on random [B, S, H] fp32 tensors with right-padded masks -- 'cls', 'mean',
    'lasttoken', 'second_to_last', 'third_to_last', a ragged batch and an all-full batch (the `left_padding` branch :49-51, :59-61, :72-74);
  * HybridModel.encode_passage / encode_query (finetune/modeling_hybrid.py:205-278, :363-401) with `--pooling_strategy` set to each of
    them, on the `llama_small_d64` model of gen_goldens.py (weights from the oracle's seeded generator, seed 5: the fixture carries inputs
    and outputs only), full width and `dense_shrink_dim = 64`.

'avg_first_last' / 'avg_top2' (:38-46) pool over two entries of the model's `hidden_states` tuple: pooling() gets a random three-entry tuple,
the operators ask HF for the real one (modeling_hybrid.py:257).  Same import shim as gen_goldens.py; runs only in the build container.

Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_pooling_goldens.py      -> tests/golden/pooling.npz
"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import gen_goldens as G  # noqa: E402  (installs the shim, imports the reference)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from transformers import LlamaConfig, LlamaForCausalLM, PreTrainedTokenizerFast  # noqa: E402

from lightretriever.finetune.dense_pooling import pooling  # noqa: E402

STRATEGIES = ("cls", "mean", "lasttoken", "second_to_last", "third_to_last", "avg_first_last", "avg_top2")


def main():
    from oracle.lrx_oracle import random_weights
    out = {}
    # ---- pooling() on random tensors
    rng = np.random.default_rng(61)
    for name, lens, S in (("ragged", [9, 3, 17, 4, 12, 17], 17), ("allfull", [11, 11, 11], 11)):
        B, H = len(lens), 48
        h = rng.standard_normal((B, S, H)).astype(np.float32)
        mask = np.zeros((B, S), dtype=np.int64)
        for b, n in enumerate(lens):
            mask[b, :n] = 1
        out[f"fn_{name}_hidden"], out[f"fn_{name}_mask"] = h, mask
        # hidden_states for the two-layer strategies: (first, middle, last = h), the extra states from their own generator (the other
        # entries of the fixture keep their values)
        rng2 = np.random.default_rng(63 + len(lens))
        hs = (rng2.standard_normal((B, S, H)).astype(np.float32), rng2.standard_normal((B, S, H)).astype(np.float32), h)
        out[f"fn_{name}_hidden_first"], out[f"fn_{name}_hidden_middle"] = hs[0], hs[1]
        for st in STRATEGIES:
            out[f"fn_{name}_{st}"] = pooling(last_hidden=torch.from_numpy(h), hidden_states=tuple(torch.from_numpy(x) for x in hs),
                                            attention_mask=torch.from_numpy(mask), pooling_strategy=st).numpy()

    # ---- through the reference's operators on the llama_small_d64 model
    tok_dir = os.path.join(HERE, "tok")
    V = len(PreTrainedTokenizerFast.from_pretrained(tok_dir))
    rope_l3 = {"rope_type": "llama3", "rope_theta": 500000.0, "factor": 32.0, "low_freq_factor": 1.0,
               "high_freq_factor": 4.0, "original_max_position_embeddings": 64}
    cfg4 = LlamaConfig(vocab_size=V, hidden_size=256, intermediate_size=512, num_hidden_layers=3, num_attention_heads=4,
                       num_key_value_heads=2, head_dim=64, rms_norm_eps=1e-5, rope_parameters=rope_l3,
                       max_position_embeddings=512, tie_word_embeddings=True, attn_implementation="eager")
    lm = LlamaForCausalLM(cfg4).eval()
    ocfg = G.hf_to_cfg(cfg4, LlamaForCausalLM)
    wnp = random_weights(ocfg, seed=5, std=0.05, bf16=True)                      # = the weights of llama_small_d64.npz
    missing, unexpected = lm.model.load_state_dict({k: torch.from_numpy(v) for k, v in wnp.items()}, strict=False)
    assert not unexpected and all("rotary" in m for m in missing)
    lens = [100, 64, 31, 130, 3, 97, 65, 33]                                      # (>= 3 tokens: 'third_to_last' asserts on shorter rows, :70-79)
    ids, mask = G.ragged_batch(np.random.default_rng(62), len(lens), 130, V, lens)
    tid, tmask = torch.from_numpy(ids), torch.from_numpy(mask)
    out["input_ids"], out["attention_mask"], out["shrink"] = ids, mask, np.int64(64)
    for st in STRATEGIES:
        common = dict(pooling_strategy=st, score_function="cos_sim", hybrid_use_dense_vector=True, hybrid_use_sparse_vector=False)
        hm = G.make_hybrid(lm, tok_dir, **common)
        out[f"psg_{st}"] = hm.encode_passage({"input_ids": tid, "attention_mask": tmask})["dense_reps"].float().numpy()
        out[f"qry_{st}"] = hm.encode_query({"input_ids": tid, "attention_mask": tmask})["dense_reps"].float().numpy()
        hm_mrl = G.make_hybrid(lm, tok_dir, dense_shrink_dim=64, **common)
        out[f"psg_{st}_mrl"] = hm_mrl.encode_passage({"input_ids": tid, "attention_mask": tmask})["dense_reps"].float().numpy()
        np.testing.assert_array_equal(out[f"psg_{st}"], out[f"qry_{st}"])           # (one tied encoder, same pooling on both sides)
    np.savez_compressed(os.path.join(HERE, "pooling.npz"), **out)
    for k, v in out.items():
        print(k, getattr(v, "shape", v))


if __name__ == "__main__":
    main()
