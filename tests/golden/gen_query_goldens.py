#!/usr/bin/env python3
"""Golden vectors for the LM-encoded QUERY representations, made by running the REAL reference (round 5):

  * symmetric dense query vectors  -- HybridModel.encode_query, dense branch (finetune/modeling_hybrid.py:363-401), the flag set of
    eval/README.md:24 (`--hybrid_use_dense_vector`), with and without `dense_shrink_dim`;
  * the LM's input embedding layer as the bag -- `hybrid_use_emb_vector` WITHOUT `noncontextual_query_embedding` (:476-486);
  * both next to the EmbeddingBag vector in one call (`hybrid_use_dense_vector + hybrid_use_emb_vector + noncontextual_query_embedding`);
  * the reference's EncoderModel.encode_query / encode_passage (finetune/modeling_encoder.py:313-401: bare tensors, model_type EncoderModel);

all on the `llama_small_d64` model of gen_goldens.py (weights from the oracle's seeded generator, seed 5: the fixture carries inputs and
outputs only), query batches from the reference's EncodeCollator on the committed synthetic tokenizer (tests/golden/tok), a query prompt
as parse_texts adds it (inference/exact_search_base.py:85-90).  Same import shim as gen_goldens.py; runs only in the build container.

Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_query_goldens.py      -> tests/golden/query_modes.npz
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

from lightretriever.finetune.nonctx_emb_utils import construct_embedding_bag  # noqa: E402
from lightretriever.finetune.modeling_encoder import EncoderModel  # noqa: E402
from lightretriever.finetune.arguments import ModelArguments  # noqa: E402
from lightretriever.inference.exact_search_base import EncodeCollator  # noqa: E402

PROMPT = "Instruct: Given a web search query, retrieve relevant passages that answer the query\nQuery: "


def main():
    from oracle.lrx_oracle import random_weights
    tok_dir = os.path.join(HERE, "tok")
    tok = PreTrainedTokenizerFast.from_pretrained(tok_dir)
    V = len(tok)
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

    queries = [{"text": "what is the capital of france?"}, {"text": "similarity search"}, {"text": "a"},
               {"text": "query that is long enough to be truncated by the q_max_len limit of twenty four tokens for sure yes it is really long"},
               {"text": "dense retrieval with large language models"}]
    prompted = [dict(q, prompt=PROMPT) for q in queries]                          # what parse_texts(queries, prompt=query_prompt) hands the collator
    q_max_len = 40
    out = {"q_max_len": np.int64(q_max_len), "shrink": np.int64(64)}
    import json
    meta = {"queries": queries, "prompt": PROMPT}

    # ---- collators: LM inputs only (symmetric / ablation) and LM inputs + EmbeddingBag fields (noncontextual_query_embedding)
    coll_lm = EncodeCollator(tokenizer=tok, encode_is_query=True, q_max_len=q_max_len, p_max_len=64, noncontextual_query_embedding=False)
    coll_bag = EncodeCollator(tokenizer=tok, encode_is_query=True, q_max_len=q_max_len, p_max_len=64, noncontextual_query_embedding=True)
    b_lm, b_bag = coll_lm(prompted), coll_bag(prompted)
    assert torch.equal(b_lm["input_ids"], b_bag["input_ids"])
    out["input_ids"], out["attention_mask"] = b_lm["input_ids"].numpy(), b_lm["attention_mask"].numpy()
    out["nonctx_ids"], out["nonctx_offsets"] = b_bag["nonctx_tok_emb_input_ids"].numpy(), b_bag["nonctx_tok_emb_offsets"].numpy()
    b_noprompt = coll_lm(queries)                                                  # no prompt column: the bare query text with specials
    out["input_ids_noprompt"], out["attention_mask_noprompt"] = b_noprompt["input_ids"].numpy(), b_noprompt["attention_mask"].numpy()

    common = dict(pooling_strategy="lasttoken", score_function="cos_sim", hybrid_use_sparse_vector=False)
    lm_in = {"input_ids": b_lm["input_ids"], "attention_mask": b_lm["attention_mask"]}

    # ---- symmetric dense (eval/README.md:24), fp32 model called directly (pure fp32, like `dense_reps` of the passage goldens)
    hm = G.make_hybrid(lm, tok_dir, hybrid_use_dense_vector=True, **common)
    r = hm.encode_query(dict(lm_in))
    assert set(r) == {"dense_reps"}
    out["dense_reps"] = r["dense_reps"].float().numpy()
    out["dense_reps_autocast"] = G.call_batch_encode(hm, dict(lm_in), True, {})["dense_reps"].float().numpy()
    hm_mrl = G.make_hybrid(lm, tok_dir, hybrid_use_dense_vector=True, dense_shrink_dim=64, **common)
    out["dense_reps_mrl"] = hm_mrl.encode_query(dict(lm_in))["dense_reps"].float().numpy()
    out["dense_reps_noprompt"] = hm.encode_query({"input_ids": b_noprompt["input_ids"], "attention_mask": b_noprompt["attention_mask"]})["dense_reps"].float().numpy()
    # the documents of the same flag set: encode_passage's dense branch switched on by hybrid_use_dense_vector alone (modeling_hybrid.py:235)
    docs = [{"title": "Paris", "text": "Paris is the capital and most populous city of France."}, {"text": "the quick brown fox"}]
    b_doc = EncodeCollator(tokenizer=tok, encode_is_query=False, q_max_len=q_max_len, p_max_len=64)(docs)
    meta["docs"] = docs
    out["doc_input_ids"], out["doc_attention_mask"] = b_doc["input_ids"].numpy(), b_doc["attention_mask"].numpy()
    out["doc_dense_reps"] = hm.encode_passage({"input_ids": b_doc["input_ids"], "attention_mask": b_doc["attention_mask"]})["dense_reps"].float().numpy()

    # ---- the LM's input embedding layer as the bag (hybrid_use_emb_vector, noncontextual_query_embedding=False)
    hm_e = G.make_hybrid(lm, tok_dir, hybrid_use_dense_vector=False, hybrid_use_emb_vector=True, noncontextual_query_embedding=False, **common)
    r = hm_e.encode_query(dict(lm_in))
    assert set(r) == {"emb_reps"}
    out["emb_reps_lm_embedding"] = r["emb_reps"].float().numpy()
    hm_e_mrl = G.make_hybrid(lm, tok_dir, hybrid_use_dense_vector=False, hybrid_use_emb_vector=True, noncontextual_query_embedding=False,
                             dense_shrink_dim=64, **common)
    out["emb_reps_lm_embedding_mrl"] = hm_e_mrl.encode_query(dict(lm_in))["emb_reps"].float().numpy()

    # ---- everything at once: dense + EmbeddingBag vector (the released checkpoints' model_args.yaml: both flags on)
    bag = construct_embedding_bag(lm.model, tok, prompt=PROMPT, batch_size=97)
    hm_b = G.make_hybrid(lm, tok_dir, hybrid_use_dense_vector=True, hybrid_use_emb_vector=True, noncontextual_query_embedding=True, **common)
    hm_b.emb_bag = bag
    r = hm_b.encode_query({"input_ids": b_bag["input_ids"], "attention_mask": b_bag["attention_mask"],
                           "nonctx_tok_emb_input_ids": b_bag["nonctx_tok_emb_input_ids"], "nonctx_tok_emb_offsets": b_bag["nonctx_tok_emb_offsets"]})
    assert set(r) == {"dense_reps", "emb_reps"}
    np.testing.assert_array_equal(r["dense_reps"].float().numpy(), out["dense_reps"])
    out["emb_reps_bag"] = r["emb_reps"].float().numpy()

    # ---- model_type EncoderModel: the same encoder, bare tensors (finetune/modeling_encoder.py:313-401)
    em = EncoderModel(lm_q=lm.model, lm_p=lm.model, model_args=ModelArguments(model_name_or_path=tok_dir, pooling_strategy="lasttoken", score_function="cos_sim")).eval()
    eq = em.encode_query(dict(lm_in))
    assert isinstance(eq, torch.Tensor)
    np.testing.assert_allclose(eq.float().numpy(), out["dense_reps"], atol=1e-6)
    out["encoder_model_query"] = eq.float().numpy()

    out["meta_json"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "query_modes.npz"), **out)
    for k, v in out.items():
        print(k, getattr(v, "shape", v))


if __name__ == "__main__":
    main()
