#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REAL reference.

Runs only in the build container (needs /root/reference and HF transformers); the outputs
(*.npz, tok/) are committed, this script is committed next to them, and nothing here travels
to the GPU box in executable form other than as documentation of how the vectors were made.

What is executed:
  * reference code (imported from /root/reference/src): dense_pooling.pooling,
    nested_input.unpad_to_seqlen_dim, nonctx_emb_utils.{tokenize_nonctx_qry_emb_bag,
    construct_embedding_bag}, modeling_hybrid.HybridModel.{encode_passage,encode_query},
    exact_search_base.{EncodeCollator, call_batch_encode}
  * the third-party stack it delegates to: HF transformers Llama/Qwen2 (eager/sdpa attention, CPU),
    torch.nn.EmbeddingBag, torch.matmul/topk (stand-in for faiss.IndexFlatIP, which is not installed).

Import shim: stub `peft` / `sparse_emb_util` packages in a temp dir,
transformers-5 tokenization_utils alias patch, no bytecode written into the reference mount.

Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_goldens.py
"""
import os
import sys
import tempfile

sys.dont_write_bytecode = True
os.environ.setdefault("TOKENIZERS_PARALLELISM", "false")
HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/src"
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))


def _install_shim():
    d = tempfile.mkdtemp(prefix="lrx_shim_")
    os.makedirs(os.path.join(d, "peft"))
    os.makedirs(os.path.join(d, "sparse_emb_util"))
    with open(os.path.join(d, "peft", "__init__.py"), "w") as f:
        f.write("import torch.nn as nn\n"
                "class PeftModel(nn.Module): pass\n"
                "class PeftMixedModel(nn.Module): pass\n"
                "class LoraConfig: pass\n"
                "class TaskType: CAUSAL_LM='CAUSAL_LM'; FEATURE_EXTRACTION='FEATURE_EXTRACTION'\n"
                "def get_peft_model(*a, **k): raise NotImplementedError\n")
    with open(os.path.join(d, "peft", "utils.py"), "w") as f:
        f.write("CONFIG_NAME='adapter_config.json'\n")
    with open(os.path.join(d, "sparse_emb_util", "__init__.py"), "w") as f:
        f.write("class ICUWordPreTokenizer:\n    def __init__(self,*a,**k): pass\n"
                "class Converter:\n    def __init__(self,*a,**k): pass\n")
    sys.path.insert(0, d)
    sys.path.insert(0, REF)
    import importlib
    from transformers import tokenization_utils_base as tub
    m = importlib.import_module("transformers.tokenization_utils_sentencepiece")
    for n in ["PreTrainedTokenizerBase", "BatchEncoding", "PaddingStrategy"]:
        setattr(m, n, getattr(tub, n))


_install_shim()

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from transformers import LlamaConfig, LlamaForCausalLM, Qwen2Config, Qwen2ForCausalLM, PreTrainedTokenizerFast  # noqa: E402

from lightretriever.finetune.dense_pooling import pooling  # noqa: E402
from lightretriever.utils.nested_input import unpad_to_seqlen_dim  # noqa: E402
from lightretriever.finetune.nonctx_emb_utils import tokenize_nonctx_qry_emb_bag, construct_embedding_bag  # noqa: E402
from lightretriever.finetune.modeling_hybrid import HybridModel  # noqa: E402
from lightretriever.finetune.modeling_encoder import EncoderModel  # noqa: E402
from lightretriever.finetune.arguments import ModelArguments  # noqa: E402
from lightretriever.inference.exact_search_base import EncodeCollator, call_batch_encode  # noqa: E402

torch.manual_seed(0)
torch.set_grad_enabled(False)


def sd_to_np(model):
    """state_dict of the inner `model.` (LlamaModel) as fp32 numpy with HF names minus the prefix."""
    out = {}
    for k, v in model.model.state_dict().items():
        out["w." + k] = v.detach().float().numpy()
    return out


def ragged_batch(rng, B, S, V, lens, pad_id=0):
    ids = np.full((B, S), pad_id, dtype=np.int64)
    mask = np.zeros((B, S), dtype=np.int64)
    for b, n in enumerate(lens):
        ids[b, :n] = rng.integers(3, V, size=n)
        mask[b, :n] = 1
    return ids, mask


def make_hybrid(lm, tokenizer_dir, **margs):
    """HybridModel without its from-disk tokenizer load in __init__ (modeling_hybrid.py:119): build via
    __new__ + EncoderModel.__init__ + the attributes encode_passage/encode_query read."""
    args = ModelArguments(model_name_or_path=tokenizer_dir, **margs)
    hm = HybridModel.__new__(HybridModel)
    EncoderModel.__init__(hm, lm_q=lm, lm_p=lm, model_args=args)
    hm.emb_bag = None
    hm.emb_bag_prompt = None
    hm.spr_pooler_q = None
    hm.spr_pooler_p = None
    hm.sep_token_id = None
    hm.eos_token_id = None
    hm.reg_scaling_factor = 1.0
    return hm.eval()


def hf_to_cfg(cfg, cls):
    from oracle.lrx_oracle import EncoderConfig
    rp = getattr(cfg, "rope_parameters", None) or {}
    rt = rp.get("rope_type", "default")
    return EncoderConfig(
        vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, num_layers=cfg.num_hidden_layers,
        num_q_heads=cfg.num_attention_heads, num_kv_heads=cfg.num_key_value_heads,
        head_dim=getattr(cfg, "head_dim", None) or cfg.hidden_size // cfg.num_attention_heads,
        intermediate_size=cfg.intermediate_size, rms_eps=cfg.rms_norm_eps,
        rope_theta=float(rp.get("rope_theta", 10000.0)), rope_type=rt,
        rope_factor=float(rp.get("factor", 1.0)), rope_low_freq_factor=float(rp.get("low_freq_factor", 1.0)),
        rope_high_freq_factor=float(rp.get("high_freq_factor", 4.0)),
        rope_original_max_position=int(rp.get("original_max_position_embeddings", 8192)),
        qkv_bias=bool(cls is Qwen2ForCausalLM), max_positions=512)


def gen_model_golden(name, cfg, cls, lens, S, seed, tokenizer_dir, shrink=16, store_weights=False, store_hidden=True):
    from oracle.lrx_oracle import random_weights
    from dataclasses import asdict
    lm = cls(cfg).eval()
    ocfg = hf_to_cfg(cfg, cls)
    # weights come from the oracle's seeded numpy generator (bf16-representable, non-trivial norm weights/biases) so the
    # fixture only has to carry the seed; the bf16 and fp32 HF models share identical values
    wnp = random_weights(ocfg, seed=seed, std=0.05, bf16=True)
    sd = {k: torch.from_numpy(v) for k, v in wnp.items()}
    missing, unexpected = lm.model.load_state_dict(sd, strict=False)
    assert not unexpected and all("rotary" in m for m in missing), (missing, unexpected)
    rng = np.random.default_rng(seed)
    V = cfg.vocab_size
    ids, mask = ragged_batch(rng, len(lens), S, V, lens)
    tid, tmask = torch.from_numpy(ids), torch.from_numpy(mask)
    out = lm.model(input_ids=tid, attention_mask=tmask, return_dict=True, use_cache=False, output_hidden_states=True)
    last = out.last_hidden_state
    pooled = pooling(last_hidden=last, attention_mask=tmask, pooling_strategy="lasttoken")
    dense = F.normalize(pooled, p=2, dim=-1)
    dense_mrl = F.normalize(pooled[..., :shrink], p=2, dim=-1)

    # through the reference's own operator boundary (B3): HybridModel.encode_passage via call_batch_encode
    hm = make_hybrid(lm, tokenizer_dir, pooling_strategy="lasttoken", score_function="cos_sim",
                     hybrid_use_dense_vector=True, hybrid_use_sparse_vector=False)
    # encode_passage called directly = pure fp32; call_batch_encode wraps it in torch.autocast (bf16 matmuls on
    # an fp32 model, exact_search_base.py:211) -> kept as a separate golden
    ref_dense = hm.encode_passage({"input_ids": tid, "attention_mask": tmask})["dense_reps"].float()
    assert torch.allclose(ref_dense, dense, atol=1e-6), (ref_dense - dense).abs().max()
    ref_autocast = call_batch_encode(hm, {"input_ids": tid, "attention_mask": tmask}, False, {})["dense_reps"].float()
    hm_mrl = make_hybrid(lm, tokenizer_dir, pooling_strategy="lasttoken", score_function="cos_sim",
                         hybrid_use_dense_vector=True, hybrid_use_sparse_vector=False, dense_shrink_dim=shrink)
    ref_mrl = hm_mrl.encode_passage({"input_ids": tid, "attention_mask": tmask})["dense_reps"].float()
    assert torch.allclose(ref_mrl, dense_mrl, atol=1e-6)

    # bf16 model on CPU (what the reference runs under --bf16): same weights, bf16 arithmetic
    lm16 = cls(cfg).eval()
    lm16.load_state_dict(lm.state_dict())
    lm16 = lm16.to(torch.bfloat16)
    hm16 = make_hybrid(lm16, tokenizer_dir, pooling_strategy="lasttoken", score_function="cos_sim",
                       hybrid_use_dense_vector=True, hybrid_use_sparse_vector=False)
    dense16 = call_batch_encode(hm16, {"input_ids": tid, "attention_mask": tmask}, False, {})["dense_reps"].float()

    g = sd_to_np(lm) if store_weights else {}
    g["weight_seed"] = np.int64(seed)
    g["weight_std"] = np.float64(0.05)
    g.update({
        "input_ids": ids, "attention_mask": mask,
        "last_hidden_state": last.numpy() if store_hidden else np.zeros(0), "pooled": pooled.numpy(),
        "dense_reps": ref_dense.numpy(), "dense_reps_mrl": ref_mrl.numpy(), "dense_reps_bf16": dense16.numpy(),
        "dense_reps_autocast": ref_autocast.numpy(),
        "layer_hidden": np.stack([h.numpy() for h in out.hidden_states[1:-1]]) if (cfg.num_hidden_layers > 1 and store_hidden) else np.zeros(0),
        "shrink": np.int64(shrink),
    })
    cfgd = asdict(ocfg)
    for k, v in cfgd.items():
        g["cfg." + k] = np.array(v)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **g)
    print(f"[{name}] dense {tuple(ref_dense.shape)}  max|fp32-bf16|={float((ref_dense - dense16).abs().max()):.3e} "
          f"min cos(fp32,bf16)={float((ref_dense * dense16).sum(-1).min()):.6f}")
    return lm, cfgd


def build_tokenizer(out_dir):
    """Synthetic byte-level BPE tokenizer with the special tokens/ template the released models use
    (utils/data_utils.py:29-117: pad/sep reserved tokens, `<bos> A <eos>` template, right padding)."""
    from tokenizers import Tokenizer, models, pre_tokenizers, decoders, trainers, processors, normalizers
    tok = Tokenizer(models.BPE())
    tok.normalizer = normalizers.Lowercase()
    tok.pre_tokenizer = pre_tokenizers.ByteLevel(add_prefix_space=False)
    tok.decoder = decoders.ByteLevel()
    corpus = [
        "the quick brown fox jumps over the lazy dog", "dense retrieval with large language models",
        "lightretriever encodes documents with a full llm and queries with an embedding bag",
        "instruct: given a web search query, retrieve relevant passages that answer the query",
        "query: what is the capital of france? paris is the capital and most populous city of france",
        "amd instinct mi355x accelerators use hbm3e memory and cdna4 matrix cores",
        "similarity search over one million documents with inner product top-k", "0123456789 !?.,;:-_()[]{}",
    ] * 4
    specials = ["<|begin_of_text|>", "<|end_of_text|>", "<|reserved_special_token_0|>", "<|reserved_special_token_1|>"]
    trainer = trainers.BpeTrainer(vocab_size=290, special_tokens=specials, initial_alphabet=pre_tokenizers.ByteLevel.alphabet())
    tok.train_from_iterator(corpus, trainer)
    bos, eos = tok.token_to_id(specials[0]), tok.token_to_id(specials[1])
    tok.post_processor = processors.TemplateProcessing(
        single=f"{specials[0]} $A {specials[1]}", pair=f"{specials[0]} $A {specials[1]} $B:1 {specials[1]}:1",
        special_tokens=[(specials[0], bos), (specials[1], eos)])
    fast = PreTrainedTokenizerFast(tokenizer_object=tok, bos_token=specials[0], eos_token=specials[1],
                                   pad_token=specials[2], sep_token=specials[3], padding_side="right")
    os.makedirs(out_dir, exist_ok=True)
    fast.save_pretrained(out_dir)
    return fast


def main():
    tok_dir = os.path.join(HERE, "tok")
    tok = build_tokenizer(tok_dir)
    V = len(tok)
    print("tokenizer len", V)

    # ---- G1: tiny llama, llama3 rope; ragged lengths incl. 1 and full-length row
    rope_l3 = {"rope_type": "llama3", "rope_theta": 500000.0, "factor": 32.0, "low_freq_factor": 1.0,
               "high_freq_factor": 4.0, "original_max_position_embeddings": 64}
    cfg1 = LlamaConfig(vocab_size=V, hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4,
                       num_key_value_heads=2, head_dim=16, rms_norm_eps=1e-5, rope_parameters=rope_l3,
                       max_position_embeddings=512, tie_word_embeddings=True, attn_implementation="eager")
    lm1, cfgd1 = gen_model_golden("llama_tiny_l3rope", cfg1, LlamaForCausalLM, lens=[24, 1, 7, 24, 13, 2], S=24, seed=1,
                                  tokenizer_dir=tok_dir, store_weights=True)
    # ---- G1b: default rope, all rows full length (hits the left_padding branch dense_pooling.py:49-51), longer S
    cfg1b = LlamaConfig(vocab_size=V, hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4,
                        num_key_value_heads=2, head_dim=16, rms_norm_eps=1e-5,
                        rope_parameters={"rope_type": "default", "rope_theta": 500000.0},
                        max_position_embeddings=512, tie_word_embeddings=True, attn_implementation="eager")
    gen_model_golden("llama_tiny_allfull", cfg1b, LlamaForCausalLM, lens=[40, 40, 40], S=40, seed=2, tokenizer_dir=tok_dir)
    # ---- G2: Qwen2-tiny (qkv bias, theta 1e6)
    cfg2 = Qwen2Config(vocab_size=V, hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4,
                       num_key_value_heads=2, rms_norm_eps=1e-6, rope_parameters={"rope_type": "default", "rope_theta": 1000000.0},
                       max_position_embeddings=512, tie_word_embeddings=True, attn_implementation="eager",
                       use_sliding_window=False)
    gen_model_golden("qwen2_tiny", cfg2, Qwen2ForCausalLM, lens=[19, 5, 33, 1], S=33, seed=3, tokenizer_dir=tok_dir)
    # ---- G3: head_dim 128 / 64 variants at kernel-friendly sizes (H=256; heads 2/1 d=128 ; heads 4/2 d=64), longer seqs
    cfg3 = LlamaConfig(vocab_size=V, hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2,
                       num_key_value_heads=1, head_dim=128, rms_norm_eps=1e-5, rope_parameters=dict(rope_l3, factor=8.0),
                       max_position_embeddings=512, tie_word_embeddings=True, attn_implementation="eager")
    gen_model_golden("llama_small_d128", cfg3, LlamaForCausalLM, lens=[70, 33, 128, 1, 65], S=128, seed=4, tokenizer_dir=tok_dir,
                     shrink=64, store_hidden=False)
    cfg4 = LlamaConfig(vocab_size=V, hidden_size=256, intermediate_size=512, num_hidden_layers=3, num_attention_heads=4,
                       num_key_value_heads=2, head_dim=64, rms_norm_eps=1e-5, rope_parameters=rope_l3,
                       max_position_embeddings=512, tie_word_embeddings=True, attn_implementation="eager")
    lm4, cfgd4 = gen_model_golden("llama_small_d64", cfg4, LlamaForCausalLM, lens=[100, 64, 31, 130, 2, 97, 32, 33], S=130, seed=5,
                                  tokenizer_dir=tok_dir, shrink=64, store_hidden=False)

    # ---- G4: packing known-answer (the toy of utils/nested_input.py:195-209 shape: 3x5 ragged) + a random one
    ids = torch.tensor([[11, 12, 13, 14, 15], [21, 22, 0, 0, 0], [31, 32, 33, 0, 0]])
    mask = torch.tensor([[1, 1, 1, 1, 1], [1, 1, 0, 0, 0], [1, 1, 1, 0, 0]])
    nested, pos, indices, bsz, slen = unpad_to_seqlen_dim(ids, mask)
    rng = np.random.default_rng(9)
    ids2, mask2 = ragged_batch(rng, 6, 17, 1000, [17, 3, 9, 1, 17, 12])
    n2, p2, i2, _, _ = unpad_to_seqlen_dim(torch.from_numpy(ids2), torch.from_numpy(mask2))
    np.savez_compressed(os.path.join(HERE, "packing.npz"),
                        ids=ids.numpy(), mask=mask.numpy(), nested=nested.numpy()[0], pos=pos.numpy()[0], indices=indices.numpy(),
                        ids2=ids2, mask2=mask2, nested2=n2.numpy()[0], pos2=p2.numpy()[0], indices2=i2.numpy())

    # ---- G5: tokenizer/collator contract (exact_search_base.py:328-437, nonctx_emb_utils.py:197-219)
    docs = [
        {"title": "Paris", "text": "Paris is the capital and most populous city of France."},
        {"title": "", "text": "dense retrieval with large language models"},
        {"text": "AMD Instinct MI355X accelerators use HBM3E memory " * 6},
        {"title": "T", "text": "x", "prompt": "passage: "},
        {"text": "the quick brown fox"},
    ]
    docs_noprompt = [d for d in docs if "prompt" not in d]
    coll_p = EncodeCollator(tokenizer=tok, encode_is_query=False, q_max_len=16, p_max_len=32)
    enc_p = coll_p(docs_noprompt)
    enc_p_prompt = coll_p([docs[3]])
    queries = [{"text": "what is the capital of france?"}, {"text": "similarity search"}, {"text": "a"},
               {"text": "query that is long enough to be truncated by the q_max_len limit of sixteen tokens for sure yes"}]
    coll_q = EncodeCollator(tokenizer=tok, encode_is_query=True, q_max_len=16, p_max_len=32, noncontextual_query_embedding=True)
    enc_q = coll_q(queries)
    import json
    with open(os.path.join(HERE, "collator.json"), "w") as f:
        json.dump({
            "docs": docs_noprompt, "doc_prompted": docs[3], "queries": queries, "p_max_len": 32, "q_max_len": 16,
            "doc_input_ids": enc_p["input_ids"].tolist(), "doc_attention_mask": enc_p["attention_mask"].tolist(),
            "doc_prompted_input_ids": enc_p_prompt["input_ids"].tolist(),
            "qry_nonctx_input_ids": enc_q["nonctx_tok_emb_input_ids"].tolist(),
            "qry_nonctx_offsets": enc_q["nonctx_tok_emb_offsets"].tolist(),
            "bos": tok.bos_token_id, "eos": tok.eos_token_id, "pad": tok.pad_token_id, "sep": tok.sep_token_id,
        }, f, indent=1)

    # ---- G6: EmbeddingBag built by the reference's construct_embedding_bag on the d64 small model + encode_query
    prompt = "Instruct: Given a web search query, retrieve relevant passages that answer the query\nQuery: "
    bag = construct_embedding_bag(lm4.model, tok, prompt=prompt, batch_size=97)
    table = bag.weight.detach().float().numpy()
    prompt_ids = tok.encode(prompt, add_special_tokens=False)
    hmq = make_hybrid(lm4, tok_dir, pooling_strategy="lasttoken", score_function="cos_sim", hybrid_use_dense_vector=False,
                      hybrid_use_sparse_vector=False, hybrid_use_emb_vector=True, noncontextual_query_embedding=True)
    hmq.emb_bag = bag
    qb = {"input_ids": enc_q["input_ids"], "attention_mask": enc_q["attention_mask"],
          "nonctx_tok_emb_input_ids": enc_q["nonctx_tok_emb_input_ids"], "nonctx_tok_emb_offsets": enc_q["nonctx_tok_emb_offsets"]}
    qout = call_batch_encode(hmq, qb, True, {})
    emb_reps = qout["emb_reps"].float().numpy()
    # a bag input with padding ids and an empty bag, straight through torch.nn.EmbeddingBag
    pad = tok.pad_token_id
    ids_b = torch.tensor([5, 9, pad, 7, 7, 100, pad, pad, 33], dtype=torch.long)
    offs_b = torch.tensor([0, 3, 3, 6], dtype=torch.long)   # bag 1 is empty, bag 3 has only... [pad,pad,33]
    raw_b = bag(ids_b, offs_b).float().numpy()
    np.savez_compressed(os.path.join(HERE, "embbag.npz"), table=table, prompt_ids=np.array(prompt_ids),
                        bos=np.int64(tok.bos_token_id), eos=np.int64(tok.eos_token_id), pad=np.int64(pad),
                        q_ids=enc_q["nonctx_tok_emb_input_ids"].numpy(), q_offsets=enc_q["nonctx_tok_emb_offsets"].numpy(),
                        emb_reps=emb_reps, ids_b=ids_b.numpy(), offs_b=offs_b.numpy(), raw_b=raw_b)
    print("[embbag] table", table.shape, "emb_reps", emb_reps.shape)

    # ---- G5b: tokenizer surgery contract (utils/data_utils.py:29-271): a RAW byte-level BPE tokenizer (no normaliser,
    #      no template, no pad/sep) -> reference load_tokenizer(lowercase, <bos> A <eos>, pad/sep reserved tokens)
    from tokenizers import Tokenizer, models, pre_tokenizers, decoders, trainers
    from lightretriever.utils.data_utils import load_tokenizer as ref_load_tokenizer
    raw = Tokenizer(models.BPE())
    raw.pre_tokenizer = pre_tokenizers.ByteLevel(add_prefix_space=False)
    raw.decoder = decoders.ByteLevel()
    specials = ["<|begin_of_text|>", "<|end_of_text|>", "<|reserved_special_token_0|>", "<|reserved_special_token_1|>"]
    raw.train_from_iterator(["The Quick Brown Fox Jumps", "Dense Retrieval With LLMs", "Paris Is The Capital Of France"] * 4,
                            trainers.BpeTrainer(vocab_size=300, special_tokens=specials, initial_alphabet=pre_tokenizers.ByteLevel.alphabet()))
    raw_dir = os.path.join(HERE, "tok_raw")
    os.makedirs(raw_dir, exist_ok=True)
    PreTrainedTokenizerFast(tokenizer_object=raw, bos_token=specials[0], eos_token=specials[1]).save_pretrained(raw_dir)
    rt = ref_load_tokenizer(raw_dir, lowercase=True, add_bos_num=1, add_eos_num=1, add_pad_token=True, pad_token=specials[2],
                            add_sep_token=True, sep_token=specials[3])
    texts = ["The Quick BROWN fox", "Paris", "Dense retrieval with LLMs is FUN " * 5]
    enc = rt(texts, max_length=24, truncation="only_first", padding=True, add_special_tokens=True)
    with open(os.path.join(HERE, "tokenizer_surgery.json"), "w") as f:
        json.dump({"texts": texts, "max_length": 24, "input_ids": enc["input_ids"], "attention_mask": enc["attention_mask"],
                   "pad": rt.pad_token_id, "sep": rt.sep_token_id, "bos": rt.bos_token_id, "eos": rt.eos_token_id,
                   "padding_side": rt.padding_side, "nospecial": rt(texts[0], add_special_tokens=False)["input_ids"]}, f, indent=1)

    # ---- G7: flat-IP search goldens: torch.matmul fp32 + topk (faiss not installed -> stand-in, see oracle header)
    g = torch.Generator().manual_seed(7)
    X = F.normalize(torch.randn(5000, 64, generator=g), dim=-1)
    Qm = F.normalize(torch.randn(32, 64, generator=g), dim=-1)
    S = Qm @ X.T
    out = {"X": X.numpy(), "Q": Qm.numpy()}
    for k in (1, 10, 100):
        d, i = torch.topk(S, k, dim=1)
        out[f"D{k}"], out[f"I{k}"] = d.numpy(), i.numpy()
    np.savez_compressed(os.path.join(HERE, "search.npz"), **out)
    print("done")


if __name__ == "__main__":
    main()
