"""Parity of the document encoder on TRAINED-LIKE weight statistics, at the real depth of the released backbones.

Gaussian N(0, 0.02) weights are the easy case for 16-bit arithmetic (q.k logits with sigma < 1, no outlier channel, biases of 0.02).
`LrxEncoder.random_init(profile="trained_like")` (lightretriever_amd/synth.py) draws a model with peaky attention (logit sigma 5-10),
an attention sink, massive-activation channels, heavy-tailed norm weights and -- Qwen -- q/k/v biases of O(10-300); the generator's
own calibration statistics are asserted, so the regime is measured rather than assumed.  The forward being matched is
finetune/modeling_hybrid.py:248-278.  The bar (round 5: ABSOLUTE, for the DEFAULT mode of every backbone -- the mode bench.py times):

    1 - cos(lrx, HF fp32)  <=  1e-3        (max over 64 documents, three weight seeds; full embedding AND the MRL-256 slice)

with no q|k|v element outside fp16's range (lrx_device_saturation_count).  The reference itself runs the forward as HF bf16
(inference/exact_search_base.py:211), whose own distance to fp32 is recorded next to it; only the harsher-amplification case (where HF
bf16 itself is 5e-3 .. 6e-2 away) keeps the relative bar max(1e-3, HF bf16)."""
import glob
import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.gpu
COS_TOL = 1e-3


def _record(rec, name):
    out_dir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out_dir):                                   # (the margins DESIGN.md section 3 quotes come from these files)
        with open(os.path.join(out_dir, name), "a") as f:
            f.write(json.dumps(rec) + "\n")


@pytest.mark.parametrize("seed", [0, 1, 2])
@pytest.mark.parametrize("preset", ["llama31_8b", "qwen25_7b", "llama32_1b", "qwen25_1_5b"])
def test_trained_like_weights_full_depth_parity(preset, seed):
    import parity_margin as pm
    # (the headline model's record also carries the bf16 stream's distance on the same weights -- the mode rounds 1-4 benchmarked, which
    # misses the bar; not asserted)
    rec = pm.measure(preset, seed=seed, profile="trained_like", n_docs=64, other_stream=preset == "llama32_1b")
    _record(rec, "r06_trained_like_parity.jsonl")
    assert rec["stream"] == "precise_fp32", rec["stream"]                          # the default mode of every backbone since round 5
    w = rec["weights"]
    # the regime, as measured by the generator's calibration forward on its own bf16-rounded weights
    assert w["logit_sigma_min"] >= 4.5 and w["top1_prob_mean"] >= 0.5, w           # peaky softmax in every layer
    assert w["sink_mass_mean"] >= 0.2, w                                            # a first-token sink takes a fifth of the attention mass or more
    assert w["stream_max_over_median_max"] >= 200, w                                # massive activations (sink token ~ 1000 x the typical channel)
    if "qwen" in preset:
        assert w["max_abs_bias"] >= 100 and w["max_abs_qkv"] >= 100, w              # Qwen-scale biases: q / k elements of a few hundred
    else:
        assert w["max_abs_qkv"] >= 30, w
    assert rec["fp16_saturations"] == 0, rec                                        # nothing left fp16's range in the fused QKV epilogue
    for full, ref in (("lrx_vs_fp32", "hfbf16_vs_fp32"), ("lrx_vs_fp32_mrl", "hfbf16_vs_fp32_mrl")):
        assert rec[full]["max"] <= COS_TOL, (preset, seed, full, rec[full], rec[ref])     # absolute: north_star's "within 1e-3 cosine"
    print("trained-like %s seed %d (%s): lrx %.2e (p50 %.2e), HF bf16 %.2e; MRL-256 %.2e / %.2e" % (
        preset, seed, rec["stream"], rec["lrx_vs_fp32"]["max"], rec["lrx_vs_fp32"]["p50"], rec["hfbf16_vs_fp32"]["max"],
        rec["lrx_vs_fp32_mrl"]["max"], rec["hfbf16_vs_fp32_mrl"]["max"]))


# (one weight seed per backbone in the default run -- 31 s each on an MI355X; LRX_TEST_WIDE_SEEDS=0,1,... widens it: seeds 0 and 1 are the
# records of profiles/r06_trained_like_wide.jsonl)
@pytest.mark.parametrize("seed", [int(s) for s in os.environ.get("LRX_TEST_WIDE_SEEDS", "0").split(",")])
@pytest.mark.parametrize("preset", ["llama31_8b", "qwen25_7b"])
def test_trained_like_wide_sample_where_the_tail_lives(preset, seed):
    """64 documents read 6-7e-4 for the 8B where the worst of an 8 000-document sample crosses 1e-3 by a
    tenth -- the asserted sample was too small to see the tail.  512 documents per backbone (one HF fp32 pass of ~130 k
    tokens is the cost).  With bf16 GEMM operands the 8B's tail sat AT the bar (512 documents: max 8.1e-4; 2 048: 3 documents over, max 1.11e-3,
    profiles/r06_trained_like_tail_2048.jsonl); the QKV projection's operands are fp16 since (the default, encoder.py: 2 048 documents 2.4e-4),
    and the bar is asserted on EVERY document, full width and MRL slice, with a factor 2 in hand -- and 4 x closer to fp32 than HF's own bf16 run."""
    import parity_margin as pm
    rec = pm.measure(preset, seed=seed, profile="trained_like", n_docs=512)
    _record(rec, "r06_trained_like_wide.jsonl")
    full, mrl, hf16 = rec["lrx_vs_fp32"], rec["lrx_vs_fp32_mrl"], rec["hfbf16_vs_fp32"]
    print("trained-like %s seed %d, 512 documents: lrx p50 %.2e p99 %.2e p99.9 %.2e max %.2e (%d over 1e-3; worst document %d tokens); MRL-256 p99.9 %.2e max %.2e; "
          "HF bf16 p99.9 %.2e max %.2e" % (preset, seed, full["p50"], full["p99"], full["p999"], full["max"], full["over_1e-3"], rec["worst_doc_len"],
                                        mrl["p999"], mrl["max"], hf16["p999"], hf16["max"]))
    assert rec["stream"] == "precise_fp32" and rec["operands"] == "fp16_qkv" and rec["fp16_saturations"] == 0 and full["n"] == 512
    assert full["max"] <= COS_TOL / 2 and mrl["max"] <= COS_TOL / 2 and full["over_1e-3"] == 0, (full, mrl)
    assert full["max"] <= hf16["max"] / 4, (full, hf16)


@pytest.mark.parametrize("preset", ["llama32_1b", "llama31_8b"])
def test_trained_like_weights_harsher_amplification(preset):
    """The one free parameter of the synthetic profile is how strongly a layer amplifies a perturbation of the residual stream (content
    logit spread x size of what a block adds).  The default puts HF's own bf16 run 1e-3 .. 7e-3 from its fp32 run; this case triples
    the amplification (HF bf16 lands at 5e-3 .. 5e-2): the bar relative to HF bf16 must hold there too."""
    import parity_margin as pm
    rec = pm.measure(preset, seed=3, profile="trained_like", n_docs=32,
                     synth={"content_sigma": [1.5, 3.0], "attn_add": 0.25, "mlp_add": 0.35})
    _record(rec, "r06_trained_like_parity.jsonl")
    assert rec["fp16_saturations"] == 0
    assert rec["lrx_vs_fp32"]["max"] <= max(COS_TOL, rec["hfbf16_vs_fp32"]["max"]), rec
    assert rec["lrx_vs_fp32_mrl"]["max"] <= max(COS_TOL, rec["hfbf16_vs_fp32_mrl"]["max"]), rec


def test_saturation_counter_sees_q_beyond_fp16_range():
    """A q bias of 1e5 cannot be stored as fp16: the fused QKV epilogue clamps it to 65504 and must say so."""
    from lightretriever_amd import EncoderConfig, LrxEncoder, _lib
    cfg = EncoderConfig(vocab_size=2000, hidden_size=256, num_layers=2, num_q_heads=4, num_kv_heads=2, head_dim=64, intermediate_size=512,
                        qkv_bias=True, rope_type="default", rope_theta=1e6, max_positions=128)
    enc = LrxEncoder.random_init(cfg, seed=0)
    ids = torch.randint(0, 2000, (70,), dtype=torch.int32).cuda()
    cu = torch.tensor([0, 40, 70], dtype=torch.int32).cuda()
    lib = _lib.lib()
    lib.lrx_device_saturation_count(1)
    enc.encode_packed(ids, cu, 64)
    assert lib.lrx_device_saturation_count(0) == 0
    sd = enc.hf_state_dict()
    b = sd["layers.1.self_attn.q_proj.bias"].clone()
    b[5] = 1e5
    sd["layers.1.self_attn.q_proj.bias"] = b
    bad = LrxEncoder(cfg, sd)
    out = bad.encode_packed(ids, cu, 64)
    assert torch.isfinite(out).all()
    assert lib.lrx_device_saturation_count(1) > 0
    assert lib.lrx_device_saturation_count(0) == 0               # reset


def test_encode_corpus_raises_on_fp16_saturation_and_on_bad_token_ids():
    """the product path reads the device counters once per encode call -- a checkpoint whose q leaves fp16's range
    raises (or warns, by choice) instead of silently returning clamped embeddings; token ids outside the table always raise.  Reference:
    finetune/modeling_hybrid.py:248-278 computes in bf16 (range 3e38) and has no such precondition -- hence loud."""
    import logging
    from transformers import PreTrainedTokenizerFast
    from helpers import GOLDEN
    from lightretriever_amd import EncoderConfig, LrxEncoder, _lib
    from lightretriever_amd.modeling import LrxExactSearchModel, LrxHybridModel
    tok = PreTrainedTokenizerFast.from_pretrained(os.path.join(GOLDEN, "tok"))
    cfg = EncoderConfig(vocab_size=max(2000, len(tok)), hidden_size=256, num_layers=2, num_q_heads=4, num_kv_heads=2, head_dim=64, intermediate_size=512,
                        qkv_bias=True, rope_type="default", rope_theta=1e6, max_positions=128)
    good = LrxEncoder.random_init(cfg, seed=0)
    sd = good.hf_state_dict()
    b = sd["layers.1.self_attn.q_proj.bias"].clone()
    b[5] = 1e5
    sd["layers.1.self_attn.q_proj.bias"] = b
    bad = LrxEncoder(cfg, sd)
    docs = ["the quick brown fox %d jumps over the lazy dog" % i for i in range(20)]
    mk = lambda enc, **kw: LrxExactSearchModel(model=LrxHybridModel(enc, normalize=True, pad_token_id=tok.pad_token_id), tokenizer=tok, q_max_len=32,
                                               p_max_len=64, **kw)
    _lib.lib().lrx_device_saturation_count(1)
    _lib.lib().lrx_device_error_count(1)
    ok = mk(good).encode_corpus(docs, batch_size=8)["dense_reps"]
    assert ok.shape == (20, 256) and torch.isfinite(ok).all()
    with pytest.raises(_lib.LrxError, match="fp16"):
        mk(bad).encode_corpus(docs, batch_size=8)
    assert _lib.lib().lrx_device_saturation_count(0) == 0                     # read AND reset by the failing call
    class _Grab(logging.Handler):
        def __init__(self):
            super().__init__(level=logging.WARNING)
            self.msgs = []
        def emit(self, rec):
            self.msgs.append(rec.getMessage())
    h = _Grab()
    logging.getLogger("lightretriever_amd.modeling").addHandler(h)
    try:
        out = mk(bad, on_fp16_saturation="warn").encode_corpus(docs, batch_size=8)["dense_reps"]
    finally:
        logging.getLogger("lightretriever_amd.modeling").removeHandler(h)
    assert torch.isfinite(out).all() and any("65504" in m for m in h.msgs), h.msgs
    # a token id beyond the embedding table (a tokenizer / checkpoint mismatch the constructor cannot see): always an error
    m = mk(good)
    small = LrxEncoder.random_init(EncoderConfig(vocab_size=64, hidden_size=256, num_layers=2, num_q_heads=4, num_kv_heads=2, head_dim=64,
                                                 intermediate_size=512, rope_type="default", max_positions=128), seed=1)
    m.model = LrxHybridModel(small, normalize=True, pad_token_id=tok.pad_token_id)          # (swapped in behind __post_init__'s vocabulary check)
    with pytest.raises(_lib.LrxError, match="token ids"):
        m.encode_corpus(docs, batch_size=8)
    assert _lib.lib().lrx_device_error_count(0) == 0


def _released_checkpoint():
    """The adapter directory of lightretriever/lightretriever-qwen2.5-1.5b, if it is on this machine (LRX_RELEASED_QWEN25_1_5B or the
    HF hub cache), else None.  The base model is resolved by the loader the same way (local directory or hub cache)."""
    p = os.environ.get("LRX_RELEASED_QWEN25_1_5B")
    if p and os.path.isdir(p):
        return p
    home = os.environ.get("HF_HOME") or os.path.join(os.path.expanduser("~"), ".cache", "huggingface")
    hits = sorted(glob.glob(os.path.join(home, "hub", "models--lightretriever--lightretriever-qwen2.5-1.5b", "snapshots", "*")))
    return hits[-1] if hits else None


def test_released_qwen25_1_5b_reproduces_the_notebook_scores():
    """scripts/asymmetric_dense_infer.ipynb cells 3-12 with the released adapter: EmbeddingBag queries x LM-encoded documents.  The
    notebook prints [[0.3945, 0.0483, 0.3105], [0.0076, 0.3945, 0.0148]] (bf16).  Skipped where the weights are absent (no network
    in the build container or on the GPU box); runs unchanged wherever they are present."""
    path = _released_checkpoint()
    if path is None:
        pytest.skip("lightretriever/lightretriever-qwen2.5-1.5b is not on this machine (set LRX_RELEASED_QWEN25_1_5B)")
    from transformers import AutoTokenizer
    from lightretriever_amd.loader import encoder_from_pretrained
    from lightretriever_amd.modeling import LrxHybridModel
    from lightretriever_amd import ops
    tok = AutoTokenizer.from_pretrained(path)
    enc = encoder_from_pretrained(path, max_positions=512, tokenizer=tok)
    hm = LrxHybridModel(enc, normalize=True, pad_token_id=tok.pad_token_id)
    prompt = "Instruct: Given a web search query, retrieve relevant passages that answer the query\nQuery: "      # scripts/cache_emb_bag.ipynb cell 4
    table = hm.construct_embedding_bag(tok, prompt=prompt, batch_size=1000)
    queries = ["How tall is Mount Everest?", "Who invented the light bulb?"]
    corpus = ["Mount Everest is the highest mountain in the world, about 8,848 meters tall.",
              "Thomas Edison invented the electric light bulb.", "Mount Fuji is the tallest mountain in Japan."]
    q_ids = tok(queries, max_length=512, truncation=True, add_special_tokens=False, return_attention_mask=False)["input_ids"]
    offs = torch.tensor(np.cumsum([0] + [len(t) for t in q_ids[:-1]]), dtype=torch.int64).cuda()
    q = ops.embedding_bag_mean(table, torch.tensor(np.concatenate(q_ids), dtype=torch.int64).cuda(), offs, normalize=True)
    c_ids = tok(corpus, max_length=512, truncation=True, add_special_tokens=True)["input_ids"]
    cu = torch.tensor(np.concatenate([[0], np.cumsum([len(t) for t in c_ids])]), dtype=torch.int32).cuda()
    c = enc.encode_packed(torch.tensor(np.concatenate(c_ids), dtype=torch.int32).cuda(), cu, max(len(t) for t in c_ids))
    scores = (q @ c.T).cpu().numpy()
    want = np.array([[0.3945, 0.0483, 0.3105], [0.0076, 0.3945, 0.0148]])
    # the notebook ran HF bf16 end to end and printed bf16 (3 significant digits): 1e-2 covers its own rounding noise
    np.testing.assert_allclose(scores, want, atol=1e-2)
    assert (scores.argmax(1) == want.argmax(1)).all()



def test_fp16_operands_bring_the_8b_to_five_digits_of_the_fp32_model():
    """`EncoderConfig(operand_dtype="fp16")` (lrx_encoder_config.precise_stream = 2): every projection multiplies fp16 operands -- activations,
    attention / SwiGLU outputs, the four weight matrices.  Llama-3.1-8B at full depth, 64 trained-like documents: every document within 1e-4
    of the HF fp32 model (recorded: 1.9e-5; 2 048 documents: 2.4e-5), nothing saturates."""
    import parity_margin as pm
    rec = pm.measure("llama31_8b", seed=0, profile="trained_like", n_docs=64, operand_dtype="fp16")
    _record(rec, "r06_trained_like_parity.jsonl")
    full, mrl = rec["lrx_vs_fp32"], rec["lrx_vs_fp32_mrl"]
    print("trained-like llama31_8b, fp16 operands: max %.2e (MRL-256 %.2e), HF bf16 %.2e" % (full["max"], mrl["max"], rec["hfbf16_vs_fp32"]["max"]))
    assert rec["operands"] == "fp16" and rec["fp16_saturations"] == 0
    assert full["max"] <= COS_TOL / 10 and mrl["max"] <= COS_TOL / 10, (full, mrl)
