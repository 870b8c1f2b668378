"""GPU end-to-end through the reference-shaped boundary (B1 searcher -> B2 model -> B3 operators -> C ABI):
HybridSearch.search / FlatIPFaissSearch.search / encode_corpus / encode_queries / EmbeddingBag construction, checked
against the oracle.  Includes BASELINE configs[0] (1k docs seq<=128 + 100 queries, the reference's plumbing case) at the
Llama-3.2-1B dims."""
import os
from dataclasses import asdict

import numpy as np
import pytest
import torch

from oracle import lrx_oracle as O
from helpers import GOLDEN, load_model_golden, min_cos

pytestmark = pytest.mark.gpu

PROMPT = "Instruct: Given a web search query, retrieve relevant passages that answer the query\nQuery: "


def tokenizer():
    from transformers import PreTrainedTokenizerFast
    return PreTrainedTokenizerFast.from_pretrained(os.path.join(GOLDEN, "tok"))


def build_stack(cfg_o, w, shrink=None):
    from lightretriever_amd import EncoderConfig, LrxEncoder
    from lightretriever_amd.modeling import LrxExactSearchModel, LrxHybridModel
    tok = tokenizer()
    enc = LrxEncoder(EncoderConfig(**asdict(cfg_o)), {k: torch.from_numpy(v) for k, v in w.items()})
    hm = LrxHybridModel(enc, normalize=True, dense_shrink_dim=shrink, pad_token_id=tok.pad_token_id)
    model = LrxExactSearchModel(model=hm, tokenizer=tok, q_max_len=32, p_max_len=64, eval_batch_size_embedding_bag=100)
    model.query_prompt = PROMPT
    return tok, enc, hm, model


WORDS = "the quick brown fox jumps over lazy dog dense retrieval with large language models amd instinct memory search query capital france paris".split()


def synth_corpus(rng, n, lo=3, hi=60):
    corpus = {}
    for i in range(n):
        text = " ".join(rng.choice(WORDS, size=rng.integers(lo, hi)))
        corpus[f"d{i}"] = {"title": "t%d" % i if i % 3 == 0 else "", "text": text}
    return corpus


def oracle_doc_embeddings(cfg_o, w, tok, docs, p_max_len, shrink=None):
    from lightretriever_amd.modeling import format_text
    enc = tok([format_text(d, prepend_prompt=True) for d in docs], max_length=p_max_len, truncation="only_first", add_special_tokens=True)["input_ids"]
    ids = np.concatenate([np.asarray(e, np.int32) for e in enc])
    cu = np.concatenate([[0], np.cumsum([len(e) for e in enc])]).astype(np.int32)
    return O.encode_passage(cfg_o, w, ids, cu, dense_shrink_dim=shrink)


def test_embedding_bag_construction_matches_reference_table():
    cfg_o, w, _, _, _, _ = load_model_golden("llama_small_d64")
    tok, enc, hm, model = build_stack(cfg_o, w)
    g = np.load(os.path.join(GOLDEN, "embbag.npz"))
    table = hm.construct_embedding_bag(tok, prompt=PROMPT, batch_size=97).cpu().numpy()
    assert table.shape == g["table"].shape
    want = O.construct_embedding_bag(cfg_o, w, int(g["bos"]), int(g["eos"]), [int(t) for t in g["prompt_ids"]], vocab_len=table.shape[0])
    assert min_cos(table, want) > 0.998                       # vs fp32 oracle (bf16 pipeline, un-normalised rows)
    assert min_cos(table, g["table"]) > 0.998                 # vs the reference's own (autocast) table
    assert abs(np.linalg.norm(table) / np.linalg.norm(want) - 1) < 1e-2
    # query operator on top of it == reference emb_reps within the same band
    q = hm.encode_query({"nonctx_tok_emb_input_ids": torch.from_numpy(g["q_ids"]), "nonctx_tok_emb_offsets": torch.from_numpy(g["q_offsets"])})
    assert min_cos(q["emb_reps"].cpu().numpy(), g["emb_reps"]) > 0.999


@pytest.mark.parametrize("prompt", [PROMPT, None])
def test_embedding_bag_shared_prefix_equals_full_sequences(prompt, tmp_path):
    """N1: encoding the shared [bos]+prompt prefix once gives the table the reference's full-sequence loop gives."""
    cfg_o, w, _, _, _, _ = load_model_golden("llama_small_d64")
    tok, enc, hm, model = build_stack(cfg_o, w)
    full = hm.construct_embedding_bag(tok, prompt=prompt, batch_size=97, shared_prefix=False).cpu().numpy()
    fast = hm.construct_embedding_bag(tok, prompt=prompt, batch_size=31, shared_prefix=True).cpu().numpy()
    assert fast.shape == full.shape
    assert min_cos(fast, full) > 0.9995                       # both bf16 pipelines of the same math
    g = np.load(os.path.join(GOLDEN, "embbag.npz"))
    prefix_ids = [int(t) for t in g["prompt_ids"]] if prompt else []
    want = O.construct_embedding_bag(cfg_o, w, int(g["bos"]), int(g["eos"]), prefix_ids, vocab_len=full.shape[0])
    assert min_cos(fast, want) > 0.998                        # vs the fp32 oracle: same band as the full path
    assert abs(np.linalg.norm(fast) / np.linalg.norm(want) - 1) < 1e-2
    # slice build == rows of the whole build; persistence round trip (the *.emb_bag.pt artefact)
    part = hm.construct_embedding_bag(tok, prompt=prompt, batch_size=31, vocab_range=(40, 175)).cpu().numpy()
    np.testing.assert_array_equal(part, fast[40:175])
    path = str(tmp_path / "t.emb_bag.pt")
    hm.save_embedding_bag(path)
    saved = torch.load(path, weights_only=True)
    assert saved.dtype == torch.float32 and saved.device.type == "cpu"
    hm.emb_bag = None
    hm.load_embedding_bag(path, prompt)
    np.testing.assert_array_equal(hm.emb_bag.cpu().numpy(), fast)
    with pytest.raises(ValueError):
        hm.load_embedding_bag(torch.zeros(10, 7))


def test_encode_prefixed_argument_errors():
    from lightretriever_amd._lib import LrxError
    cfg_o, w, _, _, _, _ = load_model_golden("llama_small_d64")
    tok, enc, hm, model = build_stack(cfg_o, w)
    pre = torch.zeros(enc.cfg.max_positions, dtype=torch.int32, device="cuda")
    suf = torch.zeros(4, 2, dtype=torch.int32, device="cuda")
    with pytest.raises(LrxError):
        enc.encode_prefixed(pre, suf)                          # prefix + suffix beyond the RoPE table
    with pytest.raises(TypeError):
        enc.encode_prefixed(pre[:3].long(), suf)


@pytest.mark.parametrize("shrink", [None, 64])
def test_search_equals_oracle_pipeline_on_same_embeddings(shrink):
    cfg_o, w, _, _, _, _ = load_model_golden("llama_small_d64")
    tok, enc, hm, model = build_stack(cfg_o, w, shrink)
    from lightretriever_amd.retriever import HybridSearch, FlatIPFaissSearch
    rng = np.random.default_rng(0)
    corpus = synth_corpus(rng, 90)
    queries = {"q0": "capital of france", "d5": "dense retrieval models", "q2": "a", "d17": corpus["d17"]["text"][:40]}
    searcher = HybridSearch(model, batch_size=16, corpus_chunk_size=40)
    res = searcher.search(corpus, queries, top_k=10, ignore_identical_ids=True)
    # embeddings the product produced, then the reference's search algorithm restated by the oracle
    cids = O.sort_corpus_ids_longest_first(corpus)
    docs = [corpus[c] for c in cids]
    emb = model.encode_corpus(docs, batch_size=16)["dense_reps"].cpu().numpy()
    qe = model.encode_queries(list(queries.values()), batch_size=8)["emb_reps"].cpu().numpy()
    want = O.search_chunks(qe, list(queries), emb, cids, top_k=10, corpus_chunk_size=40, ignore_identical_ids=True)
    assert set(res) == set(want)
    for qid in want:
        assert qid not in res[qid]
        ws, gs = sorted(want[qid].values(), reverse=True), sorted(res[qid].values(), reverse=True)
        np.testing.assert_allclose(gs, ws, atol=3e-6)
        assert len(set(res[qid]) ^ set(want[qid])) <= 2        # only fp32 near-ties may differ
    # document embeddings themselves vs the oracle (tokenisation + encode): inside the bf16 band of this small model
    ref = oracle_doc_embeddings(cfg_o, w, tok, docs, 64, shrink)
    assert 1 - min_cos(emb, ref) < 6e-3
    # the plain dense searcher class gives the same answer for the same query vectors
    class Pre:  # a model that returns precomputed query embeddings (duck-typed B2)
        def __init__(s, m): s.m, s.model = m, m.model
        def encode_queries(s, *a, **k): return torch.from_numpy(qe)
        def encode_corpus(s, *a, **k): return s.m.encode_corpus(*a, **k)
    res2 = FlatIPFaissSearch(Pre(model), batch_size=16, corpus_chunk_size=40).search(corpus, queries, top_k=10, ignore_identical_ids=True)
    for qid in want:
        np.testing.assert_allclose(sorted(res2[qid].values(), reverse=True), sorted(want[qid].values(), reverse=True), atol=3e-6)


def test_index_and_retrieve_with_emb_surface():
    from lightretriever_amd.retriever import HybridSearch
    rng = np.random.default_rng(1)
    X = O.l2_normalize(rng.standard_normal((50, 64)).astype(np.float32))
    q = O.l2_normalize(rng.standard_normal((3, 64)).astype(np.float32))
    hs = HybridSearch(model=None, batch_size=8)
    ids = [f"p{i}" for i in range(50)]
    hs.index({"dense_reps": torch.from_numpy(X)}, ids)
    out = hs.retrieve_with_emb({"emb_reps": q, "dense_reps": torch.from_numpy(q)}, ["a", "b", "c"], top_k=5)
    D, I = O.flat_ip_topk(q, X, 5)
    for name in ("den", "emb"):
        for qi, qid in enumerate(["a", "b", "c"]):
            assert list(out[name][qid]) == [ids[r] for r in I[qi]]
            np.testing.assert_allclose(list(out[name][qid].values()), D[qi], atol=2e-6)
    hs._clear()
    assert hs.dense_search.faiss_index is None
    # top_k larger than the index: no KeyError (reference bug faiss_search.py:168), just fewer hits
    hs.index({"dense_reps": torch.from_numpy(X[:4])}, ids[:4])
    assert len(hs.retrieve_with_emb({"emb_reps": q}, ["a", "b", "c"], top_k=9)["emb"]["a"]) == 4


def test_baseline_config0_plumbing_1k_docs_100_queries_llama1b_dims():
    """BASELINE configs[0]: 1k synthetic docs (seq_len <= 128) + 100 queries end to end through search()."""
    from lightretriever_amd import EncoderConfig, LrxEncoder
    from lightretriever_amd.modeling import LrxExactSearchModel, LrxHybridModel
    from lightretriever_amd.retriever import HybridSearch
    tok = tokenizer()
    enc = LrxEncoder.random_init(EncoderConfig.llama32_1b(), seed=0)
    hm = LrxHybridModel(enc, pad_token_id=tok.pad_token_id)
    model = LrxExactSearchModel(model=hm, tokenizer=tok, q_max_len=32, p_max_len=128)
    model.query_prompt = PROMPT
    rng = np.random.default_rng(2)
    corpus = synth_corpus(rng, 1000, 5, 120)
    queries = {f"q{i}": " ".join(rng.choice(WORDS, size=rng.integers(2, 9))) for i in range(100)}
    res = HybridSearch(model, batch_size=256, corpus_chunk_size=400).search(corpus, queries, top_k=100)
    cids = O.sort_corpus_ids_longest_first(corpus)
    emb = model.encode_corpus([corpus[c] for c in cids], batch_size=256)["dense_reps"]
    qe = model.encode_queries(list(queries.values()), batch_size=100)["emb_reps"]
    assert torch.allclose(emb.norm(dim=1), torch.ones(1000, device="cuda"), atol=1e-5)
    want = O.search_chunks(qe.cpu().numpy(), list(queries), emb.cpu().numpy(), cids, top_k=100, corpus_chunk_size=400)
    agree = []
    for qid in want:
        assert len(res[qid]) == 100
        np.testing.assert_allclose(sorted(res[qid].values(), reverse=True), sorted(want[qid].values(), reverse=True), atol=3e-6)
        agree.append(len(set(res[qid]) & set(want[qid])) / 100)
    assert min(agree) >= 0.98        # identical top-100 recall up to fp32 near-ties


def test_checkpoint_dir_to_search_via_reference_entry_point(tmp_path):
    """eval/eval_utils.py:179 path: InferenceArguments -> PytorchRPCExactSearchModel(args) from an HF checkpoint directory
    (safetensors + tokenizer files) -> HybridSearch.search; embeddings checked against the oracle on the same weights."""
    import shutil
    from transformers import LlamaConfig, LlamaForCausalLM
    from lightretriever.inference.arguments import InferenceArguments
    from lightretriever.inference.exact_search_torchrpc import PytorchRPCExactSearchModel
    from lightretriever.retriever.hybrid_search import HybridSearch
    tok = tokenizer()
    torch.manual_seed(3)
    hf_cfg = LlamaConfig(vocab_size=len(tok), hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2,
                         num_key_value_heads=1, head_dim=64, rms_norm_eps=1e-5, tie_word_embeddings=True,
                         rope_parameters={"rope_type": "llama3", "rope_theta": 5e5, "factor": 8.0, "low_freq_factor": 1.0,
                                          "high_freq_factor": 4.0, "original_max_position_embeddings": 64})
    m = LlamaForCausalLM(hf_cfg).to(torch.bfloat16)
    ckpt = str(tmp_path / "tiny-llama-ckpt")
    m.save_pretrained(ckpt, safe_serialization=True)
    for f in os.listdir(os.path.join(GOLDEN, "tok")):
        shutil.copy(os.path.join(GOLDEN, "tok", f), ckpt)
    args = InferenceArguments(model_name_or_path=ckpt, q_max_len=16, p_max_len=48, batch_size=8, eval_batch_size_embedding_bag=128, model_type="HybridModel",
                              hybrid_use_emb_vector=True, noncontextual_query_embedding=True, bf16=True)
    model = PytorchRPCExactSearchModel(args)
    model.query_prompt = "query: "
    rng = np.random.default_rng(5)
    corpus = synth_corpus(rng, 30, 3, 40)
    queries = {"q0": "capital of france", "q1": "dense retrieval"}
    res = HybridSearch(model, batch_size=8, corpus_chunk_size=16).search(corpus, queries, top_k=5)
    assert set(res) == {"q0", "q1"} and all(len(v) == 5 for v in res.values())
    # oracle on the same checkpoint weights + same tokenisation
    from dataclasses import asdict
    cfg_o = O.EncoderConfig(**{k: v for k, v in asdict(model.model.encoder.cfg).items() if k not in ("fold_norm", "precise_stream", "operand_dtype")})
    w = {k: v.float().numpy() for k, v in m.model.state_dict().items()}
    cids = O.sort_corpus_ids_longest_first(corpus)
    docs = [corpus[c] for c in cids]
    emb = model.encode_corpus(docs, batch_size=8)["dense_reps"].cpu().numpy()
    ref = oracle_doc_embeddings(cfg_o, w, model.tokenizer, docs, 48)
    assert 1 - min_cos(emb, ref) < 5e-3
    table = O.construct_embedding_bag(cfg_o, w, model.tokenizer.bos_token_id, model.tokenizer.eos_token_id,
                                      model.tokenizer.encode("query: ", add_special_tokens=False), vocab_len=len(model.tokenizer))
    assert min_cos(model.model.emb_bag.cpu().numpy(), table) > 0.995


def test_index_save_load_round_trip(tmp_path):
    """N4: FlatIPFaissSearch.save -> {prefix}.flat.faiss + .tsv -> load in a fresh searcher: same hits, no re-encoding."""
    from lightretriever_amd.retriever import FlatIPFaissSearch
    from lightretriever_amd.index_io import read_flat_ip
    rng = np.random.default_rng(4)
    N, D = 1234, 96
    X = O.l2_normalize(rng.standard_normal((N, D)).astype(np.float32))
    q = O.l2_normalize(rng.standard_normal((7, D)).astype(np.float32))
    cids = ["doc/%d" % i for i in range(N)]
    a = FlatIPFaissSearch(model=None, batch_size=8)
    a.index(torch.from_numpy(X), cids)
    want = a.retrieve_with_emb(torch.from_numpy(q).cuda(), ["q%d" % i for i in range(7)], top_k=25)
    a.save(str(tmp_path), prefix="my-index")
    assert sorted(os.listdir(tmp_path)) == ["my-index.flat.faiss", "my-index.flat.tsv"]
    np.testing.assert_array_equal(np.asarray(read_flat_ip(str(tmp_path / "my-index.flat.faiss"))), X)
    b = FlatIPFaissSearch(model=None, batch_size=8)
    b.load(str(tmp_path), prefix="my-index")
    assert b.dim_size == D and b.faiss_index.index.ntotal == N and b.mapping == a.mapping
    got = b.retrieve_with_emb(torch.from_numpy(q).cuda(), ["q%d" % i for i in range(7)], top_k=25)
    assert got == want
    Dw, Iw = O.flat_ip_topk(q, X, 25)
    assert [list(got["q0"].keys())[j] for j in range(25)] == ["doc/%d" % i for i in Iw[0]]


def test_token_budget_batch_merging_leaves_every_row_unchanged():
    """encode_corpus merges consecutive collated batches up to max_batch_tokens; rows must be bit-identical to one-batch-at-a-time
    encoding (dense and sparse), whatever the budget cuts."""
    cfg_o, w, _, _, _, _ = load_model_golden("llama_small_d64")
    tok, enc, hm, model = build_stack(cfg_o, w)
    docs = list(synth_corpus(np.random.default_rng(3), 157).values())
    model.max_batch_tokens = 0
    want = model.encode_corpus(docs, batch_size=8)["dense_reps"].clone()
    for budget, max_docs in [(131072, 2048), (300, 2048), (64, 2048), (100000, 20)]:
        model.max_batch_tokens, model.max_batch_docs = budget, max_docs
        got = model.encode_corpus(docs, batch_size=8)["dense_reps"]
        assert torch.equal(got, want), (budget, max_docs)
    # merged spans really are larger than one collated batch
    from lightretriever_amd.modeling import EncodeCollator, _prefetch_batches, _token_budget_batches
    coll = EncodeCollator(tok, encode_is_query=False, p_max_len=64)
    spans = [(s, e) for s, e, _ in _token_budget_batches(_prefetch_batches(coll, model.parse_texts(docs), 8), 1000, 2048)]
    assert spans[0][0] == 0 and spans[-1][1] == len(docs) and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    assert max(e - s for s, e in spans) > 8
    # sparse branch: the per-token mask travels with the merge
    from lightretriever_amd.modeling import LrxHybridModel, LrxExactSearchModel
    hs = LrxHybridModel(enc, normalize=True, pad_token_id=tok.pad_token_id, encode_sparse=True)
    ms = LrxExactSearchModel(model=hs, tokenizer=tok, q_max_len=32, p_max_len=64, max_batch_tokens=0)
    a = ms.encode_corpus(docs[:60], batch_size=8)
    ms.max_batch_tokens = 500
    b = ms.encode_corpus(docs[:60], batch_size=8)
    assert torch.equal(a["dense_reps"], b["dense_reps"]) and a["sparse_reps"] == b["sparse_reps"]


def test_cli_arguments_wire_the_sparse_branch_through_the_reference_entry_point(tmp_path):
    """The reference's model flags arrive through HfArgumentParser (eval/eval_arguments.py) and select what encode_* returns:
    --hybrid_use_token_id_vector (+ relu / log saturation) -> corpus: dense rows + quantised sparse JSON from the LM head,
    queries: EmbeddingBag rows + token-id counts; same vectors as a hand-built LrxHybridModel with those options."""
    import shutil
    from transformers import HfArgumentParser, LlamaConfig, LlamaForCausalLM
    from lightretriever.inference.arguments import InferenceArguments
    from lightretriever.inference.exact_search_torchrpc import PytorchRPCExactSearchModel
    from lightretriever_amd.modeling import LrxExactSearchModel, LrxHybridModel
    tok = tokenizer()
    torch.manual_seed(4)
    hf_cfg = LlamaConfig(vocab_size=len(tok), hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2,
                         num_key_value_heads=1, head_dim=64, rms_norm_eps=1e-5, tie_word_embeddings=True,
                         rope_parameters={"rope_type": "default", "rope_theta": 1e4})
    ckpt = str(tmp_path / "ckpt")
    LlamaForCausalLM(hf_cfg).to(torch.bfloat16).save_pretrained(ckpt, safe_serialization=True)
    for f in os.listdir(os.path.join(GOLDEN, "tok")):
        shutil.copy(os.path.join(GOLDEN, "tok", f), ckpt)
    (args,) = HfArgumentParser(InferenceArguments).parse_args_into_dataclasses(
        ["--model_name_or_path", ckpt, "--q_max_len", "16", "--p_max_len", "48", "--batch_size", "8", "--eval_batch_size_embedding_bag", "128",
         "--model_type", "HybridModel", "--hybrid_use_emb_vector", "--noncontextual_query_embedding", "--hybrid_use_token_id_vector", "--hybrid_use_sparse_vector", "--sparse_use_relu", "--sparse_use_log_saturation",
         "--sparse_top_k_psg", "12", "--token_id_vector_type", "bow", "--anserini_vector_type", "JsonVectorCollection"])
    assert args.encode_sparse and args.normalize
    model = PytorchRPCExactSearchModel(args)
    assert model.encoding_kwargs["anserini_vector_type"] == "JsonVectorCollection"
    docs = list(synth_corpus(np.random.default_rng(9), 21, 3, 40).values())
    got = model.encode_corpus(docs, batch_size=8)
    assert set(got) == {"dense_reps", "sparse_reps"} and len(got["sparse_reps"]) == len(docs)
    assert all(isinstance(k, str) and isinstance(v, int) and v > 0 for d in got["sparse_reps"] for k, v in d.items())
    assert max(len(d) for d in got["sparse_reps"]) <= 12 + 8                   # top-k 12 (ties kept) on top of min_tokens_to_keep
    hm = LrxHybridModel(model.model.encoder, normalize=True, pad_token_id=model.tokenizer.pad_token_id, encode_sparse=True,
                        sep_token_id=getattr(model.tokenizer, "sep_token_id", None), sparse_use_relu=True, sparse_use_log_saturation=True,
                        sparse_top_k_psg=12)
    want = LrxExactSearchModel(model=hm, tokenizer=model.tokenizer, q_max_len=16, p_max_len=48).encode_corpus(docs, batch_size=8)
    assert torch.equal(got["dense_reps"], want["dense_reps"]) and got["sparse_reps"] == want["sparse_reps"]
    q = model.encode_queries(["capital of france", "dense retrieval retrieval"], batch_size=8)
    # (round 6: `--hybrid_use_sparse_vector` is served -- the LM-head query vectors ride along as pseudo text, like the reference's call_batch_encode)
    assert set(q) == {"emb_reps", "token_id_reps", "sparse_reps"} and all(v == 1 for d in q["token_id_reps"] for v in d.values())   # 'bow'
    assert len(q["sparse_reps"]) == 2 and all(isinstance(t, str) and t for t in q["sparse_reps"])


def test_embedding_bag_prompt_selection_follows_the_reference():
    """exact_search_torchrpc.py:139-147: the table's prompt is model.query_prompt, else the `prompt` column of the first query, with
    noncontextual_prompt_prefix in front; the table is rebuilt only when that string changes."""
    cfg_o, w, _, _, _, _ = load_model_golden("llama_small_d64")
    tok, enc, hm, model = build_stack(cfg_o, w)
    model.query_prompt = "query: "
    a = model.encode_queries(["capital of france", "dense retrieval"], batch_size=8)["emb_reps"].clone()
    table = model.model.emb_bag
    model.query_prompt = ""                                    # evaluate_mteb.py resets the prompts per task
    b = model.encode_queries([{"text": "capital of france", "prompt": "query: "}, {"text": "dense retrieval", "prompt": "query: "}], batch_size=8)["emb_reps"]
    assert model.model.emb_bag_prompt == "query: " and model.model.emb_bag is table and torch.equal(a, b)      # same string: no rebuild
    c = model.encode_queries(["capital of france", "dense retrieval"], batch_size=8)["emb_reps"]              # no prompt anywhere
    assert model.model.emb_bag_prompt is None and not torch.equal(a, c)
    model.noncontextual_prompt_prefix = "Instruct: "
    model.encode_queries(["capital of france"], batch_size=8)
    assert model.model.emb_bag_prompt == "Instruct: "
    model.query_prompt = "query: "
    model.encode_queries(["capital of france"], batch_size=8)
    assert model.model.emb_bag_prompt == "Instruct: query: "
