"""GPU parity for the sparse document-vector row: LM-head max aggregation in the GEMM epilogue, sparsify,
quantise + compaction and the encode_passage(encode_sparse) operator, against the oracle and the reference's goldens."""
import json
import os
from dataclasses import asdict

import numpy as np
import pytest
import torch

from oracle import lrx_oracle as O
from helpers import GOLDEN, load_model_golden, min_cos

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BF16_ULP = 2.0 ** -7


def bf16_t(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV).to(torch.bfloat16).contiguous()


def close_bf16(got, want, frac_exact=0.98):
    """logits are bf16 values of fp32 sums: a different accumulation order may flip the last bf16 bit now and then."""
    assert got.shape == want.shape
    tol = BF16_ULP * np.abs(want) + 1e-6
    assert (np.abs(got - want) <= tol).all(), float(np.abs(got - want).max())
    assert (got == want).mean() >= frac_exact, float((got == want).mean())


@pytest.mark.parametrize("lens,V,H,masked", [
    ([40, 3, 2, 17, 33, 1, 40, 25], 290, 256, "default"),      # ragged, vocab not a multiple of 8, empty rows (len <= 2)
    ([7] * 90 + [1, 2, 3], 1000, 128, "random"),                # dozens of documents inside one 256-row tile
    ([700, 5, 300], 520, 64, "random"),                          # a document spanning three row tiles
    ([256, 256, 512], 256, 192, None),                           # tile-aligned boundaries, every token counted
])
def test_max_aggregate_kernel_equals_oracle(lens, V, H, masked):
    from lightretriever_amd import ops
    rng = np.random.default_rng(len(lens) + V)
    T = sum(lens)
    hid = O.round_bf16(rng.standard_normal((T, H)).astype(np.float32))
    W = O.round_bf16(rng.standard_normal((V, H)).astype(np.float32) * 0.1)
    cu = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    if masked == "default":
        tm = None                                             # kernel default: drop each sequence's first and last token
        tm_np = np.ones(T, bool)
        tm_np[cu[:-1]] = False
        tm_np[cu[1:] - 1] = False
    elif masked == "random":
        tm_np = rng.random(T) < 0.6
        tm = torch.from_numpy(tm_np.astype(np.uint8)).to(DEV)
    else:
        tm_np = np.ones(T, bool)
        tm = torch.from_numpy(tm_np.astype(np.uint8)).to(DEV)
    got = ops.sparse_max_aggregate(bf16_t(hid), bf16_t(W), torch.from_numpy(cu).to(DEV), tm).cpu().numpy()
    want = O.max_aggregate_packed(hid, cu, tm_np, W, None, bf16=True)
    empty = want == O.BF16_MIN
    np.testing.assert_array_equal(got == O.BF16_MIN, empty)
    close_bf16(got[~empty], want[~empty])


def test_max_aggregate_with_bias_and_negative_maxima():
    from lightretriever_amd import ops
    rng = np.random.default_rng(5)
    lens, V, H = [30, 9, 300], 264, 64
    T = sum(lens)
    hid = O.round_bf16(rng.standard_normal((T, H)).astype(np.float32))
    W = O.round_bf16(rng.standard_normal((V, H)).astype(np.float32) * 0.05)
    bias = O.round_bf16((rng.standard_normal(V) - 6.0).astype(np.float32))          # pushes whole columns below zero
    cu = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    tm_np = rng.random(T) < 0.5
    got = ops.sparse_max_aggregate(bf16_t(hid), bf16_t(W), torch.from_numpy(cu).to(DEV), torch.from_numpy(tm_np.astype(np.uint8)).to(DEV),
                                   bias=bf16_t(bias)).cpu().numpy()
    want = O.max_aggregate_packed(hid, cu, tm_np, W, bias, bf16=True)
    assert (want < 0).mean() > 0.5
    close_bf16(got, want)


@pytest.mark.parametrize("relu,log1p,round_bf16,top_k,min_keep", [
    (True, False, False, 0, 8), (True, False, False, 16, 8), (True, False, False, 3, 8), (False, False, False, 5, 1),
    (True, True, False, 0, 8), (True, True, True, 32, 8), (True, True, True, 100000, 8)])
def test_sparsify_equals_oracle(relu, log1p, round_bf16, top_k, min_keep):
    from lightretriever_amd import ops
    rng = np.random.default_rng(top_k + 7)
    x = O.round_bf16(rng.standard_normal((13, 3001)).astype(np.float32) * 2)          # bf16-valued: plenty of exact ties
    x[3] = O.BF16_MIN                                                                  # a document without valid tokens
    x[5, :] = -1.0                                                                      # all equal
    got = ops.sparsify_(torch.from_numpy(x.copy()).to(DEV), relu, log1p, round_bf16, top_k, min_keep).cpu().numpy()
    want = O.sparsify(x, relu=relu, log1p=log1p, top_k=top_k, min_tokens_to_keep=min_keep, bf16=round_bf16)
    if not log1p:
        np.testing.assert_array_equal(got, want)                                       # selection is exact, ties included
    else:
        np.testing.assert_allclose(got, want, rtol=2.0 ** -7 if round_bf16 else 2e-6, atol=1e-7)
        assert ((got > 0) == (want > 0)).mean() > 0.9999
    if relu:
        assert (got[3] == 0).all()


def test_compact_and_json_match_reference_converter():
    from lightretriever_amd import ops
    g = np.load(os.path.join(GOLDEN, "sparse.npz"))
    with open(os.path.join(GOLDEN, "sparse_json.json")) as f:
        want = json.load(f)
    for key, jkey in (("sparse_reps", "quant100"), ("sparse_reps_top16", "quant100_top16")):
        reps = torch.from_numpy(g[key]).to(DEV)
        ids, w, cnt = ops.sparse_compact(reps, 100)
        ids, w, cnt = ids.cpu().numpy(), w.cpu().numpy(), cnt.cpu().numpy()
        got = [{str(int(i)): int(v) for i, v in zip(ids[b, :cnt[b]], w[b, :cnt[b]])} or {"-1": 1} for b in range(len(cnt))]
        assert got == want[jkey]
        assert all(list(map(int, d)) == sorted(map(int, d)) for d in got)             # ascending token ids
    # round-half-even and capacity truncation
    halves = torch.tensor([[0.5 / 7, 1.5 / 7, 2.5 / 7, -3.0, 0.0, 0.07]], device=DEV)
    ids, w, cnt = ops.sparse_compact(halves, 7, capacity=2)
    assert int(cnt[0]) == len(want["quant7_halves"][0])
    assert [(str(int(i)), int(v)) for i, v in zip(ids[0].cpu(), w[0].cpu())] == list(want["quant7_halves"][0].items())[:2]


def _sparse_model(**kw):
    from lightretriever_amd import EncoderConfig, LrxEncoder
    from lightretriever_amd.modeling import LrxHybridModel
    cfg_o, w, _, _, _, _ = load_model_golden("llama_small_d64")
    enc = LrxEncoder(EncoderConfig(**asdict(cfg_o)), {k: torch.from_numpy(v) for k, v in w.items()})
    g = np.load(os.path.join(GOLDEN, "sparse.npz"))
    hm = LrxHybridModel(enc, normalize=True, encode_sparse=True, sep_token_id=int(g["sep_token_id"]), add_sep_token=True, **kw)
    return cfg_o, w, g, enc, hm


def test_encode_passage_sparse_matches_reference_and_oracle():
    cfg_o, w, g, enc, hm = _sparse_model(sparse_round_bf16=False)
    psg = {"input_ids": torch.from_numpy(g["input_ids"]), "attention_mask": torch.from_numpy(g["attention_mask"])}
    out = hm.encode_passage(psg)
    sp, dn = out["sparse_reps"].cpu().numpy(), out["dense_reps"].cpu().numpy()
    assert sp.shape == g["sparse_reps"].shape
    # the reference's fp32 run and its autocast run differ by 0.019 on this fixture; the bf16 pipeline sits in the same band
    assert np.abs(sp - g["sparse_reps"]).max() < 0.06
    assert np.abs(sp - g["sparse_reps_autocast"]).max() < 0.06
    np.testing.assert_array_equal((g["sparse_reps"] > 0).sum(1) == 0, (sp > 0).sum(1) == 0)      # same empty documents
    assert ((sp > 0) == (g["sparse_reps"] > 0)).mean() > 0.985                                     # support differs only at |logit| ~ 0
    assert min_cos(dn, g["dense_reps"]) > 0.998
    # tight: the oracle's bf16 restatement of the same pipeline
    ids, _, _, cu, _ = O.pack_padded(g["input_ids"], g["attention_mask"])
    am = g["attention_mask"].astype(bool)
    want = O.encode_passage_sparse(cfg_o, w, ids, cu, g["mask_noprompt"][am], bf16=True, relu=True, log1p=True)
    assert np.abs(sp - want).max() < 0.04
    # dense branch of the same call == the dense-only operator
    dn2 = hm.encode_passage(psg, encode_sparse=False)["dense_reps"].cpu().numpy()
    assert min_cos(dn, dn2) > 0.99999
    # quantised JSON: same keys up to entries that quantise to 0/1, weights within 0.06 * 100
    got = hm.convert_sparse_reps_to_json(out["sparse_reps"], 100)
    with open(os.path.join(GOLDEN, "sparse_json.json")) as f:
        ref = json.load(f)["quant100"]
    for a, b in zip(got, ref):
        if b == {"-1": 1}:
            assert a == b
            continue
        for k in set(a) | set(b):
            assert abs(a.get(k, 0) - b.get(k, 0)) <= 6, (k, a.get(k), b.get(k))


@pytest.mark.parametrize("kw,key", [(dict(sparse_top_k_psg=16), "sparse_reps_top16"), (dict(sparse_top_k_psg=3), "sparse_reps_top3_min8"),
                                    (dict(sparse_top_p_psg=0.3), "sparse_reps_topp")])
def test_encode_passage_sparse_sampling_options(kw, key):
    cfg_o, w, g, enc, hm = _sparse_model(sparse_round_bf16=False, **kw)
    psg = {"input_ids": torch.from_numpy(g["input_ids"]), "attention_mask": torch.from_numpy(g["attention_mask"])}
    sp = hm.encode_passage(psg)["sparse_reps"].cpu().numpy()
    want = g[key]
    np.testing.assert_array_equal((sp > 0).sum(1) == 0, (want > 0).sum(1) == 0)
    nz_got, nz_want = (sp > 0).sum(1), (want > 0).sum(1)
    if "sparse_top_k_psg" in kw:
        assert (nz_got[nz_want > 0] >= nz_want[nz_want > 0]).all() and (nz_got <= nz_want + 2).all()    # k survivors (+ bf16 ties)
    # the kept entries are the same tokens except where two logits are within bf16 noise of the threshold
    overlap = ((sp > 0) & (want > 0)).sum() / max(1, (want > 0).sum())
    assert overlap > 0.9
    both = (sp > 0) & (want > 0)
    assert np.abs(sp[both] - want[both]).max() < 0.06


def test_encode_query_sparse_matches_the_reference():
    """Round 6: `--hybrid_use_sparse_vector` -- LM-encoded sparse QUERY vectors (finetune/modeling_hybrid.py:404-438; the `spr` / `den_spr` query modes
    of retriever/hybrid_search.py:160-180) next to the dense vector of the same forward, against what the reference's encode_query returned
    (tests/golden/sparse_query.npz, gen_sparse_query_goldens.py); the *_qry sampling ratios apply, not the passage ones; the pseudo text of a
    query vector is the reference converter's up to the bf16 band of the logits."""
    from lightretriever_amd.modeling import LrxHybridModel
    cfg_o, w, sp, enc, _ = _sparse_model()
    g = np.load(os.path.join(GOLDEN, "sparse_query.npz"))
    mk = lambda **kw: LrxHybridModel(enc, normalize=True, encode_sparse=True, hybrid_use_sparse_vector=True, hybrid_use_dense_vector=True,
                                     hybrid_use_emb_vector=False, sep_token_id=int(g["sep_token_id"]), add_sep_token=True, sparse_round_bf16=False, **kw)
    qry = {"input_ids": torch.from_numpy(g["input_ids"]), "attention_mask": torch.from_numpy(g["attention_mask"])}
    out = mk().encode_query(qry)
    assert set(out) == {"dense_reps", "sparse_reps"}
    spq, dn = out["sparse_reps"].cpu().numpy(), out["dense_reps"].cpu().numpy()
    assert spq.shape == g["sparse_reps"].shape and np.abs(spq - g["sparse_reps"]).max() < 0.06
    np.testing.assert_array_equal((g["sparse_reps"] > 0).sum(1) == 0, (spq > 0).sum(1) == 0)
    assert ((spq > 0) == (g["sparse_reps"] > 0)).mean() > 0.985 and min_cos(dn, g["dense_reps"]) > 0.998
    only = mk().encode_query(qry, encode_dense=False)
    assert set(only) == {"sparse_reps"} and torch.equal(only["sparse_reps"], out["sparse_reps"])
    # the query-side ratios (the passage ones, set differently, must not be the ones applied)
    for kw, key in ((dict(sparse_top_k_qry=8, sparse_min_tokens_to_keep=4, sparse_top_k_psg=16), "sparse_reps_top8_qry"),
                    (dict(sparse_top_p_qry=0.4, sparse_min_tokens_to_keep=8, sparse_top_p_psg=0.9), "sparse_reps_topp_qry")):
        got, want = mk(**kw).encode_query(qry)["sparse_reps"].cpu().numpy(), g[key]
        np.testing.assert_array_equal((got > 0).sum(1) == 0, (want > 0).sum(1) == 0)
        nz_got, nz_want = (got > 0).sum(1), (want > 0).sum(1)
        if "sparse_top_k_qry" in kw:
            assert (nz_got[nz_want > 0] >= nz_want[nz_want > 0]).all() and (nz_got <= nz_want + 2).all()
        both = (got > 0) & (want > 0)
        assert both.sum() / max(1, (want > 0).sum()) > 0.9 and np.abs(got[both] - want[both]).max() < 0.06
    # pseudo text (what call_batch_encode hands the sparse engine for a query): token ids repeated by their quantised weight
    txt = mk().convert_sparse_reps_to_pseudo_text(out["sparse_reps"], 100)
    ref = json.load(open(os.path.join(GOLDEN, "sparse_query_text.json")))["quant100_row1"][0]
    from collections import Counter
    a, b = Counter(txt[1].split()), Counter(ref.split())
    assert txt[2] == "-1" and all(abs(a.get(k, 0) - b.get(k, 0)) <= 6 for k in set(a) | set(b))
    assert mk().convert_sparse_reps_to_pseudo_text(torch.tensor([[0.5 / 7, 1.5 / 7, 2.5 / 7, -3.0, 0.0, 0.07], [0.0] * 6]), 7) == json.load(open(os.path.join(GOLDEN, "sparse_query_text.json")))["quant7_halves"]


def test_sparse_vectors_pooled_from_the_input_ids_match_the_reference():
    """`--sparse_pool_from_original_input_ids_psg / _qry` (finetune/modeling_hybrid.py:175-180): only the sequence's own tokens (sparse attention
    mask) keep their aggregated logit -- against what the reference's encode_passage / encode_query returned with the flags on
    (tests/golden/sparse_pool_ids.npz, gen_sparse_pool_ids_goldens.py); each flag is its own side's."""
    from lightretriever_amd.modeling import LrxHybridModel
    cfg_o, w, sp, enc, _ = _sparse_model()
    g = np.load(os.path.join(GOLDEN, "sparse_pool_ids.npz"))
    mk = lambda **kw: LrxHybridModel(enc, normalize=True, encode_sparse=True, hybrid_use_sparse_vector=True, hybrid_use_dense_vector=True,
                                     hybrid_use_emb_vector=False, sep_token_id=int(g["sep_token_id"]), add_sep_token=True, sparse_round_bf16=False, **kw)
    batch = {"input_ids": torch.from_numpy(g["input_ids"]), "attention_mask": torch.from_numpy(g["attention_mask"])}
    plain = mk().encode_passage(batch)
    for side, flag in (("psg", "sparse_pool_from_original_input_ids_psg"), ("qry", "sparse_pool_from_original_input_ids_qry")):
        hm = mk(**{flag: True})
        call = hm.encode_passage if side == "psg" else hm.encode_query
        other = hm.encode_query if side == "psg" else hm.encode_passage
        out = call(batch)
        got, want = out["sparse_reps"].cpu().numpy(), g[side]
        np.testing.assert_array_equal(got > 0, want > 0)                              # the support is the set of own tokens with a positive logit
        assert np.abs(got - want).max() < 0.06                                        # (the bf16 band of the logits, as for the full vector)
        assert torch.equal(out["dense_reps"], plain["dense_reps"])                    # the dense vector of the same pass is untouched
        assert torch.equal(other(batch)["sparse_reps"], plain["sparse_reps"])         # the other side's vectors keep their expansion terms
        kw = {flag: True, "sparse_top_k_psg" if side == "psg" else "sparse_top_k_qry": 4, "sparse_min_tokens_to_keep": 2}
        hk = mk(**kw)
        top = (hk.encode_passage if side == "psg" else hk.encode_query)(batch)["sparse_reps"].cpu().numpy()
        want4 = g[side + "_top4"]
        nz_got, nz_want = (top > 0).sum(1), (want4 > 0).sum(1)
        assert (nz_got >= nz_want).all() and (nz_got <= nz_want + 2).all()            # k survivors (+ ties inside the bf16 band)
        both = (top > 0) & (want4 > 0)
        assert both.sum() / max(1, (want4 > 0).sum()) > 0.8 and np.abs(top[both] - want4[both]).max() < 0.06


def test_encode_queries_returns_sparse_pseudo_text_for_the_spr_mode():
    """B2 level: `--hybrid_use_sparse_vector` without token-id queries -> encode_queries returns one pseudo-text string per query
    (inference/exact_search_base.py:231-236) and no token_id_reps; with both flags, both."""
    from transformers import PreTrainedTokenizerFast
    from lightretriever_amd.modeling import LrxExactSearchModel, LrxHybridModel
    cfg_o, w, sp, enc, _ = _sparse_model()
    tok = PreTrainedTokenizerFast.from_pretrained(os.path.join(GOLDEN, "tok"))
    mk = lambda **kw: LrxExactSearchModel(model=LrxHybridModel(enc, normalize=True, encode_sparse=True, hybrid_use_sparse_vector=True, hybrid_use_emb_vector=False,
                                                               pad_token_id=tok.pad_token_id, **kw), tokenizer=tok, q_max_len=32, p_max_len=48)
    qs = ["dense retrieval with large language models", "a", "memory search"]
    r = mk().encode_queries(qs, batch_size=2)
    assert set(r) == {"sparse_reps"} and len(r["sparse_reps"]) == 3 and all(isinstance(t, str) for t in r["sparse_reps"])
    toks = r["sparse_reps"][0].split()
    assert len(toks) > 10 and all(t.isdigit() and int(t) < cfg_o.vocab_size for t in toks)
    both = mk(hybrid_use_token_id_vector=True, hybrid_use_dense_vector=True).encode_queries(qs, batch_size=3)
    assert set(both) == {"sparse_reps", "token_id_reps", "dense_reps"} and both["sparse_reps"] == r["sparse_reps"] and both["dense_reps"].shape == (3, cfg_o.hidden_size)
    # each query separately (batching must not matter) against the oracle's pipeline
    from lightretriever_amd.modeling import format_text
    e = tok([format_text({"text": qs[0]}, prepend_prompt=True)], max_length=32, truncation="only_first", add_special_tokens=True)["input_ids"][0]
    tm = np.ones(len(e), bool)
    tm[0] = tm[-1] = False
    want = O.sparse_reps_to_json(O.encode_query_sparse(cfg_o, w, np.asarray(e, np.int32), np.array([0, len(e)], np.int32), tm, bf16=True, relu=True, log1p=True), 100)[0]
    from collections import Counter
    got = Counter(toks)
    assert all(abs(got.get(k, 0) - want.get(k, 0)) <= 5 for k in set(got) | set(want))


def test_packed_input_with_collator_mask_equals_padded_input():
    cfg_o, w, g, enc, hm = _sparse_model()
    from lightretriever_amd.modeling import sparse_token_mask
    ids, _, _, cu, max_len = O.pack_padded(g["input_ids"], g["attention_mask"])
    psg_packed = {"input_ids": torch.from_numpy(ids.astype(np.int32)), "cu_seqlens": torch.from_numpy(cu), "max_seqlen": max_len,
                  "sparse_mask": torch.from_numpy(sparse_token_mask(ids, cu, int(g["sep_token_id"]), True))}
    a = hm.encode_passage(psg_packed)["sparse_reps"]
    b = hm.encode_passage({"input_ids": torch.from_numpy(g["input_ids"]), "attention_mask": torch.from_numpy(g["attention_mask"])})["sparse_reps"]
    assert torch.equal(a, b)


def test_max_aggregate_full_vocab_against_device_matmul():
    """Llama-3.2-1B head dims (V = 128256, H = 2048): sampled vocabulary columns against a plain fp32 matmul on the device."""
    from lightretriever_amd import ops
    gen = torch.Generator(device=DEV).manual_seed(0)
    V, H, lens = 128256, 2048, [512, 100, 512, 1, 37, 300]
    T = sum(lens)
    hid = (torch.randn(T, H, generator=gen, device=DEV)).to(torch.bfloat16)
    W = (torch.randn(V, H, generator=gen, device=DEV) * 0.02).to(torch.bfloat16)
    cu = torch.tensor(np.concatenate([[0], np.cumsum(lens)]), dtype=torch.int32, device=DEV)
    got = ops.sparse_max_aggregate(hid, W, cu, None)
    cols = torch.cat([torch.arange(0, 300, device=DEV), torch.randint(0, V, (500,), generator=gen, device=DEV), torch.arange(V - 300, V, device=DEV)])
    logits = (hid.float() @ W[cols].float().T).to(torch.bfloat16).float()
    want = torch.full((len(lens), cols.numel()), float(O.BF16_MIN), device=DEV)
    for b, n in enumerate(lens):
        s = int(cu[b])
        if n > 2:
            want[b] = logits[s + 1:s + n - 1].max(0).values
    g_, w_ = got[:, cols].cpu().numpy(), want.cpu().numpy()
    empty = w_ == O.BF16_MIN
    np.testing.assert_array_equal(g_ == O.BF16_MIN, empty)
    close_bf16(g_[~empty], w_[~empty], frac_exact=0.97)
    assert torch.isfinite(got).all()


def test_encode_corpus_and_queries_carry_sparse_vectors():
    """B2 level: encode_corpus -> dense rows + one quantised JSON vector per document; encode_queries -> token-count vectors;
    the notebook's similarity (scripts/asymmetric_sparse_infer.ipynb: sum over shared token ids) on them equals the oracle's."""
    from transformers import PreTrainedTokenizerFast
    from lightretriever_amd.modeling import LrxExactSearchModel, format_text
    cfg_o, w, g, enc, hm = _sparse_model()
    tok = PreTrainedTokenizerFast.from_pretrained(os.path.join(GOLDEN, "tok"))
    hm.pad_token_id, hm.sep_token_id, hm.add_sep_token = tok.pad_token_id, None, False
    model = LrxExactSearchModel(model=hm, tokenizer=tok, q_max_len=32, p_max_len=48, eval_batch_size_embedding_bag=100)
    corpus = [{"title": "", "text": "dense retrieval with large language models"}, {"title": "amd", "text": "instinct memory search"},
              {"title": "", "text": "a"}, {"title": "", "text": "the quick brown fox jumps over the lazy dog " * 3}]
    res = model.encode_corpus(corpus, batch_size=3)
    assert res["dense_reps"].shape == (4, cfg_o.hidden_size) and len(res["sparse_reps"]) == 4
    encd = tok([format_text(d, prepend_prompt=True) for d in corpus], max_length=48, truncation="only_first", add_special_tokens=True)["input_ids"]
    for b, e in enumerate(encd):                       # each document separately through the oracle (batching must not matter)
        ids, cu = np.asarray(e, np.int32), np.array([0, len(e)], np.int32)
        tm = np.ones(len(e), bool)
        tm[0] = tm[-1] = False
        want = O.sparse_reps_to_json(O.encode_passage_sparse(cfg_o, w, ids, cu, tm, bf16=True, relu=True, log1p=True), 100)[0]
        got = res["sparse_reps"][b]
        if want == {"-1": 1}:
            assert got == want
            continue
        for k in set(got) | set(want):
            assert abs(got.get(k, 0) - want.get(k, 0)) <= 5, (b, k, got.get(k), want.get(k))
    q = model.encode_queries(["memory search search", "fox"], batch_size=2)
    assert "emb_reps" in q and len(q["token_id_reps"]) == 2
    from collections import Counter
    want_q = Counter(tok(" memory search search", add_special_tokens=False)["input_ids"])       # 'sum': token -> count, leading blank
    assert q["token_id_reps"][0] == {str(k): v for k, v in want_q.items()} and max(want_q.values()) >= 2
    model.token_id_vector_type = "bow"
    assert set(model.encode_queries(["memory search search"], batch_size=1)["token_id_reps"][0].values()) == {1}
    score = lambda qr, pr: sum(v * pr[k] for k, v in qr.items() if k in pr)
    assert score(q["token_id_reps"][0], res["sparse_reps"][1]) >= 0


def test_untied_lm_head_matches_reference_golden(tmp_path):
    """a checkpoint with tie_word_embeddings=false must project the sparse branch with its OWN head, through the
    loader (safetensors dir -> load_hf_checkpoint -> LrxEncoder.lm_head), not with embed_tokens.  Golden = the reference's
    HybridModel.encode_passage on the same untied tiny model (tests/golden/gen_sparse_untied_golden.py)."""
    from safetensors.torch import save_file
    from lightretriever_amd.loader import encoder_from_pretrained
    from lightretriever_amd.modeling import LrxHybridModel
    cfg_o, w, _, _, _, _ = load_model_golden("llama_small_d64")
    g = np.load(os.path.join(GOLDEN, "sparse_untied.npz"))
    head = np.random.default_rng(int(g["head_seed"])).standard_normal((cfg_o.vocab_size, cfg_o.hidden_size)).astype(np.float32) * np.float32(0.05)
    head = torch.from_numpy(head).to(torch.bfloat16)
    d = tmp_path / "untied-llama"
    d.mkdir()
    sd = {"model." + k: torch.from_numpy(v).to(torch.bfloat16).contiguous() for k, v in w.items()}
    sd["lm_head.weight"] = head.contiguous()
    save_file(sd, str(d / "model.safetensors"))
    json.dump({"model_type": "llama", "vocab_size": cfg_o.vocab_size, "hidden_size": cfg_o.hidden_size, "intermediate_size": cfg_o.intermediate_size,
               "num_hidden_layers": cfg_o.num_layers, "num_attention_heads": cfg_o.num_q_heads, "num_key_value_heads": cfg_o.num_kv_heads,
               "head_dim": cfg_o.head_dim, "rms_norm_eps": cfg_o.rms_eps, "tie_word_embeddings": False,
               "rope_parameters": {"rope_type": "llama3", "rope_theta": cfg_o.rope_theta, "factor": cfg_o.rope_factor, "low_freq_factor": cfg_o.rope_low_freq_factor,
                                   "high_freq_factor": cfg_o.rope_high_freq_factor, "original_max_position_embeddings": cfg_o.rope_original_max_position}},
              open(d / "config.json", "w"))
    enc = encoder_from_pretrained(str(d), max_positions=512)
    assert enc.lm_head is not None and torch.equal(enc.lm_head.cpu(), head)
    hm = LrxHybridModel(enc, normalize=True, encode_sparse=True, sep_token_id=int(g["sep_token_id"]), add_sep_token=True, sparse_round_bf16=False,
                        sparse_use_relu=True, sparse_use_log_saturation=True)
    out = hm.encode_passage({"input_ids": torch.from_numpy(g["input_ids"]), "attention_mask": torch.from_numpy(g["attention_mask"])})
    sp = out["sparse_reps"].cpu().numpy()
    assert np.abs(sp - g["sparse_reps"]).max() < 0.06 and np.abs(sp - g["sparse_reps_autocast"]).max() < 0.06
    np.testing.assert_array_equal((g["sparse_reps"] > 0).sum(1) == 0, (sp > 0).sum(1) == 0)
    assert ((sp > 0) == (g["sparse_reps"] > 0)).mean() > 0.985
    tied = np.load(os.path.join(GOLDEN, "sparse.npz"))["sparse_reps"]
    assert np.abs(sp - tied).max() > 0.5                           # and it is NOT what the embedding matrix would give
    assert min_cos(out["dense_reps"].cpu().numpy(), g["dense_reps"]) > 0.998
