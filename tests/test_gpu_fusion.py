"""GPU parity for hit-list fusion: the device kernels behind fuse_scores_rrf / fuse_scores_linear against the
reference's own outputs (tests/golden/fusion.json) -- exact float64 equality -- and the array form against the oracle."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import lrx_oracle as O
from helpers import GOLDEN

pytestmark = pytest.mark.gpu


def golden():
    with open(os.path.join(GOLDEN, "fusion.json")) as f:
        return json.load(f)


def test_dict_form_equals_reference_bit_for_bit():
    from lightretriever_amd.score_fuse_utils import fuse_scores_linear, fuse_scores_rrf
    g = golden()
    two, three = [g["dense"], g["sparse"]], [g["dense"], g["sparse"], g["third"]]
    assert fuse_scores_rrf(two) == g["rrf"]
    assert fuse_scores_rrf(two, k=10) == g["rrf_k10"]
    assert fuse_scores_rrf(three) == g["rrf_three"]
    assert fuse_scores_linear(two, weights=[0.7, 0.3]) == g["linear"]
    assert fuse_scores_linear(two, weights=[0.5, 0.5], eps=1e-6) == g["linear_5050"]
    assert fuse_scores_linear(three, weights=[0.5, 0.3, 0.2]) == g["linear_three"]
    assert fuse_scores_rrf([]) == {}
    with pytest.raises(AssertionError):
        fuse_scores_linear(two, weights=[1.0])


@pytest.mark.parametrize("method,n_sys", [("rrf", 2), ("linear", 2), ("rrf", 3), ("linear", 3)])
def test_array_form_top1000_lists(method, n_sys):
    """Two systems x top-1000 per query (the eval setting) and three (the reference accepts any number of lists): union, sums,
    ordering and padding against the oracle."""
    from lightretriever_amd.score_fuse_utils import fuse_hits
    rng = np.random.default_rng(3)
    Q, k, N = 9, 1000, 5000
    sys_np = []
    for s in range(n_sys):
        ids = np.stack([rng.choice(N, size=k, replace=False) for _ in range(Q)]).astype(np.int64)
        sc = np.sort(rng.standard_normal((Q, k)).astype(np.float32), axis=1)[:, ::-1].copy()      # sorted hit lists, like a search returns
        ids[2, 700:] = -1                                                                         # a short list (k > hits)
        if s == 1:
            ids[5, :] = -1                                                                        # a query the second system has nothing for
        sys_np.append((sc, ids))
    sc, ids, cnt = fuse_hits([(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()) for a, b in sys_np], method=method, k=60,
                             weights=[0.7, 0.3, 0.2][:n_sys])
    sc, ids, cnt = sc.cpu().numpy(), ids.cpu().numpy(), cnt.cpu().numpy()
    dicts = [{str(q): {str(int(p)): float(v) for p, v in zip(b[q], a[q]) if p >= 0} for q in range(Q) if (b[q] >= 0).any()} for a, b in sys_np]
    want = O.fuse_scores_rrf(dicts, k=60) if method == "rrf" else O.fuse_scores_linear(dicts, [0.7, 0.3, 0.2][:n_sys])
    for q in range(Q):
        w = want[str(q)]
        assert cnt[q] == len(w)
        got = {str(int(p)): float(v) for p, v in zip(ids[q, :cnt[q]], sc[q, :cnt[q]])}
        assert got == w                                                                           # bit-identical float64 sums
        assert (np.diff(sc[q, :cnt[q]]) <= 0).all()                                               # sorted by fused score
        assert (ids[q, cnt[q]:] == -1).all() and np.isneginf(sc[q, cnt[q]:]).all()
        ties = np.flatnonzero(np.diff(sc[q, :cnt[q]]) == 0)
        assert (ids[q, ties] < ids[q, ties + 1]).all()                                            # lower id first among equal scores


def test_fuse_argument_errors():
    from lightretriever_amd._lib import LrxError
    from lightretriever_amd.score_fuse_utils import fuse_hits
    a = (torch.zeros(2, 1500, device="cuda"), torch.zeros(2, 1500, dtype=torch.int64, device="cuda"))
    with pytest.raises(LrxError):
        fuse_hits([a, a, a])                    # 4500 entries per query exceed the 4096-entry workgroup sort
    with pytest.raises(NotImplementedError):
        fuse_hits([a], method="borda")


class _DictSparseEngine:
    """Stand-in for the (out-of-scope) Lucene impact index with AnseriniSearch's interface: exact integer dot products of the
    query token counts with the quantised document vectors, in plain Python."""

    def __init__(self):
        self._clear()

    def _clear(self):
        self.docs, self.ids = [], []

    def index(self, corpus_emb, corpus_ids):
        self.docs += list(corpus_emb)
        self.ids += list(corpus_ids)

    def retrieve_with_emb(self, query_emb, query_ids, top_k):
        out = {}
        for qid, q in zip(query_ids, query_emb):
            sc = {pid: float(sum(v * d[t] for t, v in q.items() if t in d)) for pid, d in zip(self.ids, self.docs)}
            out[qid] = dict(sorted(((p, s) for p, s in sc.items() if s > 0), key=lambda kv: -kv[1])[:top_k])
        return out


@pytest.mark.parametrize("method", ["linear", "rrf"])
def test_hybrid_search_fuses_dense_and_sparse_hits(method):
    """B1 level: HybridSearch.index / retrieve_with_emb with a sparse engine plugged in -> emb, tok and their fusion, equal to the
    reference's fusion functions (restated by the oracle) applied to the same two hit lists."""
    from lightretriever_amd.retriever import HybridSearch
    rng = np.random.default_rng(1)
    N, D, Q = 300, 64, 5
    X = O.l2_normalize(rng.standard_normal((N, D)).astype(np.float32))
    q = O.l2_normalize(rng.standard_normal((Q, D)).astype(np.float32))
    docs_sparse = [{str(int(t)): int(w) for t, w in zip(rng.choice(50, 12, replace=False), rng.integers(1, 300, 12))} for _ in range(N)]
    q_tok = [{str(int(t)): int(c) for t, c in zip(rng.choice(50, 4, replace=False), rng.integers(1, 3, 4))} for _ in range(Q)]
    cids, qids = ["d%d" % i for i in range(N)], ["q%d" % i for i in range(Q)]
    hs = HybridSearch(model=None, batch_size=16, score_fuse_method=method, fuse_weights=[0.6, 0.4], sparse_search=_DictSparseEngine())
    hs.index({"dense_reps": torch.from_numpy(X).cuda(), "sparse_reps": docs_sparse}, cids)
    res = hs.retrieve_with_emb({"emb_reps": torch.from_numpy(q).cuda(), "token_id_reps": q_tok}, qids, top_k=40)
    assert set(res) == {"emb", "tok", "emb_tok"}
    want = O.fuse_scores_rrf([res["emb"], res["tok"]]) if method == "rrf" else O.fuse_scores_linear([res["emb"], res["tok"]], [0.6, 0.4])
    assert res["emb_tok"] == want
    hs._clear()
    assert hs.sparse_search.ids == []


class _DictImpactEngine:
    """Stand-in for the reference's AnseriniSearch (JVM/Lucene, out of scope): exact impact scores = sum_t count_q[t] * weight_d[t]
    over the documents indexed so far; same interface (index / retrieve_with_emb / _clear)."""

    def __init__(self):
        self.docs = {}

    def index(self, corpus_emb, corpus_ids):
        assert len(corpus_emb) == len(corpus_ids)
        self.docs.update(dict(zip(corpus_ids, corpus_emb)))

    def retrieve_with_emb(self, query_emb, query_ids, top_k, **kw):
        out = {}
        from collections import Counter
        for qid, qv in zip(query_ids, query_emb):
            if isinstance(qv, str):                                  # pseudo text of an LM-head sparse query vector: tokens repeated by weight
                qv = Counter(qv.split())
            sc = {pid: float(sum(c * dv.get(t, 0) for t, c in qv.items())) for pid, dv in self.docs.items()}
            top = sorted(((s, p) for p, s in sc.items() if s > 0), key=lambda x: (-x[0], x[1]))[:top_k]
            out[qid] = {p: s for s, p in top}
        return out

    def _clear(self):
        self.docs = {}


def test_hybrid_search_feeds_a_sparse_engine_per_chunk_and_fuses_at_the_end():
    """hybrid_search.py:301-403 with a sparse engine plugged in: every chunk's quantised document vectors are indexed as the chunk
    is encoded, `tok` is one retrieval over the whole engine with the queries' token counts, `emb_tok` the fusion of the final `emb`
    and `tok` lists (not of per-chunk lists)."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_gpu_api import build_stack, synth_corpus
    from helpers import load_model_golden
    from lightretriever_amd.modeling import LrxExactSearchModel, LrxHybridModel
    from lightretriever_amd.retriever import HybridSearch
    from lightretriever_amd.score_fuse_utils import fuse_scores_linear
    cfg_o, w, _, _, _, _ = load_model_golden("llama_small_d64")
    tok, enc, hm, _ = build_stack(cfg_o, w)
    hs = LrxHybridModel(enc, normalize=True, pad_token_id=tok.pad_token_id, encode_sparse=True, sparse_top_k_psg=24)
    model = LrxExactSearchModel(model=hs, tokenizer=tok, q_max_len=32, p_max_len=64, eval_batch_size_embedding_bag=100)
    model.query_prompt = "query: "
    corpus = synth_corpus(np.random.default_rng(2), 60)
    queries = {"q0": "capital of france paris", "q1": "dense retrieval with large language models", "q2": "amd instinct memory search"}
    engine = _DictImpactEngine()
    searcher = HybridSearch(model, batch_size=8, corpus_chunk_size=25, fuse_weights=[0.6, 0.4], return_all_results=True, sparse_search=engine)
    res = searcher.search(corpus, queries, top_k=10)
    assert set(res) == {"emb", "tok", "emb_tok"} and engine.docs == {}          # three chunks went in, cleared at the end
    # the same three lists put together by hand
    docs = list(corpus.values())
    enc_all = model.encode_corpus(docs, batch_size=8)
    ref_engine = _DictImpactEngine()
    ref_engine.index(enc_all["sparse_reps"], list(corpus))
    qe = model.encode_queries(list(queries.values()), batch_size=8)
    tok_want = ref_engine.retrieve_with_emb(qe["token_id_reps"], list(queries), 10)
    assert res["tok"] == tok_want and all(len(v) > 0 for v in tok_want.values())
    emb_only = HybridSearch(model, batch_size=8, corpus_chunk_size=25, return_all_results=True).search(corpus, queries, top_k=10)["emb"]
    assert res["emb"] == emb_only
    assert res["emb_tok"] == fuse_scores_linear([emb_only, tok_want], weights=[0.6, 0.4])
    assert HybridSearch(model, batch_size=8, corpus_chunk_size=25, fuse_weights=[0.6, 0.4], sparse_search=_DictImpactEngine()).search(
        corpus, queries, top_k=10) == res["emb_tok"]                              # default result = the fused list


def test_hybrid_search_serves_the_spr_and_den_spr_query_modes():
    """Round 6 (`--hybrid_use_sparse_vector`): LM-head sparse query vectors reach the sparse engine as pseudo text; result kinds and the default
    follow retriever/hybrid_search.py:364-403 -- den, spr, den_spr (the default when both are on), next to emb / tok / emb_tok when those flags
    are on too."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_gpu_api import build_stack, synth_corpus
    from helpers import load_model_golden
    from lightretriever_amd.modeling import LrxExactSearchModel, LrxHybridModel
    from lightretriever_amd.retriever import HybridSearch
    from lightretriever_amd.score_fuse_utils import fuse_scores_linear
    cfg_o, w, _, _, _, _ = load_model_golden("llama_small_d64")
    tok, enc, _, _ = build_stack(cfg_o, w)
    hs = LrxHybridModel(enc, normalize=True, pad_token_id=tok.pad_token_id, encode_sparse=True, sparse_top_k_psg=24, hybrid_use_sparse_vector=True,
                        hybrid_use_dense_vector=True, hybrid_use_emb_vector=False, sparse_top_k_qry=12)
    model = LrxExactSearchModel(model=hs, tokenizer=tok, q_max_len=32, p_max_len=64)
    corpus = synth_corpus(np.random.default_rng(2), 60)
    queries = {"q0": "capital of france paris", "q1": "dense retrieval with large language models", "q2": "amd instinct memory search"}
    res = HybridSearch(model, batch_size=8, corpus_chunk_size=25, fuse_weights=[0.6, 0.4], return_all_results=True, sparse_search=_DictImpactEngine()).search(
        corpus, queries, top_k=10)
    assert list(res) == ["den", "spr", "den_spr"]
    enc_all = model.encode_corpus(list(corpus.values()), batch_size=8)
    ref_engine = _DictImpactEngine()
    ref_engine.index(enc_all["sparse_reps"], list(corpus))
    qe = model.encode_queries(list(queries.values()), batch_size=8)
    assert set(qe) == {"dense_reps", "sparse_reps"} and all(isinstance(t, str) for t in qe["sparse_reps"])
    spr_want = ref_engine.retrieve_with_emb(qe["sparse_reps"], list(queries), 10)
    assert res["spr"] == spr_want and all(len(v) > 0 for v in spr_want.values())
    den_only = HybridSearch(model, batch_size=8, corpus_chunk_size=25, return_all_results=True).search(corpus, queries, top_k=10)["den"]
    assert res["den"] == den_only and res["den_spr"] == fuse_scores_linear([den_only, spr_want], weights=[0.6, 0.4])
    assert HybridSearch(model, batch_size=8, corpus_chunk_size=25, fuse_weights=[0.6, 0.4], sparse_search=_DictImpactEngine()).search(
        corpus, queries, top_k=10) == res["den_spr"]                              # default = the last kind set, like the reference
