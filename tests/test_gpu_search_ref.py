"""The product searchers against tests/golden/search_ref.json = outputs of the REFERENCE's own HybridSearch.search /
FlatIPFaissSearch.search / retrieve_with_emb / FaissIndex.search (tests/golden/gen_search_goldens.py).  Dyadic data: every score
is exact in fp32 under any summation order, so the dicts must be EQUAL, ties (chunk boundaries, k-th place) included."""
import numpy as np
import pytest
import torch

from oracle import lrx_oracle as O
from helpers import load_search_ref

pytestmark = pytest.mark.gpu
FLT_MAX = float(np.finfo(np.float32).max)


class PrecomputedModel:
    """Duck-typed B2 model returning fixed embeddings (documents are recognised by their text)."""

    def __init__(self, X, text_row, qs, as_dict=True):
        self.X, self.text_row, self.qs, self.as_dict = X, text_row, qs, as_dict

    def encode_queries(self, queries, batch_size=None, **kw):
        if self.as_dict:
            return {k: torch.from_numpy(v) for k, v in self.qs.items()}
        return torch.from_numpy(self.qs["emb_reps"])

    def encode_corpus(self, corpus, batch_size=None, out=None, **kw):
        e = torch.from_numpy(self.X[[self.text_row[d["text"]] for d in corpus]])
        return {"dense_reps": e} if self.as_dict else e


def _inputs(st, case):
    corpus = {c: st["corpus"][c] for c in case.get("corpus_ids", st["corpus"])}
    text_row = {st["corpus"][c]["text"]: i for i, c in enumerate(st["corpus"])}
    queries = {q: f"query {q}" for q in case["query_ids"]}
    return corpus, text_row, queries


def _short_chunk(case, n):
    c, k = case["corpus_chunk_size"], case["top_k"]
    return any(min(s + c, n) - s < k for s in range(0, n, c))


@pytest.mark.parametrize("set_name", ["dyadic", "random"])
def test_searchers_equal_the_reference_outputs(set_name):
    from lightretriever_amd.retriever import FlatIPFaissSearch, HybridSearch
    fx = load_search_ref()
    st = fx["sets"][set_name]
    qs = {k: st[k] for k in ("dense_reps", "emb_reps")}
    n_checked = 0
    for case in st["cases"]:
        corpus, text_row, queries = _inputs(st, case)
        kw = dict(top_k=case["top_k"], ignore_identical_ids=case["ignore_identical_ids"])
        hs = HybridSearch(PrecomputedModel(st["X"], text_row, qs), batch_size=16, corpus_chunk_size=case["corpus_chunk_size"],
                          return_all_results=True, show_progress_bar=False)
        got = hs.search(corpus, queries, **kw)
        flat = FlatIPFaissSearch(PrecomputedModel(st["X"], text_row, qs, as_dict=False), batch_size=16,
                                 corpus_chunk_size=case["corpus_chunk_size"], show_progress_bar=False).search(corpus, queries, **kw)
        if _short_chunk(case, len(corpus)):
            # a chunk shorter than top_k: the reference loses that chunk's last document (padding id -1 -> numpy [-1]); the product
            # keeps it -- checked against the oracle's restatement without that artefact (pinned to the reference WITH it on CPU)
            cids = O.sort_corpus_ids_longest_first(corpus)
            all_ids = list(st["corpus"])
            emb = st["X"][[all_ids.index(c) for c in cids]]
            want = {n: O.search_chunks(qs[kd], case["query_ids"], emb, cids, case["top_k"], case["corpus_chunk_size"],
                                       case["ignore_identical_ids"]) for kd, n in (("dense_reps", "den"), ("emb_reps", "emb"))}
        else:
            want = case["hybrid"]
            assert case["flat"] == case["hybrid"]["emb"]
        for name in ("den", "emb"):
            for qid in case["query_ids"]:
                g, w = got[name][qid], want[name][qid]
                if set_name == "dyadic":
                    assert g == w, (case["name"], name, qid, sorted(set(g) ^ set(w)))
                else:
                    assert set(g) == set(w), (case["name"], name, qid)
                    np.testing.assert_allclose([g[p] for p in w], list(w.values()), atol=2e-6)
                n_checked += 1
        for qid in case["query_ids"]:
            if set_name == "dyadic":
                assert flat[qid] == want["emb"][qid], (case["name"], "flat", qid)
            else:
                assert set(flat[qid]) == set(want["emb"][qid])
    assert n_checked >= 2 * 6 * 3


def test_retrieve_with_emb_and_faiss_index_search_equal_the_reference_outputs():
    from lightretriever_amd.retriever import FaissIndex, FlatIPFaissSearch
    fx = load_search_ref()
    st = fx["sets"]["dyadic"]
    X = st["X"]
    r = fx["retrieve_with_emb"]
    fs = FlatIPFaissSearch(model=None, batch_size=8)
    fs.index(torch.from_numpy(X), [f"p{i}" for i in range(len(X))])
    got = fs.retrieve_with_emb(st["emb_reps"], list(r["result"]), top_k=r["top_k"])
    assert got == r["result"]
    assert [list(got[q]) for q in got] == [list(r["result"][q]) for q in got]      # order of the hits inside each dict too
    f = fx["faiss_index_search"]
    fi = FaissIndex.build(f["passage_ids"], torch.from_numpy(X), buffer_size=77)
    D, I = fi.search(torch.from_numpy(st["dense_reps"]).cuda(), f["k"])
    np.testing.assert_array_equal(I.cpu().numpy(), np.asarray(f["I"]))
    np.testing.assert_array_equal(D.cpu().numpy(), np.asarray(f["D"], np.float32))
