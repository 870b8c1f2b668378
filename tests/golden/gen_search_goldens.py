#!/usr/bin/env python3
"""Golden vectors for the SEARCH half of the path, produced by running the reference's own Python.

Executed reference code (imported from /root/reference/src, nothing copied):
  * retriever/faiss_index.py   FaissIndex.build (:45-58, the 50 000-row add loop) and FaissIndex.search (:27-40, the
                               `_passage_ids` row -> id mapping)
  * retriever/faiss_search.py  FlatIPFaissSearch.index / _index / _create_mapping_ids (:490-504, :125-134, :82-86),
                               retrieve_with_emb (:143-173), DenseRetrievalFaissSearch.search (:176-291: corpus sort, chunk
                               loop, heap merge, ignore_identical_ids)
  * retriever/hybrid_search.py HybridSearch.search (:234-403) with its _add_to_heap (:182-205), _parse_heap_results
                               (:347-355), retrieve_with_emb (:121-180), index (:106-119), _clear

`faiss` is not installed in this image and not vendored by the reference (pyproject.toml:6 `faiss>=1.7.4`).  The module
below named `faiss` is a numpy STAND-IN with the few entry points the reference calls (IndexFlatIP.add/search/reset/ntotal):
exact fp32 inner products, hits by (score descending, row ascending), (-FLT_MAX, -1) padding when k > ntotal.  It pins
NOTHING about Faiss itself -- Faiss stays "parity unpinned" -- it only lets the reference's own code around the index run,
so that everything the reference does WITH the index's answers (id mapping, float conversion, chunk loop, heap order with
its (score, pid-string) tuple comparison, identical-id removal, dict assembly) is pinned by its real output.

Two data sets:
  * "dyadic": vectors whose components are multiples of 1/8 with |x| <= 1/2, D = 32.  Every inner product is a multiple of
    1/64 below 8, exactly representable and exact under ANY summation order and in bf16 -- the stand-in, the numpy oracle
    and the HIP kernels must agree on every score BITWISE, and exact score ties are everywhere (also across chunk
    boundaries and at the k-th place), which is what exercises the heap's tie rule.
  * "random": L2-normalised gaussian vectors, D = 64 (no ties; scores to 2e-6, ids identical).

Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_search_goldens.py     (writes tests/golden/search_ref.json)
"""
import json
import os
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/src"

import numpy as np  # noqa: E402
import torch  # noqa: E402

FLT_MAX = np.finfo(np.float32).max


# ---- the numpy stand-in for the absent third-party module ----------------------------------------------------------
def _make_faiss_stub():
    m = types.ModuleType("faiss")

    class Index:
        pass

    class IndexFlatIP(Index):
        def __init__(self, d):
            self.d, self.ntotal, self._x = int(d), 0, np.zeros((0, int(d)), np.float32)

        def add(self, x):
            x = np.ascontiguousarray(np.asarray(x), dtype=np.float32)
            assert x.ndim == 2 and x.shape[1] == self.d
            self._x = np.concatenate([self._x, x], 0)
            self.ntotal = self._x.shape[0]

        def search(self, q, k):
            q = np.ascontiguousarray(np.asarray(q), dtype=np.float32)
            S = (q.astype(np.float64) @ self._x.T.astype(np.float64)).astype(np.float32)   # exact products, one rounding
            Q, N = S.shape
            D = np.full((Q, k), -FLT_MAX, np.float32)
            I = np.full((Q, k), -1, np.int64)
            kk = min(k, N)
            if kk:
                order = np.lexsort((np.broadcast_to(np.arange(N), S.shape), -S), axis=-1)[:, :kk]
                D[:, :kk] = np.take_along_axis(S, order, axis=1)
                I[:, :kk] = order
            return D, I

        def reset(self):
            self._x, self.ntotal = np.zeros((0, self.d), np.float32), 0

    m.Index, m.IndexFlatIP = Index, IndexFlatIP
    m.METRIC_INNER_PRODUCT, m.METRIC_L2 = 0, 1          # default arguments evaluated when faiss_search.py is imported
    m.get_num_gpus = lambda: 0
    return m


import datasets  # noqa: E402,F401  (before the stand-in exists: `datasets` probes find_spec("faiss") at import)

sys.modules["faiss"] = _make_faiss_stub()
sys.path.insert(0, REF)

from lightretriever.retriever.faiss_index import FaissIndex  # noqa: E402
from lightretriever.retriever.faiss_search import FlatIPFaissSearch  # noqa: E402
from lightretriever.retriever.hybrid_search import HybridSearch  # noqa: E402


def b64(a):
    """ndarray -> JSON-able dict (raw little-endian bytes, base64): exact and 5x smaller than decimal text."""
    import base64
    a = np.ascontiguousarray(a)
    return {"dtype": str(a.dtype), "shape": list(a.shape), "b64": base64.b64encode(a.tobytes()).decode()}


class PrecomputedModel:
    """Duck-typed B2 model: returns fixed embeddings (documents are found by their text)."""

    def __init__(self, X, text_row, q_by_kind, as_dict=True):
        self.X, self.text_row, self.q_by_kind, self.as_dict = X, text_row, q_by_kind, as_dict
        self.corpus_calls = []

    def encode_queries(self, queries, batch_size=None, **kw):
        if self.as_dict:
            return {k: torch.from_numpy(v) for k, v in self.q_by_kind.items()}
        return torch.from_numpy(next(iter(self.q_by_kind.values())))

    def encode_corpus(self, corpus, batch_size=None, **kw):
        rows = [self.text_row[d["text"]] for d in corpus]
        self.corpus_calls.append(len(rows))
        e = torch.from_numpy(self.X[rows])
        return {"dense_reps": e} if self.as_dict else e


class NullSparse:
    def _clear(self):
        pass


def make_hybrid(model, batch_size, chunk):
    """HybridSearch without its constructor's unconditional Anserini/JVM import (hybrid_search.py:78-84)."""
    hs = HybridSearch.__new__(HybridSearch)
    hs.model, hs.batch_size, hs.corpus_chunk_size = model, batch_size, chunk
    hs.show_progress_bar, hs.convert_to_tensor = False, True
    hs.score_fuse_method, hs.fuse_weights, hs.return_all_results = "linear", [0.7, 0.3], True
    hs.dense_search = FlatIPFaissSearch(model=model, batch_size=batch_size, corpus_chunk_size=chunk, show_progress_bar=False)
    hs.sparse_search = NullSparse()
    return hs


def make_corpus(rng, n, id_fmt):
    """Texts of many different (and many equal: the sort is stable) lengths; every text unique."""
    corpus, text_row = {}, {}
    for i in range(n):
        ln = int(rng.integers(3, 40))
        text = (f"{i:05d}" + "x" * ln)[: 5 + ln]
        d = {"text": text}
        if i % 3 == 0:
            d["title"] = "t" * int(rng.integers(0, 60))      # titles do not take part in the sort key
        corpus[id_fmt(i)] = d
        text_row[text] = i
    return corpus, text_row


def run_case(name, X, Qs, qids, corpus, text_row, top_k, chunk, ignore):
    """-> dict with the outputs of both reference searchers (or the exception the reference raises)."""
    out = {"name": name, "top_k": top_k, "corpus_chunk_size": chunk, "ignore_identical_ids": ignore, "query_ids": qids}
    queries = {q: f"query {q}" for q in qids}
    try:
        m = PrecomputedModel(X, text_row, Qs)
        res = make_hybrid(m, 16, chunk).search(corpus, queries, top_k=top_k, ignore_identical_ids=ignore)
        out["hybrid"] = res                        # {"den": {qid: {pid: score}}, "emb": {...}}
        out["chunk_sizes"] = m.corpus_calls
    except KeyError as e:
        out["hybrid_raises"] = f"KeyError({e})"
    try:
        m = PrecomputedModel(X, text_row, {"emb_reps": Qs["emb_reps"]}, as_dict=False)
        out["flat"] = FlatIPFaissSearch(model=m, batch_size=16, corpus_chunk_size=chunk, show_progress_bar=False).search(
            corpus, queries, top_k=top_k, ignore_identical_ids=ignore)
    except KeyError as e:
        out["flat_raises"] = f"KeyError({e})"
    return out


def persist_goldens():
    """Row f-N4 (index persistence): the `.tsv` id map exactly as the reference's own save_dict_to_tsv writes it
    (retriever/faiss_search.py:28-33: csv.writer, tab delimiter, QUOTE_MINIMAL, header row from mapping_tsv_keys :63), read back by
    its load_tsv_to_dict (:35-43), and the file name / id order DenseRetrievalFaissSearch.save and ._load derive (:99-123).  The ids
    carry everything the csv dialect treats specially: tabs, double quotes, commas, leading / trailing blanks, non-ASCII, an
    embedded line feed, the empty string.  Bytes are stored base64; nothing of the reference's source is."""
    import base64
    import tempfile
    from lightretriever.retriever import faiss_search as ref_fs
    ids = ["doc-0", "with\ttab", 'say "hi"', "comma,inside", " lead", "trail ", "\u00fcml\u00e4ut-\u6587\u66f8", "multi\nline", "", "quote\"and\ttab", "123", "d9", "d10"]
    mapping = {pid: i for i, pid in enumerate(ids)}
    out = {"generator": "tests/golden/gen_search_goldens.py:persist_goldens", "ids": ids}
    with tempfile.TemporaryDirectory() as td:
        f = os.path.join(td, "m.tsv")
        ref_fs.save_dict_to_tsv(mapping, f, keys=["beir-docid", "faiss-docid"])
        out["tsv_with_header_b64"] = base64.b64encode(open(f, "rb").read()).decode()
        back = ref_fs.load_tsv_to_dict(f, header=True)
        out["loaded_with_header"] = [[k, v] for k, v in back.items()]
        ref_fs.save_dict_to_tsv(mapping, f)                                     # keys=[] -> no header row
        out["tsv_no_header_b64"] = base64.b64encode(open(f, "rb").read()).decode()
        out["loaded_no_header"] = [[k, v] for k, v in ref_fs.load_tsv_to_dict(f, header=False).items()]

        # DenseRetrievalFaissSearch.save / ._load around a recording index object (the .faiss bytes themselves are Faiss's: unpinnable here)
        class RecordingIndex:
            def __init__(self):
                self.saved = []

            def save(self, fname):
                self.saved.append(os.path.basename(fname))
                open(fname, "wb").write(b"x")

        fs = FlatIPFaissSearch(model=None, batch_size=8, show_progress_bar=False)
        fs.mapping = {pid: i for i, pid in enumerate(ids[:6])}
        fs.faiss_index = RecordingIndex()
        ref_fs.DenseRetrievalFaissSearch.save(fs, td, "my-index", "flat")
        out["save_files"] = sorted(os.listdir(td))
        out["save_index_file"] = fs.faiss_index.saved
        out["save_tsv_b64"] = base64.b64encode(open(os.path.join(td, "my-index.flat.tsv"), "rb").read()).decode()
        fs2 = FlatIPFaissSearch(model=None, batch_size=8, show_progress_bar=False)
        path, passage_ids = ref_fs.DenseRetrievalFaissSearch._load(fs2, td, "my-index", "flat")
        out["load_faiss_path_basename"] = os.path.basename(path)
        out["load_passage_ids"] = passage_ids
        out["load_mapping"] = [[k, v] for k, v in fs2.mapping.items()]
        out["load_rev_mapping"] = [[k, v] for k, v in fs2.rev_mapping.items()]
    with open(os.path.join(HERE, "persist_ref.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print(f"wrote persist_ref.json: {len(ids)} ids, {os.path.getsize(os.path.join(HERE, 'persist_ref.json'))} bytes")


def main():
    persist_goldens()
    rng = np.random.default_rng(20260203)
    fx = {"generator": "tests/golden/gen_search_goldens.py", "faiss": "numpy stand-in (pins nothing about Faiss)", "sets": {}}

    # ---- dyadic set: exact arithmetic, ties everywhere --------------------------------------------------------------
    N, D, Q = 300, 32, 9
    X = (rng.integers(-4, 5, size=(N, D)) / 8.0).astype(np.float32)
    X[250] = X[17]; X[251] = X[17]; X[40] = X[17]; X[299] = X[17]            # the same row in four different chunks
    X[100:130] = X[100]                                                        # 30 identical rows: a tie run longer than top_k
    Qe = (rng.integers(-4, 5, size=(Q, D)) / 8.0).astype(np.float32)
    Qd = (rng.integers(-4, 5, size=(Q, D)) / 8.0).astype(np.float32)
    Qe[0] = X[17]; Qe[1] = X[100]
    # ids as MTEB hands them over: strings whose lexicographic order is not the numeric one ("d9" > "d10")
    corpus, text_row = make_corpus(rng, N, lambda i: f"d{i}")
    qids = ["d17", "q1", "d100", "q3", "d250", "q5", "q6", "d299", "q8"]       # some query ids equal document ids
    dy = {"X": b64(X), "emb_reps": b64(Qe), "dense_reps": b64(Qd),
          "corpus": corpus, "cases": []}
    Qs = {"dense_reps": Qd, "emb_reps": Qe}
    for (k, chunk, ign) in [(10, 64, True), (10, 64, False), (25, 50, True), (7, 300, False), (3, 16, True), (40, 100, True),
                            (50, 64, True)]:                                    # last: final chunk (44 rows) < top_k
        dy["cases"].append(run_case(f"dyadic_k{k}_c{chunk}_ign{int(ign)}", X, Qs, qids, corpus, text_row, k, chunk, ign))
    # a corpus smaller than top_k (first 30 documents): Faiss pads with id -1, which the reference's numpy lookup
    # `_passage_ids[-1]` (faiss_index.py:34) turns into the chunk's LAST document, and `dict(zip(doc_ids, scores))`
    # (faiss_search.py:171) then overwrites that document's real score with -FLT_MAX.  Recorded as the reference behaves.
    sub_ids = [f"d{i}" for i in range(30)]
    sub_corpus = {c: corpus[c] for c in sub_ids}
    c = run_case("dyadic_first30_k50_c64_ign0", X, Qs, qids, sub_corpus, text_row, 50, 64, False)
    c["corpus_ids"] = sub_ids
    dy["cases"].append(c)
    fx["sets"]["dyadic"] = dy
    fx_dy_arrays = (X.copy(), Qe.copy(), Qd.copy())

    # ---- random set: no ties -----------------------------------------------------------------------------------------
    N, D, Q = 500, 64, 6
    X = rng.standard_normal((N, D)).astype(np.float32)
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    Qe = rng.standard_normal((Q, D)).astype(np.float32); Qe /= np.linalg.norm(Qe, axis=1, keepdims=True)
    Qd = rng.standard_normal((Q, D)).astype(np.float32); Qd /= np.linalg.norm(Qd, axis=1, keepdims=True)
    corpus, text_row = make_corpus(rng, N, lambda i: f"doc-{i}")
    qids = ["doc-3", "qa", "qb", "doc-499", "qc", "qd"]
    Qe[0] = X[3]; Qe[3] = X[499]                                               # the identical document would be the top hit
    rd = {"X": b64(X), "emb_reps": b64(Qe), "dense_reps": b64(Qd), "corpus": corpus, "cases": []}
    Qs = {"dense_reps": Qd, "emb_reps": Qe}
    for (k, chunk, ign) in [(20, 128, True), (100, 250, False), (5, 500, True)]:
        rd["cases"].append(run_case(f"random_k{k}_c{chunk}_ign{int(ign)}", X, Qs, qids, corpus, text_row, k, chunk, ign))
    fx["sets"]["random"] = rd

    # ---- retrieve_with_emb + FaissIndex.search id mapping, straight ----------------------------------------------------
    Xd, Qe_d, Qd_d = fx_dy_arrays
    ids = [f"p{i}" for i in range(Xd.shape[0])]
    fs = FlatIPFaissSearch(model=None, batch_size=8, show_progress_bar=False)
    fs.index(torch.from_numpy(Xd), ids)
    fx["retrieve_with_emb"] = {"set": "dyadic", "ids_fmt": "p{i}", "top_k": 12,
                               "result": fs.retrieve_with_emb(Qe_d, [f"q{i}" for i in range(9)], top_k=12)}
    perm = rng.permutation(Xd.shape[0]).astype(np.int64) + 1000               # FaissIndex with an arbitrary passage-id array
    fi = FaissIndex.build(perm.tolist(), Xd, buffer_size=77)                   # 77-row add slices
    Ds, Is = fi.search(Qd_d, 8)
    fx["faiss_index_search"] = {"set": "dyadic", "passage_ids": perm.tolist(), "k": 8, "D": Ds.tolist(), "I": Is.tolist()}

    with open(os.path.join(HERE, "search_ref.json"), "w") as f:
        json.dump(fx, f, separators=(",", ":"))
    n_cases = sum(len(s["cases"]) for s in fx["sets"].values())
    raised = [c["name"] for s in fx["sets"].values() for c in s["cases"] if "hybrid_raises" in c]
    print(f"wrote search_ref.json: {n_cases} cases, reference raised in {raised}, {os.path.getsize(os.path.join(HERE, 'search_ref.json'))} bytes")


if __name__ == "__main__":
    main()
