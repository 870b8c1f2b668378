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


@pytest.mark.parametrize("method", ["rrf", "linear"])
def test_array_form_top1000_lists(method):
    """Two systems x top-1000 per query (the eval setting): union, sums, ordering and padding against the oracle."""
    from lightretriever_amd.score_fuse_utils import fuse_hits
    rng = np.random.default_rng(3)
    Q, k, N = 9, 1000, 5000
    sys_np = []
    for s in range(2):
        ids = np.stack([rng.choice(N, size=k, replace=False) for _ in range(Q)]).astype(np.int64)
        sc = np.sort(rng.standard_normal((Q, k)).astype(np.float32), axis=1)[:, ::-1].copy()      # sorted hit lists, like a search returns
        ids[2, 700:] = -1                                                                         # a short list (k > hits)
        if s == 1:
            ids[5, :] = -1                                                                        # a query the second system has nothing for
        sys_np.append((sc, ids))
    sc, ids, cnt = fuse_hits([(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()) for a, b in sys_np], method=method, k=60,
                             weights=[0.7, 0.3])
    sc, ids, cnt = sc.cpu().numpy(), ids.cpu().numpy(), cnt.cpu().numpy()
    dicts = [{str(q): {str(int(p)): float(v) for p, v in zip(b[q], a[q]) if p >= 0} for q in range(Q) if (b[q] >= 0).any()} for a, b in sys_np]
    want = O.fuse_scores_rrf(dicts, k=60) if method == "rrf" else O.fuse_scores_linear(dicts, [0.7, 0.3])
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
        fuse_hits([a, a])                       # 3000 entries per query exceed the 2048-entry workgroup sort
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
