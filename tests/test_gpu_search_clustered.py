"""Exactness of the bounded search on NON-iid corpora at the headline index size: 1 000 vMF-like clusters with
intra-cluster cosine ~0.9, 1 % exact duplicates, queries near cluster centres -- rows iid over the clusters and rows stored cluster by
cluster (a strided block sample then misses whole clusters).  The reference is an fp64 evaluation of every inner product on the same GPU,
rounded once (the product's definition of the result, tests/helpers.py:flat_ip_topk_fp64), ties to the lower row.  What replaces:
faiss.IndexFlatIP.search behind retriever/faiss_index.py:60-70."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def topk_fp64_gpu(q, X, k, chunk=65536):
    """(scores fp32 [Q,k], rows int64 [Q,k]): fp64-accumulated products rounded once, (score desc, row asc) -- by a top-k over the packed
    64-bit (monotone score key, ~row) words, merged chunk by chunk."""
    Q = q.shape[0]
    best = torch.full((Q, 0), 0, dtype=torch.int64, device=q.device)
    qd = q.double()
    for s in range(0, X.shape[0], chunk):
        e = min(s + chunk, X.shape[0])
        sc = (qd @ X[s:e].double().T).float()
        u = sc.view(torch.int32).to(torch.int64)
        key = torch.where(u < 0, -u - 1 - 0x80000000, u)                        # monotone in the float, as a signed 32-bit value
        rows = torch.arange(s, e, device=q.device, dtype=torch.int64)
        words = (key << 32) | (0xFFFFFFFF - rows)[None, :]
        best = torch.cat([best, words], 1).topk(min(k, best.shape[1] + e - s), dim=1).values
    rows = 0xFFFFFFFF - (best & 0xFFFFFFFF)
    key = best >> 32
    u = torch.where(key < 0, -(key + 1 + 0x80000000), key).to(torch.int32)
    return u.view(torch.float32), rows


@pytest.mark.parametrize("order", ["shuffled", "by_cluster"])
def test_clustered_corpus_1m_x_2048_exact(order):
    from lightretriever_amd import FlatIPIndex, _lib
    from lightretriever_amd.synth import clustered_corpus, cluster_queries
    N, D, Q, k = 1_000_000, 2048, 100, 100
    idx = FlatIPIndex(D, capacity=N)
    info = clustered_corpus(idx.append_slot(N), n_clusters=1000, intra_cos=0.9, dup_frac=0.01, seed=5, order=order)
    idx.commit(N)
    q = cluster_queries(info["centres"], Q, query_cos=0.9, seed=6)
    lib = _lib.lib()
    lib.lrx_search_fallback_count(1)
    Dg, Ig = idx.search(q, k)
    hits = idx.last_list_counts()
    n_fb = lib.lrx_search_fallback_count(1)
    Dr, Ir = topk_fp64_gpu(q, idx.vectors, k)
    # the corpus really is clustered: the k-th result of every query is a member of the query's cluster (cosine ~0.9 x 0.9), and exact
    # duplicates sit next to each other in the results
    assert (Dr[:, -1] > 0.7).all()
    assert int((Dr[:, 1:] == Dr[:, :-1]).sum()) > 0
    assert torch.equal(Ig, Ir), (order, int((Ig != Ir).sum()))
    assert torch.equal(Dg, Dr)
    print("clustered 1M x 2048 (%s): hits reaching the refine step per query mean %.0f max %d; fallback queries %d of %d"
          % (order, hits.float().mean().item(), int(hits.max()), n_fb, Q))
    # k = 1000 (the reference's default top_k) on the same corpus
    D1, I1 = idx.search(q[:20], 1000)
    Dr1, Ir1 = topk_fp64_gpu(q[:20], idx.vectors, 1000)
    assert torch.equal(I1, Ir1) and torch.equal(D1, Dr1)
