"""The FUSED filter launch of the bounded search (round 5: sample pass, threshold selection and main pass in one persistent kernel without a
grid barrier, csrc/lrx_search.hip k_filter_fused) against the three-launch chain and against the definition of the result.

Replaces nothing new in the reference -- it is another schedule of the same exact search (retriever/faiss_index.py:27-40) -- so the bar is
bit-identity: ids and fp32 score bits equal to the chain's (LRX_SEARCH_FUSED_NEVER) and to the fp64 host evaluation rounded once
(helpers.flat_ip_topk_fp64), on shapes the measured rule picks by itself and on shapes forced through it (LRX_SEARCH_FUSED_ALWAYS): more
query tiles, k = 1000, D = 256 (query planes resident in LDS), row counts that are no multiple of 128, a corpus stored cluster by cluster
(the row-exact threshold branch, overflowing lists -> gated fallback), and two searches in flight on two HIP streams -- each kernel then
holds only part of the chip, the case a grid barrier would deadlock on."""
import numpy as np
import pytest
import torch

from helpers import flat_ip_topk_fp64

pytestmark = pytest.mark.gpu


def build(N, D, seed, scale="unit"):
    from lightretriever_amd import FlatIPIndex
    g = torch.Generator(device="cuda").manual_seed(seed)
    idx = FlatIPIndex(D, capacity=N)
    slot = idx.append_slot(N)
    for s in range(0, N, 65536):
        e = min(s + 65536, N)
        x = torch.randn(e - s, D, generator=g, device="cuda")
        slot[s:e] = torch.nn.functional.normalize(x, dim=-1) if scale == "unit" else x * torch.rand(e - s, 1, generator=g, device="cuda") * 3
    idx.commit(N)
    return idx, g


def both(idx, q, k):
    from lightretriever_amd import _lib
    idx.search_flags = _lib.SEARCH_FILTER_AUTO | _lib.SEARCH_FUSED_ALWAYS
    Df, If = idx.search(q, k)
    Df, If = Df.clone(), If.clone()
    hits = idx.last_list_counts().clone()
    idx.search_flags = _lib.SEARCH_FILTER_AUTO | _lib.SEARCH_FUSED_NEVER
    Dc, Ic = idx.search(q, k)
    return (Df, If), (Dc.clone(), Ic.clone()), hits


@pytest.mark.parametrize("N,D,Q,k,scale", [
    (40000, 512, 1, 10, "unit"), (40000, 512, 32, 100, "unit"), (125000, 2048, 17, 100, "unit"),      # shapes the rule itself sends down the fused launch
    (70001, 1024, 100, 100, "unit"), (30077, 768, 128, 256, "mixed"), (200000, 256, 64, 50, "unit"),   # forced: 7-8 query tiles, ragged N, resident q
    (100000, 2048, 48, 1000, "unit"), (300000, 512, 5, 1, "mixed"),                                     # forced: k = 1000; k = 1
])
def test_fused_launch_is_bit_identical_to_the_chain_and_to_the_definition(N, D, Q, k, scale):
    idx, g = build(N, D, seed=N % 97 + D)
    q = torch.randn(Q, D, generator=g, device="cuda")
    if scale == "unit":
        q = torch.nn.functional.normalize(q, dim=-1)
    q[0] = idx.vectors[N // 3]                                   # a planted exact match
    (Df, If), (Dc, Ic), hits = both(idx, q, k)
    assert torch.equal(If, Ic) and torch.equal(Df, Dc), (N, D, Q, k, int((If != Ic).sum()))
    Dw, Iw = flat_ip_topk_fp64(q.cpu().numpy(), idx.vectors.cpu().numpy(), k)
    np.testing.assert_array_equal(If.cpu().numpy(), Iw)
    np.testing.assert_array_equal(Df.cpu().numpy(), Dw)
    assert If[0, 0].item() == N // 3
    assert int(hits.min()) >= min(k, N)                          # the lists the fused launch left hold at least the k results


def test_rule_and_flags_select_the_launch():
    """statistics of the last search tell which path ran: the chain and the fused launch sample different blocks, so their candidate lists differ
    in length while the results agree; a shape outside the rule is unchanged by default"""
    from lightretriever_amd import _lib
    idx, g = build(125000, 2048, seed=5)
    q = torch.nn.functional.normalize(torch.randn(16, 2048, generator=g, device="cuda"), dim=-1)
    idx.search_flags = _lib.SEARCH_FILTER_AUTO
    Da, Ia = idx.search(q, 100)
    Da, Ia, ha = Da.clone(), Ia.clone(), idx.last_list_counts().clone()
    (Df, If), (Dc, Ic), hf = both(idx, q, 100)
    hc = idx.last_list_counts()
    assert torch.equal(Ia, If) and torch.equal(Ia, Ic) and torch.equal(Da, Df) and torch.equal(Da, Dc)
    assert torch.equal(ha, hf) and not torch.equal(hf, hc)       # 16 queries over a 125 k-row shard: the rule picks the fused launch
    q2 = torch.nn.functional.normalize(torch.randn(100, 2048, generator=g, device="cuda"), dim=-1)
    idx.search_flags = _lib.SEARCH_FILTER_AUTO
    idx.search(q2, 100)
    h_auto = idx.last_list_counts().clone()
    idx.search_flags = _lib.SEARCH_FILTER_AUTO | _lib.SEARCH_FUSED_NEVER
    idx.search(q2, 100)
    assert torch.equal(h_auto, idx.last_list_counts())           # 100 queries: the chain, as before
    with pytest.raises(_lib.LrxError):
        idx.search_flags = 12                                    # ALWAYS | NEVER
        idx.search(q2, 100)
    idx.search_flags = _lib.SEARCH_FILTER_AUTO


@pytest.mark.parametrize("order", ["by_cluster", "shuffled"])
def test_fused_launch_on_a_clustered_corpus(order):
    from lightretriever_amd import FlatIPIndex
    from lightretriever_amd.synth import clustered_corpus, cluster_queries
    N, D, Q, k = 200_000, 512, 24, 100
    idx = FlatIPIndex(D, capacity=N)
    info = clustered_corpus(idx.append_slot(N), n_clusters=400, intra_cos=0.9, dup_frac=0.01, seed=11, order=order)
    idx.commit(N)
    q = cluster_queries(info["centres"], Q, query_cos=0.9, seed=12)
    (Df, If), (Dc, Ic), hits = both(idx, q, k)
    assert torch.equal(If, Ic) and torch.equal(Df, Dc)
    Dw, Iw = flat_ip_topk_fp64(q.cpu().numpy(), idx.vectors.cpu().numpy(), k)
    np.testing.assert_array_equal(If.cpu().numpy(), Iw)
    np.testing.assert_array_equal(Df.cpu().numpy(), Dw)


def test_two_fused_searches_in_flight_do_not_wait_for_each_other():
    """pipeline.SearchLanes with the fused launch forced: two persistent kernels share the chip, each resident only in part.  Inside a launch a
    workgroup only ever waits for work that a RUNNING workgroup has claimed from a counter, so both finish; results bit-identical to one at a time."""
    from lightretriever_amd import _lib
    from lightretriever_amd.pipeline import SearchLanes
    idx, g = build(125000, 2048, seed=9)
    idx.search_flags = _lib.SEARCH_FILTER_AUTO | _lib.SEARCH_FUSED_ALWAYS
    qs = [torch.nn.functional.normalize(torch.randn(n, 2048, generator=g, device="cuda"), dim=-1) for n in (100, 7, 64, 100, 1, 33, 100, 100)]
    one = [tuple(t.clone() for t in idx.search(q, 100)) for q in qs]
    lanes = SearchLanes(idx, lanes=2)
    pend = [lanes.submit(q, 100) for q in qs]
    for (dw, iw), p in zip(one, pend):
        dg, ig = p.result()
        assert torch.equal(ig, iw) and torch.equal(dg, dw)
    lanes.drain()
    torch.cuda.synchronize()
    idx.search_flags = _lib.SEARCH_FILTER_AUTO
