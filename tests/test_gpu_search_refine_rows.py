"""The row-grouped exact rescoring of lrx_flat_ip_search_bounded (include/lrx.h, LRX_SEARCH_REFINE_ROWS_*): many queries x large k over a small
shard -- the reference's evaluation point, top-1000 of ~1000 queries per 100 k-row corpus chunk (eval/call_evaluate_mteb.sh:8-10,
retriever/faiss_index.py:27-40) -- against the per-query gather: the same bits, and both against the fp64 oracle."""
import numpy as np
import pytest
import torch

from helpers import flat_ip_topk_fp64

pytestmark = pytest.mark.gpu


def _index(X):
    from lightretriever_amd import FlatIPIndex
    idx = FlatIPIndex(X.shape[1], capacity=X.shape[0])
    idx.shadow_f16 = True
    idx.add(X)
    return idx


def _search(idx, q, k, flags):
    from lightretriever_amd import FlatIPIndex
    old = FlatIPIndex.search_flags
    FlatIPIndex.search_flags = flags
    try:
        D, I = idx.search(q, k)
        torch.cuda.synchronize()
        return D.clone(), I.clone()
    finally:
        FlatIPIndex.search_flags = old


@pytest.mark.parametrize("N,D,Q,k,filt", [
    (20000, 256, 300, 1000, 0),        # two chunks (256 + 44): GEMM main pass for the first, the 128-row kernel for the second
    (30000, 2048, 130, 500, 0),        # 16-row groups of 128 KiB
    (17000, 4096, 40, 700, 2),         # 8-row groups
    (50000, 128, 257, 1000, 0),        # narrow rows
    (20000, 192, 100, 1000, 2),        # width not a multiple of 128
    (16385, 64, 20, 2048, 2),          # the largest k
    (100000, 1024, 250, 1000, 0),      # the evaluation point's proportions: every row wanted by ~3 queries of the chunk
])
def test_row_grouped_rescoring_equals_the_per_query_gather_bitwise(N, D, Q, k, filt):
    from lightretriever_amd import _lib
    g = torch.Generator(device="cuda").manual_seed(N + D + Q)
    X = torch.randn(N, D, generator=g, device="cuda")
    X = X / X.norm(dim=1, keepdim=True)
    q = torch.randn(Q, D, generator=g, device="cuda")
    idx = _index(X)
    Dg, Ig = _search(idx, q, k, filt | _lib.SEARCH_REFINE_ROWS_NEVER)
    Dr, Ir = _search(idx, q, k, filt | _lib.SEARCH_REFINE_ROWS_ALWAYS)
    Da, Ia = _search(idx, q, k, filt)                     # the rule's own choice
    assert torch.equal(Dg, Dr) and torch.equal(Ig, Ir)
    assert torch.equal(Dg, Da) and torch.equal(Ig, Ia)
    from lightretriever_amd import _lib as L
    assert L.lib().lrx_search_fallback_count(0) >= 0
    # a few queries against the fp64 oracle (exact scores, exact order)
    sel = [0, Q // 2, Q - 1]
    wd, wi = flat_ip_topk_fp64(q[sel].cpu().numpy(), X.cpu().numpy(), k)
    np.testing.assert_array_equal(Ir[sel].cpu().numpy(), wi)
    np.testing.assert_allclose(Dr[sel].cpu().numpy(), wd, rtol=0, atol=2e-6)


def test_both_refine_flags_together_are_rejected():
    from lightretriever_amd import _lib
    X = torch.randn(20000, 64, device="cuda")
    idx = _index(X)
    with pytest.raises(Exception):
        _search(idx, torch.randn(4, 64, device="cuda"), 10, _lib.SEARCH_REFINE_ROWS_ALWAYS | _lib.SEARCH_REFINE_ROWS_NEVER)


def test_row_grouped_rescoring_with_duplicate_rows_and_overflowing_queries():
    """Near-duplicate clusters send some queries to the exact fallback (list or band overflow): those parts emit no pairs, the others are
    rescored by row group; the result is the per-query path's, bit for bit."""
    from lightretriever_amd import _lib
    g = torch.Generator(device="cuda").manual_seed(3)
    N, D, Q, k = 40000, 256, 200, 1000
    base = torch.randn(40, D, generator=g, device="cuda")
    X = base[torch.randint(0, 40, (N,), generator=g, device="cuda")] + 1e-4 * torch.randn(N, D, generator=g, device="cuda")
    X = X / X.norm(dim=1, keepdim=True)
    q = base[:Q % 40 + 10].repeat(20, 1)[:Q] + 0.01 * torch.randn(Q, D, generator=g, device="cuda")
    idx = _index(X)
    Dg, Ig = _search(idx, q, k, _lib.SEARCH_REFINE_ROWS_NEVER)
    Dr, Ir = _search(idx, q, k, _lib.SEARCH_REFINE_ROWS_ALWAYS)
    assert torch.equal(Dg, Dr) and torch.equal(Ig, Ir)
