"""GPU parity of the flat-IP index (lrx_flat_ip_search / lrx_merge_topk) against the oracle, the torch goldens and
size-independent properties at the BASELINE index size (1M x 2048 fp32)."""
import os

import numpy as np
import pytest
import torch

from oracle import lrx_oracle as O
from helpers import GOLDEN

pytestmark = pytest.mark.gpu
FLT_MAX = np.finfo(np.float32).max


def check_against_oracle(D, I, q, X, k, id_base=0, score_tol=2e-6, max_mismatch=0.01):
    """Scores within fp32 summation noise; ids identical except where the oracle's own neighbouring scores are within
    that noise (near-ties may swap); every returned (id, score) pair is self-consistent."""
    Do, Io = O.flat_ip_topk(q, X, k)
    D, I = D.cpu().numpy(), I.cpu().numpy()
    valid = Io >= 0
    np.testing.assert_allclose(D[valid], Do[valid], atol=score_tol, rtol=1e-5)
    assert (I[~valid] == -1).all() and (D[~valid] == -FLT_MAX).all()
    Iloc = I - id_base
    mism = (Iloc != Io) & valid
    if mism.any():
        qi, ri = np.nonzero(mism)
        s_ours = np.einsum("ij,ij->i", q[qi], X[Iloc[qi, ri]])
        assert np.abs(s_ours - Do[qi, ri]).max() < score_tol * 4       # a near-tie, not a wrong row
    assert mism.mean() < max_mismatch
    for r in range(D.shape[0]):                                        # sorted descending, ids unique
        dv = D[r][valid[r]]
        assert np.all(np.diff(dv) <= 0)
        assert len(set(Iloc[r][valid[r]].tolist())) == valid[r].sum()


@pytest.mark.parametrize("k", [1, 10, 100])
def test_golden_search_fixture(k):
    from lightretriever_amd import FlatIPIndex
    g = np.load(os.path.join(GOLDEN, "search.npz"))
    idx = FlatIPIndex(64)
    for s in range(0, 5000, 1300):       # add in slices like FaissIndex.build does (faiss_index.py:54-56)
        idx.add(g["X"][s:s + 1300])
    assert idx.ntotal == 5000
    D, I = idx.search(g["Q"], k)
    np.testing.assert_allclose(D.cpu().numpy(), g[f"D{k}"], atol=2e-6)
    assert (I.cpu().numpy() == g[f"I{k}"]).mean() > 0.999
    check_against_oracle(D, I, g["Q"], g["X"], k)


@pytest.mark.parametrize("N,D,Q,k", [(70000, 128, 100, 100), (1000, 32, 1, 5), (257, 64, 17, 300), (4096, 256, 200, 10), (33333, 96, 48, 1000)])
def test_random_vs_oracle(N, D, Q, k):
    from lightretriever_amd import FlatIPIndex
    rng = np.random.default_rng(N + Q)
    X = O.l2_normalize(rng.standard_normal((N, D)).astype(np.float32))
    q = O.l2_normalize(rng.standard_normal((Q, D)).astype(np.float32))
    idx = FlatIPIndex(D, capacity=N, id_base=7000)
    idx.add(X)
    Dg, Ig = idx.search(q, k)
    check_against_oracle(Dg, Ig, q, X, k, id_base=7000)


def test_ties_and_degenerate_inputs():
    from lightretriever_amd import FlatIPIndex
    rng = np.random.default_rng(4)
    base = O.l2_normalize(rng.standard_normal((700, 64)).astype(np.float32))
    X = np.concatenate([base, base, base[:100]], 0)          # exact duplicates -> ties broken by lower row id
    q = base[:9].copy()
    idx = FlatIPIndex(64)
    idx.add(X)
    D, I = idx.search(q, 7)
    Do, Io = O.flat_ip_topk(q, X, 7)
    np.testing.assert_array_equal(I.cpu().numpy()[:, :3], Io[:, :3])   # the three copies of the query row, ascending ids
    np.testing.assert_allclose(D.cpu().numpy(), Do, atol=2e-6)
    # massive tie: every row identical -> ids 0..k-1 (ordered-scan fallback path)
    idx.reset()
    idx.add(np.tile(base[:1], (6000, 1)))
    D, I = idx.search(q[:3], 10)
    np.testing.assert_array_equal(I.cpu().numpy(), np.tile(np.arange(10), (3, 1)))
    # k > ntotal and the empty index
    idx.reset()
    idx.add(base[:5])
    D, I = idx.search(q[:2], 8)
    assert (I.cpu().numpy()[:, 5:] == -1).all() and (I.cpu().numpy()[:, :5] >= 0).all()
    idx.reset()
    D, I = idx.search(q[:2], 4)
    assert (I.cpu().numpy() == -1).all()


def test_merge_topk_matches_oracle_and_single_index():
    from lightretriever_amd import FlatIPIndex, merge_topk
    rng = np.random.default_rng(8)
    N, Dm, Q, k, R = 9000, 64, 33, 50, 4
    X = O.l2_normalize(rng.standard_normal((N, Dm)).astype(np.float32))
    X[100] = X[8000]                                          # a cross-shard tie
    q = O.l2_normalize(rng.standard_normal((Q, Dm)).astype(np.float32))
    bounds = [0, 1000, 4096, 4097, N]                         # ragged shards, one with a single row (k > shard rows)
    Dp, Ip = [], []
    for r in range(R):
        sh = FlatIPIndex(Dm, id_base=bounds[r])
        sh.add(X[bounds[r]:bounds[r + 1]])
        d, i = sh.search(q, k)
        Dp.append(d), Ip.append(i)
    Dm_, Im_ = merge_topk(torch.stack(Dp), torch.stack(Ip))
    whole = FlatIPIndex(Dm)
    whole.add(X)
    Dw, Iw = whole.search(q, k)
    np.testing.assert_array_equal(Im_.cpu().numpy(), Iw.cpu().numpy())
    np.testing.assert_array_equal(Dm_.cpu().numpy(), Dw.cpu().numpy())
    Do, Io = O.merge_topk([d.cpu().numpy() for d in Dp], [i.cpu().numpy() for i in Ip], k)
    np.testing.assert_array_equal(Im_.cpu().numpy(), Io)


def test_merge_at_the_widest_supported_exchange():
    """8 shards x k = 2048 (R * k = 16384, the limit; ADVICE r1: k > 1024 over 8 ranks used to be refused) -- both forms of the merge,
    and one entry more is refused loudly."""
    from lightretriever_amd import FlatIPIndex, merge_topk, _lib
    rng = np.random.default_rng(12)
    N, Dm, Q, k, R = 40000, 32, 3, 2048, 8
    X = O.l2_normalize(rng.standard_normal((N, Dm)).astype(np.float32))
    q = O.l2_normalize(rng.standard_normal((Q, Dm)).astype(np.float32))
    Dp, Ip = [], []
    for r in range(R):
        sh = FlatIPIndex(Dm, id_base=r * N // R)
        sh.add(X[r * N // R:(r + 1) * N // R])
        d, i = sh.search(q, k)
        Dp.append(d), Ip.append(i)
    Dm_, Im_ = merge_topk(torch.stack(Dp), torch.stack(Ip))
    Do, Io = O.flat_ip_topk(q, X, k)
    np.testing.assert_array_equal(Im_.cpu().numpy(), Io)
    lib = _lib.lib()
    words = torch.empty(R, Q, k, dtype=torch.int64, device="cuda")
    for r in range(R):
        _lib.check(lib.lrx_pack_topk(_lib.ptr(Dp[r]), _lib.ptr(Ip[r]), None, 0, Q * k, _lib.ptr(words[r]), _lib.current_stream()))
    D2, I2 = torch.empty_like(Dm_), torch.empty_like(Im_)
    _lib.check(lib.lrx_merge_topk_packed(_lib.ptr(words), R, Q, k, _lib.ptr(D2), _lib.ptr(I2), _lib.current_stream()))
    assert torch.equal(D2, Dm_) and torch.equal(I2, Im_)
    with pytest.raises(_lib.LrxError):
        merge_topk(torch.zeros(9, 1, 2048, device="cuda"), torch.zeros(9, 1, 2048, dtype=torch.int64, device="cuda"))


def test_full_size_1m_x_2048_properties():
    """BASELINE config 2 index size.  Properties: sorted, unique ids, every score equals the recomputed dot product, the
    top-1 equals a chunked torch argmax, planted exact matches are found at rank 0, shard-merge equals whole-index."""
    from lightretriever_amd import FlatIPIndex, merge_topk
    N, D, Q, k = 1_000_000, 2048, 100, 100
    g = torch.Generator(device="cuda").manual_seed(7)
    idx = FlatIPIndex(D, capacity=N)
    slot = idx.append_slot(N)
    for s in range(0, N, 100_000):
        blk = torch.randn(100_000, D, generator=g, device="cuda")
        slot[s:s + 100_000] = torch.nn.functional.normalize(blk, dim=-1)
    idx.commit(N)
    q = torch.nn.functional.normalize(torch.randn(Q, D, generator=g, device="cuda"), dim=-1)
    planted = torch.randint(0, N, (10,), generator=g, device="cuda")
    q[:10] = idx.vectors[planted]
    Dg, Ig = idx.search(q, k)
    assert (Ig[:10, 0] == planted).all() and torch.allclose(Dg[:10, 0], torch.ones(10, device="cuda"), atol=1e-5)
    assert (Dg[:, 1:] <= Dg[:, :-1]).all()
    assert all(len(set(r.tolist())) == k for r in Ig.cpu())
    rec = torch.einsum("qkd,qd->qk", idx.vectors[Ig.reshape(-1)].view(Q, k, D), q)
    assert torch.allclose(rec, Dg, atol=3e-6)
    best = torch.full((Q,), -2.0, device="cuda")
    arg = torch.zeros(Q, dtype=torch.int64, device="cuda")
    kth = torch.full((Q,), 0.0, device="cuda")
    for s in range(0, N, 250_000):
        sc = q @ idx.vectors[s:s + 250_000].T
        m, a = sc.max(dim=1)
        upd = m > best
        best, arg = torch.where(upd, m, best), torch.where(upd, a + s, arg)
        kth = kth + (sc > Dg[:, -1:] + 3e-6).sum(1)
    assert torch.allclose(best, Dg[:, 0], atol=3e-6)
    assert ((arg == Ig[:, 0]) | ((best - Dg[:, 0]).abs() < 3e-6)).all()
    assert (kth <= k - 1).all()          # nothing clearly better than the k-th result was missed
    # two-shard merge == whole index
    cut = 600_064                                         # a multiple of 128: the tiled shadow is shared in whole 128-row blocks
    a, b = FlatIPIndex(D, id_base=0), FlatIPIndex(D, id_base=cut)
    a._x, a._xb, a._bounds, a.ntotal = idx._x[:cut], idx._xb[:cut * D], idx._bounds, cut               # views of the same rows, shadow and bounds
    b._x, b._xb, b._bounds, b.ntotal = idx._x[cut:], idx._xb[cut * D:], idx._bounds, N - cut
    Da, Ia = a.search(q, k)
    Db, Ib = b.search(q, k)
    Dm, Im = merge_topk(torch.stack([Da, Db]), torch.stack([Ia, Ib]))
    assert torch.equal(Im, Ig) and torch.equal(Dm, Dg)


# ---- two-pass bounded search (bf16 filter + exact rescoring), Q > 32 ---------------------------------------------------
@pytest.mark.parametrize("N,D,Q,k,scale", [(120000, 256, 100, 100, "unit"), (50000, 128, 64, 10, "mixed"), (9000, 64, 33, 1000, "unit"),
                                            (30000, 2048, 40, 50, "unit"), (20000, 128, 300, 7, "unit"), (6000, 64, 5, 2048, "mixed"),
                                            # more than 128 queries: the shadow filter takes up to 256 per pass (9..16 MFMA query tiles)
                                            (25000, 64, 145, 20, "unit"), (25000, 320, 255, 33, "mixed"), (15000, 128, 513, 5, "unit"),
                                            (12000, 192, 177, 100, "unit")])
def test_two_pass_equals_six_product_path_and_oracle(N, D, Q, k, scale):
    from lightretriever_amd import FlatIPIndex
    rng = np.random.default_rng(N + D)
    X = O.l2_normalize(rng.standard_normal((N, D)).astype(np.float32))
    if scale == "mixed":                                   # rows of very different norms: the band scales with the largest one
        X *= rng.uniform(0.05, 3.0, size=(N, 1)).astype(np.float32)
    q = O.l2_normalize(rng.standard_normal((Q, D)).astype(np.float32)) * np.float32(1.7)
    idx = FlatIPIndex(D, capacity=N)
    idx.add(X[:N // 2])
    idx.add(X[N // 2:])
    assert abs(float(idx._norm_bound) / float(np.linalg.norm(X, axis=1).max()) - 1) < 1e-5
    D2, I2 = idx.search(q, k)
    idx.two_pass = False
    D6, I6 = idx.search(q, k)
    check_against_oracle(D2, I2, q, X, k)
    same = (I2 == I6).float().mean().item()
    assert same > 0.999                                    # only fp32-noise near-ties may swap between the two exact paths
    np.testing.assert_allclose(D2.cpu().numpy(), D6.cpu().numpy(), atol=3e-6, rtol=1e-5)


def test_two_pass_band_overflow_falls_back_per_query():
    """A corpus of near-duplicates puts far more rows inside the filter band than the on-chip candidate list holds: those queries
    must come back from the gated six-product fallback, the others from the refine kernel -- all exact."""
    from lightretriever_amd import FlatIPIndex
    rng = np.random.default_rng(8)
    N, D, Q, k = 60000, 128, 48, 20
    X = O.l2_normalize(rng.standard_normal((N, D)).astype(np.float32))
    base = X[123].copy()
    X[10000:30000] = O.l2_normalize(base[None, :] + 1e-4 * rng.standard_normal((20000, D)).astype(np.float32))   # 20k near-copies
    X[40000:40010] = base                                                                                         # and exact ties
    q = O.l2_normalize(rng.standard_normal((Q, D)).astype(np.float32))
    q[:5] = O.l2_normalize(base[None, :] + 0.01 * rng.standard_normal((5, D)).astype(np.float32))                 # these hit the cluster
    idx = FlatIPIndex(D, capacity=N)
    idx.add(X)
    Dg, Ig = idx.search(q, k)
    # the 5 cluster queries see 20k rows within ~1e-6 of each other: the oracle's own sgemm noise reorders those (near-ties are
    # verified as such inside the check); everything else must match exactly
    check_against_oracle(Dg, Ig, q, X, k, score_tol=3e-6, max_mismatch=0.12)
    Do, Io = O.flat_ip_topk(q[5:], X, k)
    assert (Ig[5:].cpu().numpy() == Io).mean() > 0.999
    # exact duplicates come back lowest row id first on both paths
    idx2 = FlatIPIndex(D, capacity=N)
    Xd = np.repeat(X[:50], 200, axis=0)                  # 10 000 rows, every vector 200 times
    idx2.add(Xd)
    Dd, Id = idx2.search(q, 8)
    Dd_o, Id_o = O.flat_ip_topk(q, Xd, 8)
    np.testing.assert_array_equal(Id.cpu().numpy(), Id_o)


def test_two_pass_adversarial_bf16_rounding():
    """Rows built so that the bf16 filter ranks them in the wrong order: the rescoring must restore the exact order."""
    from lightretriever_amd import FlatIPIndex
    rng = np.random.default_rng(2)
    N, D, Q, k = 20000, 64, 40, 5
    X = O.l2_normalize(rng.standard_normal((N, D)).astype(np.float32)) * np.float32(0.5)
    q = O.l2_normalize(rng.standard_normal((Q, D)).astype(np.float32))
    # planted near-winners: q_j scaled by values just below / above a bf16 rounding boundary, so bf16(x) over- or under-states them
    for j in range(Q):
        for t, f in enumerate([0.99805, 0.99902, 1.0, 1.00195, 1.0039]):       # around 1.0 the bf16 grid is 2^-8 / 2^-7 wide
            X[j * 7 + t] = q[j] * np.float32(f * 0.97)
    idx = FlatIPIndex(D, capacity=N)
    idx.add(X)
    Dg, Ig = idx.search(q, k)
    Do, Io = O.flat_ip_topk(q, X, k)
    np.testing.assert_array_equal(Ig.cpu().numpy(), Io)
    np.testing.assert_allclose(Dg.cpu().numpy(), Do, atol=2e-6)


@pytest.mark.parametrize("Q", [1, 20, 100])
def test_fp16_shadow_filter_gives_the_same_exact_result(Q):
    """The filter pass may stream the fp16 shadow of the rows instead of the fp32 rows (half the bytes): same error band, same exact
    rescoring from fp32 -> bitwise the same scores and ids as without the shadow, for every query batch size."""
    from lightretriever_amd import FlatIPIndex
    rng = np.random.default_rng(40 + Q)
    N, D, k = 60000, 256, 64
    X = O.l2_normalize(rng.standard_normal((N, D)).astype(np.float32)) * rng.uniform(0.5, 1.5, size=(N, 1)).astype(np.float32)
    q = O.l2_normalize(rng.standard_normal((Q, D)).astype(np.float32))
    a = FlatIPIndex(D, capacity=N)
    a.add(X[:25000]); a.add(X[25000:])
    assert a._xb is not None and a._xb.dtype == torch.float16 and torch.equal(a.shadow_rows(), torch.from_numpy(X).cuda().to(torch.float16))
    Da, Ia = a.search(q, k)
    b = FlatIPIndex(D, capacity=N)
    b.shadow_f16 = False
    b.add(X)
    assert b._xb is None
    Db, Ib = b.search(q, k)
    assert torch.equal(Ia, Ib) and torch.equal(Da, Db)
    check_against_oracle(Da, Ia, q, X, k)
    # rows rewritten in place after commit: refresh rebuilds bound and shadow
    a._x[:10] = torch.from_numpy(X[100:110]).cuda() * 3.0
    a.refresh_norm_bound()
    X2 = X.copy(); X2[:10] = X[100:110] * 3.0
    D2, I2 = a.search(q, k)
    check_against_oracle(D2, I2, q, X2, k)


def test_query_chunking_under_a_workspace_cap_changes_nothing():
    """FlatIPIndex.search keeps the [queries, rows] score workspace under max_workspace_bytes by chunking the queries: same hits."""
    from lightretriever_amd import FlatIPIndex
    rng = np.random.default_rng(11)
    N, D, Q, k = 30000, 128, 700, 10
    X = O.l2_normalize(rng.standard_normal((N, D)).astype(np.float32))
    q = O.l2_normalize(rng.standard_normal((Q, D)).astype(np.float32))
    idx = FlatIPIndex(D, capacity=N)
    idx.add(X)
    D0, I0 = idx.search(q, k)
    ws_full = idx._ws.numel()
    for cap in (ws_full // 2, ws_full // 5, 40 * N * 4):          # -> chunks of 256, 128, 40-ish queries
        idx._ws = None
        idx.max_workspace_bytes = cap
        D1, I1 = idx.search(q, k)
        assert idx._ws.numel() < ws_full
        assert torch.equal(D0, D1) and torch.equal(I0, I1)
    check_against_oracle(D0, I0, q, X, k)


@pytest.mark.parametrize("R,k", [(1, 7), (2, 30), (2, 64), (8, 100), (3, 333), (8, 256), (2, 2048), (7, 1000), (8, 2048), (8, 2047), (4, 1023), (3, 100), (5, 1)])
def test_merge_topk_every_sort_width_against_a_host_sort(R, k):
    """lrx_merge_topk / lrx_merge_topk_packed over the whole range of R x k (round 4: the sort behind them runs in registers, 2 .. 16 entries
    per thread, for 128 .. 16384 padded entries; the LDS form below that): random scores with exact ties across parts, -1 padding, against a
    host sort by (score descending, id ascending)."""
    from lightretriever_amd import merge_topk, _lib
    from lightretriever_amd.sharded import pack_pairs
    g = torch.Generator().manual_seed(R * 10007 + k)
    Q = 13
    D = torch.randint(0, 50, (R, Q, k), generator=g).float() / 7.0              # many exact ties
    I = torch.stack([torch.randperm(R * k * 3, generator=g)[:R * k].view(R, k) for _ in range(Q)], 1).to(torch.int64)   # distinct ids per query
    pad = torch.rand(R, Q, k, generator=g) < 0.1
    D[pad], I[pad] = -3.4028234663852886e38, -1
    if (R + k) % 2 == 0:
        # lists in order (what searches return; the merge then ranks by binary search instead of sorting): sort every list by (score desc, id asc),
        # padding last
        key = torch.where(I >= 0, D.double() * 1e9 - I.double() * 1e-3, torch.full_like(D, -1e30, dtype=torch.float64))
        order = key.argsort(dim=-1, descending=True)
        D, I = torch.gather(D, -1, order), torch.gather(I, -1, order)
    Dm, Im = merge_topk(D.cuda(), I.cuda())
    words = pack_pairs(D, I).cuda()
    D2 = torch.empty(Q, k, device="cuda")
    I2 = torch.empty(Q, k, dtype=torch.int64, device="cuda")
    _lib.check(_lib.lib().lrx_merge_topk_packed(_lib.ptr(words), R, Q, k, _lib.ptr(D2), _lib.ptr(I2), _lib.current_stream()))
    for q in range(Q):
        ent = [(float(D[r, q, j]), int(I[r, q, j])) for r in range(R) for j in range(k) if int(I[r, q, j]) >= 0]
        ent.sort(key=lambda e: (-e[0], e[1]))
        want = ent[:k] + [(-3.4028234663852886e38, -1)] * max(0, k - len(ent))
        for got_d, got_i in ((Dm, Im), (D2, I2)):
            assert got_i[q].tolist() == [e[1] for e in want], (R, k, q)
            assert got_d[q].tolist() == [e[0] for e in want]
