"""GPU parity of the score-free (candidate-list) filter of lrx_flat_ip_search_bounded, the rigorous per-query error band and the
shard maintenance fused into the row producers -- against the oracle, against the score-matrix filter (bitwise) and, at the index
shapes of BASELINE configs 2-5, through size-independent properties."""
import numpy as np
import pytest
import torch

from oracle import lrx_oracle as O
from helpers import flat_ip_topk_fp64
from test_gpu_search import check_against_oracle

pytestmark = pytest.mark.gpu


@pytest.fixture
def search_mode():
    """Selects the filter of lrx_flat_ip_search_bounded (its `flags` argument, include/lrx.h LRX_SEARCH_FILTER_*) for every FlatIPIndex.search of
    the test: 0 auto, 1 score-matrix filter, 2 score-free filter, 3 score-free without the GEMM main pass."""
    from lightretriever_amd import FlatIPIndex

    def set_mode(m):
        FlatIPIndex.search_flags = int(m)
    yield set_mode
    FlatIPIndex.search_flags = 0


def _index(X, shadow=True, id_base=0, pieces=2):
    from lightretriever_amd import FlatIPIndex
    idx = FlatIPIndex(X.shape[1], capacity=X.shape[0], id_base=id_base)
    idx.shadow_f16 = shadow
    step = -(-X.shape[0] // pieces)
    for s in range(0, X.shape[0], step):
        idx.add(X[s:s + step])
    return idx


@pytest.mark.parametrize("N,D,Q,k,scale,shadow", [
    (120000, 256, 100, 100, "unit", True), (50000, 128, 64, 10, "mixed", True), (40000, 64, 33, 1000, "unit", True),
    (30000, 2048, 40, 50, "unit", True), (70000, 128, 300, 7, "unit", True), (25000, 64, 5, 2048, "mixed", True),
    (25000, 64, 145, 20, "unit", True), (25000, 320, 255, 33, "mixed", True), (20000, 128, 513, 5, "unit", True),
    (33000, 4096, 48, 100, "unit", True),                       # the 8B width (BASELINE configs 2-3)
    # 129..256 queries over a wide shard: the main pass runs on the GEMM kernel (256-row tiles, sample in 256-row units)
    (40000, 1024, 200, 10, "unit", True), (33001, 2048, 300, 5, "mixed", True), (70000, 1024, 256, 100, "unit", True), (20000, 1088, 129, 3, "unit", True),
    # round 6 -- WIDE chunks: more than 256 queries over a shadow of D >= 1024 are ONE pass of the GEMM kernel per up to 1024 queries (2-4 query
    # n-tiles per streamed A tile; the sample pass, the fallback planes and the group flags per 256 / 128 queries inside): 600 = 3 n-tiles with a
    # ragged last one, 1000 / 1024 = 4, 1100 = two balanced chunks of 560 + 540, the 8B width, k = 1000 (row-grouped rescoring over the chunk)
    (70000, 1024, 600, 100, "unit", True), (40000, 2048, 1000, 10, "mixed", True), (50000, 1024, 1100, 20, "unit", True),
    (33000, 4096, 513, 100, "unit", True), (30000, 1024, 1024, 1000, "unit", True), (20480, 1536, 257, 64, "mixed", True),
    (60000, 256, 100, 100, "unit", False), (45000, 128, 128, 10, "mixed", False), (30011, 96, 40, 64, "unit", False),   # fp32 rows converted on the fly
    # the register-streaming filter kernels of the tiled shadow: persistent main pass (D/64 % 4 == 0: 512, 768; whole q resident in LDS at
    # 256), one-workgroup-per-block form otherwise (192 = 3 slices), odd block counts for the two-blocks-at-a-time walk, 9..16 query tiles
    (50001, 512, 100, 20, "unit", True), (30000, 768, 17, 5, "mixed", True), (40000, 192, 60, 10, "unit", True), (90000, 256, 200, 10, "unit", True),
    (16512, 256, 2, 3, "unit", True), (16640, 1024, 128, 100, "mixed", True),
    # many hits per wave without any list overflowing (256 queries x ~3000 rows above the threshold over ~600 waves): the per-wave LDS hit
    # lists of the persistent pass fill up and are flushed in mid-pass
    (20000, 256, 256, 1000, "unit", True), (24000, 512, 128, 2000, "mixed", True),
    # more than two sample block pairs per CU: the sample pass runs as persistent workgroups (k_filter_xreg_store), with the whole q
    # resident in LDS (D = 256) and with a cycling q ring (D = 512), two blocks at a time and one (seven query tiles)
    (300000, 256, 20, 1000, "unit", True), (300001, 512, 100, 1000, "mixed", True),
    (120000, 256, 30, 1, "unit", True), (200000, 128, 9, 3, "mixed", True),      # samples of >= 8 k blocks: threshold from the block maxima
])
def test_score_free_filter_equals_score_matrix_filter_bitwise(N, D, Q, k, scale, shadow, search_mode):
    """Both filters end in the same exact rescoring of a superset of the exact top-k: same ids, same score bits."""
    rng = np.random.default_rng(N + D + Q)
    X = O.l2_normalize(rng.standard_normal((N, D)).astype(np.float32))
    if scale == "mixed":
        X *= rng.uniform(0.05, 3.0, size=(N, 1)).astype(np.float32)
    q = O.l2_normalize(rng.standard_normal((Q, D)).astype(np.float32)) * np.float32(1.7)
    idx = _index(X, shadow=shadow, id_base=1000)
    search_mode(2)
    De, Ie = idx.search(q, k)
    search_mode(1)
    Dm, Im = idx.search(q, k)
    search_mode(3)                                                # score-free, main pass never on the GEMM kernel
    D3, I3 = idx.search(q, k)
    search_mode(0)
    Da, Ia = idx.search(q, k)
    assert torch.equal(Ie, Im) and torch.equal(De, Dm)
    assert torch.equal(I3, Im) and torch.equal(D3, Dm)
    assert torch.equal(Ia, Im) and torch.equal(Da, Dm)
    check_against_oracle(De, Ie, q, X, k, id_base=1000)


def test_tiled_shadow_layout_is_the_documented_one():
    """include/lrx.h: element k of row r of the tiled shadow sits at
    ((r/128)(D/64) + k/64) 8192 + ((((r/16)%8) 2 + (k/32)%2) 64 + ((k/8)%4) 16 + r%16) 8 + k%8."""
    rng = np.random.default_rng(9)
    N, D = 1000, 192
    X = rng.standard_normal((N, D)).astype(np.float32)
    idx = _index(X, pieces=3)
    assert idx._xb.ndim == 1 and idx._xb.dtype == torch.float16
    flat = idx._xb.view(torch.int16).cpu().numpy()
    want = torch.from_numpy(X).to(torch.float16).view(torch.int16).numpy()
    r = rng.integers(0, N, size=4000)
    k = rng.integers(0, D, size=4000)
    off = ((r // 128) * (D // 64) + k // 64) * 8192 + ((((r // 16) % 8) * 2 + (k // 32) % 2) * 64 + ((k // 8) % 4) * 16 + r % 16) * 8 + k % 8
    assert np.array_equal(flat[off], want[r, k])


def test_workspace_stops_growing_at_256_queries():
    from lightretriever_amd import _lib
    lib = _lib.lib()
    w = [int(lib.lrx_flat_ip_bounded_workspace_bytes(10_000_000, 256, nq, 100, 0)) for nq in (100, 256, 1000, 7000)]
    assert w[1] == w[2] == w[3] and w[0] <= w[1]
    assert w[3] < 6 << 30                                        # round 1: 280 GB of [queries, rows] scores for 7000 queries
    full = 10_000_000 * 100 * 4
    assert int(lib.lrx_flat_ip_bounded_workspace_bytes(10_000_000, 256, 100, 100, 0)) < 1.1 * full + (64 << 20)
    # top_k = 1000 (the reference's default): lists of 64 Ki entries, 128 MB for 256 queries.  D >= 1024 (round 6): a call of more than 256
    # queries is walked in WIDE chunks of up to 1024 (one GEMM pass over the shadow each): the workspace stops growing there
    w1k = [int(lib.lrx_flat_ip_bounded_workspace_bytes(1_000_000, 2048, nq, 1000, 0)) for nq in (256, 1024, 5000, 50000)]
    assert w1k[0] < 1 << 30 and w1k[0] < w1k[1] and w1k[2] <= w1k[1] and w1k[3] <= w1k[1] < 2 << 30
    ch = lambda n, d=2048, rows=1_000_000: int(lib.lrx_flat_ip_bounded_chunk_queries(rows, d, n, 100, 0, 1))
    assert [ch(n) for n in (1, 100, 256, 257, 1000, 1024, 1025, 3000)] == [256, 256, 256, 272, 1008, 1024, 528, 1008]
    assert ch(1000, d=256) == 256 and ch(1000, rows=3000) == 256 and int(lib.lrx_flat_ip_bounded_chunk_queries(1_000_000, 2048, 1000, 100, 0, 0)) == 128
    assert int(lib.lrx_flat_ip_bounded_chunk_queries(1_000_000, 2048, 1000, 100, 1, 1)) == 256     # (score-matrix filter forced: no wide chunks)


def test_adversarial_fp16_rounding_midpoints(search_mode):
    """ADVICE r1, moved to the fp16 filter of round 3: fp16 has unit roundoff 2^-11.  Rows whose elements all sit just below a rounding
    midpoint lose 0.05 % of their score in the filter; rows with half their elements just above one gain as much and outrank them there
    although they are worse.  The band is built from the measured |q - fp16(q)| and |x - fp16(x)|, so the exact top-k must survive on both
    filters and equal the six-product path."""
    rng = np.random.default_rng(5)
    N, D, Q, k = 40000, 128, 48, 10
    X = O.l2_normalize(rng.standard_normal((N, D)).astype(np.float32)) * np.float32(0.9)
    w = np.float32(1 + 2.0 ** -11 - 2.0 ** -15)         # rounds DOWN to 1.0            (exact 1.000458)
    a = np.float32(1 + 2.0 ** -11 + 2.0 ** -15)         # rounds UP   to 1 + 2^-10      (exact 1.000519, filter 1.000977)
    b = np.float32(1 - 2.0 ** -12 + 2.0 ** -15)         # rounds UP   to 1.0            (exact 0.999786)
    q = np.abs(O.l2_normalize(rng.standard_normal((Q, D)).astype(np.float32)))
    q[:Q // 2] = w * np.float32(2.0) ** rng.integers(-3, 1, size=(Q // 2, D)).astype(np.float32)   # queries made of understated elements too
    s = np.float32(0.9 / np.sqrt(D))
    half = np.arange(D) % 2 == 0
    for j in range(Q):
        # block j = the pattern scaled by (1 - j / 1000): block 0 holds every query's exact top-10 (all q > 0, rows parallel to 1)
        for t in range(12):                               # the true winners: all elements understated, filter score 2.0 (in units of s)
            X[j * 40 + t] = s * w * np.float32(1 - 1.2e-5 * t) * np.float32(1 - 1e-3 * j)
        for t in range(12, 30):                           # decoys: exact mean 1.000153 < 1.000458, filter 1.000488 > 1.0
            X[j * 40 + t] = s * np.where(half, a, b).astype(np.float32) * np.float32(1 - 6e-7 * t) * np.float32(1 - 1e-3 * j)
    Do, Io = O.flat_ip_topk(q, X, k)
    Xh = torch.from_numpy(X).to(torch.float16).float().numpy()
    qh = torch.from_numpy(q).to(torch.float16).float().numpy()
    h_rank = np.argsort(-(qh.astype(np.float64) @ Xh.astype(np.float64).T), axis=1)[:, :k]
    assert (np.sort(h_rank, 1) != np.sort(Io, 1)).any(axis=1).mean() > 0.7      # an fp16-only ranking gets most queries wrong
    idx = _index(X)
    res = {}
    for mode in (1, 2):
        search_mode(mode)
        res[mode] = idx.search(q, k)
    search_mode(0)
    idx.two_pass = False
    D6, I6 = idx.search(q, k)
    for mode in (1, 2):
        Dg, Ig = res[mode]
        np.testing.assert_allclose(Dg.cpu().numpy(), Do, atol=3e-6, rtol=2e-6)
        np.testing.assert_array_equal(Ig.cpu().numpy(), Io)
        assert torch.equal(Ig, I6)
        assert (Ig.cpu().numpy() < 12).all()                 # block 0's understated winners, not the overstated decoys


def test_rows_outside_the_fp16_range_fall_back_to_the_exact_path(search_mode):
    """The shadow saturates at +-65504: a row with larger elements shows up as a huge measured rounding error E -> an unusable band -> every
    query takes the exact six-product fallback.  Slow, never wrong -- and no inf / NaN anywhere."""
    rng = np.random.default_rng(6)
    N, D, Q, k = 30000, 128, 20, 10
    X = O.l2_normalize(rng.standard_normal((N, D)).astype(np.float32))
    X[777] *= np.float32(4e6)                                # elements ~ 3e5 > 65504
    X[12345, 5] = np.float32(1e5)
    q = O.l2_normalize(rng.standard_normal((Q, D)).astype(np.float32))
    idx = _index(X)
    assert torch.isfinite(idx.shadow_rows().float()).all() and float(idx._bounds[1]) > 1e4
    for mode in (2, 1):
        search_mode(mode)
        Dg, Ig = idx.search(q, k)
        assert torch.isfinite(Dg).all()
        check_against_oracle(Dg, Ig, q, X, k, score_tol=0.5)          # (scores ~ 1e5: fp32 resolution of the exact value)
        np.testing.assert_array_equal(Ig.cpu().numpy(), O.flat_ip_topk(q, X, k)[1])


def test_candidate_list_overflow_falls_back_per_query(search_mode):
    """More rows inside T' - 2 eps than a candidate list holds (near-duplicate corpus): those queries must come back from the gated
    six-product fallback, everything exact."""
    rng = np.random.default_rng(8)
    N, D, Q, k = 90000, 128, 48, 20
    X = O.l2_normalize(rng.standard_normal((N, D)).astype(np.float32))
    base = X[123].copy()
    X[10000:40000] = O.l2_normalize(base[None, :] + 1e-4 * rng.standard_normal((30000, D)).astype(np.float32))   # 30k near-copies > CAND_CAP
    X[50000:50010] = base
    q = O.l2_normalize(rng.standard_normal((Q, D)).astype(np.float32))
    q[:5] = O.l2_normalize(base[None, :] + 0.01 * rng.standard_normal((5, D)).astype(np.float32))
    idx = _index(X)
    search_mode(2)
    Dg, Ig = idx.search(q, k)
    check_against_oracle(Dg, Ig, q, X, k, score_tol=3e-6, max_mismatch=0.12)    # (the oracle's own fp32 sgemm cannot order the 30k near-copies)
    D64, I64 = flat_ip_topk_fp64(q, X, k)                                       # ... the fp64 reference can: exact ids and score bits, all queries
    np.testing.assert_array_equal(Ig.cpu().numpy(), I64)
    np.testing.assert_array_equal(Dg.cpu().numpy(), D64)
    Xd = np.repeat(X[:50], 400, axis=0)                          # 20 000 rows, every vector 400 times: exact ties, lowest row first
    idx2 = _index(Xd)
    Dd, Id = idx2.search(q, 8)
    _, Id_o = O.flat_ip_topk(q, Xd, 8)
    np.testing.assert_array_equal(Id.cpu().numpy(), Id_o)
    # more than 128 flagged queries: two groups of the gated fallback, each with its own pre-split query planes (written by the
    # packing kernel at the head of the chain), the second one short enough for the exact-fp32 kernel (no planes)
    q2 = O.l2_normalize(rng.standard_normal((150, D)).astype(np.float32))
    _, I2 = idx2.search(q2, 8)
    _, I2_o = O.flat_ip_topk(q2, Xd, 8)
    np.testing.assert_array_equal(I2.cpu().numpy(), I2_o)
    # a shard with more 128-row blocks than the grid cap of the gated six-product launch (8 x CUs): its workgroups walk several blocks
    Xw = np.repeat(X[:50], 6000, axis=0)                         # 300 000 rows = 2344 blocks, every vector 6000 times
    idxw = _index(Xw, pieces=3)
    _, Iw = idxw.search(q, 8)
    _, Iw_o = O.flat_ip_topk(q, Xw, 8)
    np.testing.assert_array_equal(Iw.cpu().numpy(), Iw_o)
    q3 = O.l2_normalize(rng.standard_normal((230, D)).astype(np.float32))       # both groups on the six-product kernel
    _, I3 = idx2.search(q3, 8)
    _, I3_o = O.flat_ip_topk(q3, Xd, 8)
    np.testing.assert_array_equal(I3.cpu().numpy(), I3_o)


def test_searches_in_flight_on_two_streams_give_the_same_bits():
    """Round 4: pipeline.SearchLanes keeps two searches in flight on two HIP streams (own workspaces); a search of more than 256 queries
    alternates its 256-query chunks over two internal streams inside the call.  Same bits as one search after the other, also with a
    row-map exchange tail, also when the caller consumes results late."""
    from lightretriever_amd import FlatIPIndex
    from lightretriever_amd.pipeline import SearchLanes
    from lightretriever_amd.sharded import ShardedFlatIPIndex
    rng = np.random.default_rng(31)
    N, D, k = 70000, 256, 50
    X = O.l2_normalize(rng.standard_normal((N, D)).astype(np.float32))
    idx = _index(X)
    qs = [torch.from_numpy(O.l2_normalize(rng.standard_normal((n, D)).astype(np.float32))).cuda() for n in (100, 37, 256, 100, 700, 1)]
    idx.chunk_lanes = 1
    want = [idx.search(q, k) for q in qs]                               # one stream, no internal fork
    idx.chunk_lanes = 2
    for q, (Dw, Iw) in zip(qs, want):                                   # the 700-query search forks its three chunks
        Dg, Ig = idx.search(q, k)
        assert torch.equal(Dg, Dw) and torch.equal(Ig, Iw)
    lanes = SearchLanes(idx, lanes=2)
    pend = [lanes.submit(q, k) for q in qs * 3]
    for i, h in enumerate(pend):
        Dg, Ig = h.result()
        assert torch.equal(Dg, want[i % len(qs)][0]) and torch.equal(Ig, want[i % len(qs)][1])
    # the query producer on the lane's stream, results consumed one submit late (the serving-loop pattern)
    table = torch.randn(500, D, device="cuda")
    from lightretriever_amd import ops
    ids = torch.randint(0, 500, (40 * 9,), device="cuda")
    offs = torch.arange(0, 40 * 9, 9, device="cuda")
    qe = ops.embedding_bag_mean(table, ids, offs, normalize=True)
    De, Ie = idx.search(qe, k)
    prev = None
    for _ in range(6):
        h = lanes.submit(lambda: ops.embedding_bag_mean(table, ids, offs, normalize=True), k)
        if prev is not None:
            Dg, Ig = prev.result()
            assert torch.equal(Dg, De) and torch.equal(Ig, Ie)
        prev = h
    lanes.drain()
    row_map = torch.from_numpy(rng.permutation(N).astype(np.int64)).cuda()
    sh = ShardedFlatIPIndex(idx, row_map=row_map)
    Ds, Is = sh.search(qs[0], k)
    lanes_s = SearchLanes(sh, lanes=3)
    for h in [lanes_s.submit(qs[0], k) for _ in range(5)]:
        Dg, Ig = h.result()
        assert torch.equal(Dg, Ds) and torch.equal(Ig, Is)
    torch.cuda.synchronize()


def test_wire_words_written_by_the_search_itself_equal_the_packing_kernel(search_mode):
    """Round 4: a row-sharded search lets the chain's last kernel write the 64-bit exchange words (lrx_flat_ip_search_bounded_wire), with or
    without a row map, for results that came from the refine step AND for results the gated fallback rewrote; equal to lrx_pack_topk."""
    from lightretriever_amd import _lib
    from lightretriever_amd.sharded import ShardedFlatIPIndex, pack_pairs, unpack_pairs
    rng = np.random.default_rng(21)
    N, D, k = 90000, 128, 20
    X = O.l2_normalize(rng.standard_normal((N, D)).astype(np.float32))
    base = X[123].copy()
    X[10000:40000] = O.l2_normalize(base[None, :] + 1e-4 * rng.standard_normal((30000, D)).astype(np.float32))   # near-copies: list overflow
    q = O.l2_normalize(rng.standard_normal((300, D)).astype(np.float32))                                         # two chunks of queries
    q[:5] = O.l2_normalize(base[None, :] + 0.01 * rng.standard_normal((5, D)).astype(np.float32))                # ... five of them take the fallback
    q[270:273] = q[:3]
    qd = torch.from_numpy(q).cuda()
    idx = _index(X, id_base=1000)
    row_map = torch.from_numpy(rng.permutation(N).astype(np.int64) * 3 + 7).cuda()
    lib = _lib.lib()
    for mode in (0, 2, 1):
        search_mode(mode)
        lib.lrx_search_fallback_count(1)
        D0, I0 = idx.search(qd, k)
        n_fb = lib.lrx_search_fallback_count(1)
        assert n_fb >= 8 if mode != 1 else n_fb >= 0, n_fb                     # (the flagged queries of both chunks were counted)
        for rm in (None, row_map):
            words = torch.full((300, k), -7, dtype=torch.int64, device="cuda")
            D1, I1 = idx.search(qd, k, wire_out=words, row_map=rm)
            assert torch.equal(D1, D0) and torch.equal(I1, I0)
            want_ids = I0 if rm is None else rm[I0 - 1000]
            assert torch.equal(words, pack_pairs(D0, want_ids))
            Du, Iu = unpack_pairs(words)
            assert torch.equal(Du, D0) and torch.equal(Iu, want_ids)
    search_mode(0)
    # list counts of the last chunk (44 queries): every query reached the refine step with at least k rows; the overflowed ones say so
    D0, I0 = idx.search(qd, k)
    cnts = idx.last_list_counts()
    assert cnts.numel() == 44 and (cnts >= k).all()
    assert (cnts[14:17] > 16384).all()                                        # queries 270..272 = the near-copy queries: more hits than the list holds
    # a sharded index with a row map goes through local_search -> finish without a packing launch and gives the mapped global rows
    sh = ShardedFlatIPIndex(idx, row_map=row_map)
    Dl, Il, W = sh.local_search(qd, k)
    assert W is not None and torch.equal(W, pack_pairs(D0, row_map[I0 - 1000]))
    Ds, Is = sh.finish(Dl, Il, W)
    # (the merge orders ties by GLOBAL row, the shard by local row: same scores, same rows, exact ties possibly permuted)
    assert torch.equal(Ds, D0) and torch.equal(Is.sort(dim=1).values, row_map[I0 - 1000].sort(dim=1).values)
    strict = torch.ones_like(D0, dtype=torch.bool)
    strict[:, 1:] &= D0[:, 1:] < D0[:, :-1]
    strict[:, :-1] &= D0[:, :-1] > D0[:, 1:]
    assert torch.equal(Is[strict], row_map[I0 - 1000][strict])
    # tiny shard (plain path) and k > ntotal padding
    small = _index(X[:300], id_base=5)
    words = torch.zeros(300, 400, dtype=torch.int64, device="cuda")
    Dp, Ip = small.search(qd, 400, wire_out=words)
    assert (Ip[:, 300:] == -1).all() and torch.equal(words, pack_pairs(Dp, Ip))


@pytest.mark.parametrize("cluster,k", [(800, 100), (6000, 300), (9000, 2048)])
def test_near_tie_cluster_around_the_kth_score_is_resolved_exactly(cluster, k, search_mode):
    """`cluster` rows within 3e-7 relative of each other straddle the k-th place -- far inside the fp32 accumulation noise of
    the six-product score matrix, so no selection BY those scores can be trusted.  The plain path and the gated fallback rescore every row
    within 2 eps6(q) of the k-th matrix score (all of the cluster; more than 4096 of them -> the streaming form) and must return exactly the
    fp64 oracle's ids under the (score desc, row asc) rule, on every path."""
    rng = np.random.default_rng(cluster + k)
    N, D, Q = 60000, 128, 6
    X = O.l2_normalize(rng.standard_normal((N, D)).astype(np.float32)) * np.float32(0.5)
    base = O.l2_normalize(rng.standard_normal((1, D)).astype(np.float32))[0]
    rows = rng.choice(N, size=cluster, replace=False)
    X[rows] = base[None, :] * (1.0 + rng.uniform(-3e-7, 3e-7, size=(cluster, 1))).astype(np.float32)
    better = rng.choice(np.setdiff1d(np.arange(N), rows), size=k // 2, replace=False)     # k/2 clearly better rows: the k-th place falls inside the cluster
    X[better] = base[None, :] * np.float32(1.5)
    q = np.repeat(base[None, :], Q, 0) * rng.uniform(0.5, 2.0, size=(Q, 1)).astype(np.float32)
    q[1:] += (1e-4 * rng.standard_normal((Q - 1, D))).astype(np.float32)
    Do, Io = flat_ip_topk_fp64(q, X, k)
    idx = _index(X)
    got = {}
    idx.two_pass = False
    got["plain"] = idx.search(q, k)
    idx.two_pass = True
    for mode in (2, 1):
        search_mode(mode)
        got["bounded%d" % mode] = idx.search(q, k)
    search_mode(0)
    for name, (Dg, Ig) in got.items():
        np.testing.assert_array_equal(Ig.cpu().numpy(), Io, err_msg=name)
        np.testing.assert_array_equal(Dg.cpu().numpy(), Do, err_msg=name)          # the exactly rescored fp32 values, bit for bit


def test_non_finite_query_does_not_poison_the_batch(search_mode):
    rng = np.random.default_rng(3)
    N, D, Q, k = 30000, 64, 40, 5
    X = O.l2_normalize(rng.standard_normal((N, D)).astype(np.float32))
    q = O.l2_normalize(rng.standard_normal((Q, D)).astype(np.float32))
    qn = q.copy()
    qn[7, 3] = np.inf
    idx = _index(X)
    search_mode(2)
    Dg, Ig = idx.search(qn, k)
    keep = np.arange(Q) != 7
    Do, Io = O.flat_ip_topk(q[keep], X, k)
    np.testing.assert_array_equal(Ig.cpu().numpy()[keep], Io)


def test_shard_commit_rows_kernel_shadow_and_bounds():
    from lightretriever_amd import FlatIPIndex
    rng = np.random.default_rng(1)
    N, D = 5003, 192
    X = (rng.standard_normal((N, D)) * rng.uniform(0.1, 4.0, size=(N, 1))).astype(np.float32)
    idx = FlatIPIndex(D)
    idx.add(X[:1000]); idx.add(X[1000:])
    xb = torch.from_numpy(X).cuda().to(torch.float16)
    assert idx._xb.ndim == 1 and torch.equal(idx.shadow_rows(), xb)            # tiled layout, same values
    R = np.linalg.norm(X.astype(np.float64), axis=1).max()
    E = np.linalg.norm(X.astype(np.float64) - xb.float().cpu().numpy().astype(np.float64), axis=1).max()
    b = idx._bounds.cpu().numpy()
    assert R <= b[0] <= R * (1 + 1e-5) and E * (1 - 1e-6) <= b[1] <= E * (1 + 1e-4)
    # no torch kernels are needed for maintenance: reset + refresh gives the same state again
    idx._xb.zero_(); idx._bounds.zero_()
    idx.refresh_norm_bound()
    assert torch.equal(idx.shadow_rows(), xb) and np.array_equal(idx._bounds.cpu().numpy(), b)
    # an index without a shadow measures the same bounds (bounds-only kernel; its own summation order -- equal to fp32 rounding)
    idx2 = FlatIPIndex(D)
    idx2.shadow_f16 = False
    idx2.add(X)
    assert idx2._xb is None and np.allclose(idx2._bounds.cpu().numpy(), b, rtol=2e-6, atol=0)


def test_encoder_writes_shadow_and_bounds_of_the_slot_it_fills():
    """lrx_encode_packed_shard: rows encoded into a FlatIPIndex slot arrive with their fp16 shadow and the shard bounds -- commit() has
    nothing left to do (no second pass over the rows), and searching them needs no refresh."""
    from dataclasses import asdict
    from lightretriever_amd import EncoderConfig, FlatIPIndex, LrxEncoder
    cfg_o = O.EncoderConfig(vocab_size=500, hidden_size=256, num_layers=2, num_q_heads=4, num_kv_heads=2, head_dim=64,
                            intermediate_size=512, rope_type="llama3", rope_original_max_position=64, max_positions=256)
    w = O.random_weights(cfg_o, seed=1, std=0.04)
    enc = LrxEncoder(EncoderConfig(**asdict(cfg_o)), {k: torch.from_numpy(v) for k, v in w.items()})
    rng = np.random.default_rng(0)
    lens = rng.integers(1, 120, size=37)
    ids = torch.from_numpy(rng.integers(0, 500, size=int(lens.sum())).astype(np.int32)).cuda()
    cu = torch.from_numpy(np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)).cuda()
    idx = FlatIPIndex(256, capacity=64)
    idx.add(np.zeros((3, 256), np.float32))                      # some rows already there
    slot = idx.append_slot(37)
    calls = []
    orig = idx._maintain
    idx._maintain = lambda a, b: (calls.append((a, b)), orig(a, b))
    enc.encode_packed(ids, cu, int(lens.max()), out=slot)
    idx.commit(37)
    assert all(b <= a for a, b in calls), f"commit() re-read encoder rows: {calls}"
    ref = enc.encode_packed(ids, cu, int(lens.max()))
    assert torch.equal(idx.vectors[3:], ref)
    assert torch.equal(idx.shadow_rows()[3:40], ref.to(torch.float16))
    b = idx._bounds.cpu().numpy()
    assert 1.0 <= b[0] < 1.00001 and 0 < b[1] < 2.0 ** -11
    # in-place re-encode of committed rows (what bench.py does) keeps shadow + bounds valid without refresh_norm_bound()
    enc.encode_packed(ids, cu, int(lens.max()), out=idx._x[3:40])
    assert torch.equal(idx.shadow_rows()[3:40], ref.to(torch.float16))
    # MRL slice narrower than the shard row: not a shard slot, plain output
    out = enc.encode_packed(ids, cu, int(lens.max()), out_dim=64)
    assert out.shape == (37, 64)


def test_rows_rewritten_after_a_partial_commit_get_fresh_shadow_and_bounds():
    """encode 100 rows into a slot, commit only 60, then add() other rows -- they land in rows 60.. that the encoder had
    vouched for.  The encoder's interval must not survive: the next commit maintains them, and the search equals the oracle."""
    from dataclasses import asdict
    from lightretriever_amd import EncoderConfig, FlatIPIndex, LrxEncoder
    cfg_o = O.EncoderConfig(vocab_size=500, hidden_size=256, num_layers=2, num_q_heads=4, num_kv_heads=2, head_dim=64,
                            intermediate_size=512, rope_type="llama3", rope_original_max_position=64, max_positions=256)
    w = O.random_weights(cfg_o, seed=1, std=0.04)
    enc = LrxEncoder(EncoderConfig(**asdict(cfg_o)), {k: torch.from_numpy(v) for k, v in w.items()})
    rng = np.random.default_rng(4)
    lens = rng.integers(1, 60, size=100)
    ids = torch.from_numpy(rng.integers(0, 500, size=int(lens.sum())).astype(np.int32)).cuda()
    cu = torch.from_numpy(np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)).cuda()
    idx = FlatIPIndex(256, capacity=40000)
    enc.encode_packed(ids, cu, int(lens.max()), out=idx.append_slot(100))
    idx.commit(60)
    assert idx._fused == []
    extra = (O.l2_normalize(rng.standard_normal((30000, 256)).astype(np.float32)) * np.float32(3.0))   # larger norms: stale bounds would be too small
    idx.add(extra)
    X = np.concatenate([idx.vectors[:60].cpu().numpy(), extra])
    assert torch.equal(idx.shadow_rows(), torch.from_numpy(X).cuda().to(torch.float16))
    assert idx._bounds[0].item() >= 3.0 * (1 - 1e-6)
    q = O.l2_normalize(rng.standard_normal((40, 256)).astype(np.float32))
    Dg, Ig = idx.search(q, 10)
    check_against_oracle(Dg, Ig, q, X, 10)
    # the same rows handed out twice before a commit: the second receiver's rows are maintained by the commit
    idx2 = FlatIPIndex(256, capacity=40000)
    enc.encode_packed(ids, cu, int(lens.max()), out=idx2.append_slot(100))
    idx2.append_slot(30000).copy_(torch.from_numpy(extra))
    idx2.commit(30000)
    assert torch.equal(idx2.shadow_rows(), torch.from_numpy(extra).cuda().to(torch.float16))
    check_against_oracle(*idx2.search(q, 10), q, extra, 10)


def test_shadow_switched_on_after_rows_were_added_is_built_from_the_rows():
    """shadow_f16 = True on an index that already holds rows must not leave their shadow uninitialised."""
    from lightretriever_amd import FlatIPIndex
    rng = np.random.default_rng(12)
    X = O.l2_normalize(rng.standard_normal((30000, 128)).astype(np.float32))
    idx = FlatIPIndex(128, capacity=40000)
    idx.shadow_f16 = False
    idx.add(X[:20000])
    assert idx._xb is None
    idx.shadow_f16 = True
    idx.add(X[20000:])
    assert torch.equal(idx.shadow_rows(), torch.from_numpy(X).cuda().to(torch.float16))
    q = O.l2_normalize(rng.standard_normal((50, 128)).astype(np.float32))
    check_against_oracle(*idx.search(q, 10), q, X, 10)


def test_shadow_switched_off_and_on_again_covers_the_rows_added_in_between():
    """with an EXISTING shadow, rows committed while shadow_f16 was off (bounds only) must get their shadow rows before the
    filter streams them -- the index tracks how many committed rows the shadow covers."""
    from lightretriever_amd import FlatIPIndex
    rng = np.random.default_rng(13)
    X = O.l2_normalize(rng.standard_normal((40000, 128)).astype(np.float32))
    idx = FlatIPIndex(128, capacity=40000)
    idx.add(X[:15000])                                   # shadow on: 15 000 rows covered
    assert idx._shadow_rows == 15000
    idx.shadow_f16 = False
    idx.add(X[15000:28000])                              # bounds only
    assert idx._shadow_rows == 15000 and idx.ntotal == 28000
    q = O.l2_normalize(rng.standard_normal((40, 128)).astype(np.float32))
    check_against_oracle(*idx.search(q, 10), q, X[:28000], 10)           # shadow off: searched from the fp32 rows
    idx.shadow_f16 = True
    check_against_oracle(*idx.search(q, 10), q, X[:28000], 10)           # on again, BEFORE any further add: search() completes the shadow first
    assert idx._shadow_rows == 28000
    idx.add(X[28000:])
    assert idx._shadow_rows == 40000
    assert torch.equal(idx.shadow_rows(), torch.from_numpy(X).cuda().to(torch.float16))
    check_against_oracle(*idx.search(q, 10), q, X, 10)
    idx.reset()
    assert idx._shadow_rows == 0


# ---- BASELINE index shapes (configs 2-5): properties that do not need a CPU pass over the whole index -------------------------------
def _fill_normalised(idx, N, D, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    slot = idx.append_slot(N)
    step = max(1, (1 << 28) // D)
    for s in range(0, N, step):
        e = min(s + step, N)
        slot[s:e] = torch.nn.functional.normalize(torch.randn(e - s, D, generator=g, device="cuda"), dim=-1)
    idx.commit(N)
    return g


def _check_properties(idx, q, k, planted, Dg, Ig, chunk):
    N, D, Q = idx.ntotal, idx.d, q.shape[0]
    assert (Ig[:len(planted), 0] == planted).all() and torch.allclose(Dg[:len(planted), 0], torch.ones(len(planted), device="cuda"), atol=1e-5)
    assert (Dg[:, 1:] <= Dg[:, :-1]).all()
    assert all(len(set(r.tolist())) == k for r in Ig.cpu())
    rec = torch.einsum("qkd,qd->qk", idx.vectors[Ig.reshape(-1)].view(Q, k, D).double(), q.double()).float()
    assert torch.allclose(rec, Dg, atol=3e-6)
    best = torch.full((Q,), -2.0, device="cuda")
    better = torch.zeros(Q, device="cuda")
    for s in range(0, N, chunk):
        sc = q @ idx.vectors[s:s + chunk].T
        best = torch.maximum(best, sc.max(dim=1).values)
        better = better + (sc > Dg[:, -1:] + 3e-6).sum(1)
    assert torch.allclose(best, Dg[:, 0], atol=3e-6)
    assert (better <= k - 1).all()                                # nothing clearly better than the k-th result was missed


@pytest.mark.parametrize("N,D", [(1_000_000, 4096), (10_000_000, 256)], ids=["config2_1Mx4096", "config5_10Mx256_mrl"])
def test_baseline_index_shapes_properties(N, D, search_mode):
    from lightretriever_amd import FlatIPIndex, merge_topk
    Q, k = 100, 100
    idx = FlatIPIndex(D, capacity=N)
    g = _fill_normalised(idx, N, D, seed=11)
    q = torch.nn.functional.normalize(torch.randn(Q, D, generator=g, device="cuda"), dim=-1)
    planted = torch.randint(0, N, (10,), generator=g, device="cuda")
    q[:10] = idx.vectors[planted]
    Dg, Ig = idx.search(q, k)
    _check_properties(idx, q, k, planted, Dg, Ig, chunk=max(1, (1 << 29) // D // 4))
    search_mode(1)                                                # the score-matrix filter gives the same bits
    Dm, Im = idx.search(q, k)
    search_mode(0)
    assert torch.equal(Im, Ig) and torch.equal(Dm, Dg)
    # eight-way row shard + merge == whole index (config 4's layout; views of the same rows, no copy)
    Dp, Ip = [], []
    cuts = [(r * N // 8) // 128 * 128 for r in range(8)] + [N]       # whole 128-row blocks of the tiled shadow per shard
    for r in range(8):
        a, b = cuts[r], cuts[r + 1]
        sh = FlatIPIndex(D, id_base=a)
        sh._x, sh._xb, sh._bounds, sh.ntotal = idx._x[a:b], idx._xb[a * D:], idx._bounds, b - a
        d, i = sh.search(q, k)
        Dp.append(d), Ip.append(i)
    Dm8, Im8 = merge_topk(torch.stack(Dp), torch.stack(Ip))
    assert torch.equal(Im8, Ig) and torch.equal(Dm8, Dg)


@pytest.mark.skipif(torch.cuda.is_available() and torch.cuda.get_device_properties(0).total_memory < 270 << 30, reason="needs one 288-GB GPU")
def test_config3_10m_x_4096_single_gpu_properties():
    """BASELINE config 3's whole index on ONE GPU: 164 GB of fp32 rows + 82 GB fp16 shadow + < 6 GB of search workspace.  Deselect
    with -k 'not config3' when the box is short on time: filling the rows takes about a minute."""
    from lightretriever_amd import FlatIPIndex
    N, D, Q, k = 10_000_000, 4096, 100, 100
    idx = FlatIPIndex(D, capacity=N)
    idx._ensure_shadow()
    g = _fill_normalised(idx, N, D, seed=13)
    q = torch.nn.functional.normalize(torch.randn(Q, D, generator=g, device="cuda"), dim=-1)
    planted = torch.randint(0, N, (10,), generator=g, device="cuda")
    q[:10] = idx.vectors[planted]
    Dg, Ig = idx.search(q, k)
    _check_properties(idx, q, k, planted, Dg, Ig, chunk=32768)
    assert idx._ws.numel() < 6 << 30


def test_search_on_a_side_stream_equals_default_stream(search_mode):
    """The library launches on the stream it is handed (the caller's current HIP stream): a search issued on a non-default stream --
    what an RPC server thread or a prefetching caller does -- gives the same bits, for both filters."""
    rng = np.random.default_rng(17)
    N, D, Q, k = 40000, 256, 150, 20
    X = O.l2_normalize(rng.standard_normal((N, D)).astype(np.float32))
    q = torch.from_numpy(O.l2_normalize(rng.standard_normal((Q, D)).astype(np.float32))).cuda()
    idx = _index(X)
    ref = {}
    for mode in (1, 2):
        search_mode(mode)
        ref[mode] = idx.search(q, k)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for mode in (1, 2):
            search_mode(mode)
            for _ in range(3):
                Ds, Is = idx.search(q, k)
            side.synchronize()
            assert torch.equal(Ds, ref[mode][0]) and torch.equal(Is, ref[mode][1])
    torch.cuda.current_stream().wait_stream(side)


@pytest.mark.parametrize("N,D,Q,k,two_pass", [(200000, 256, 100, 100, True), (40000, 1024, 200, 10, True), (30000, 128, 48, 1000, True), (30000, 64, 20, 10, False),
                                                     (50000, 1024, 700, 20, True)])      # (round 6: a wide chunk -- three query n-tiles, both passes on the GEMM kernel -- in one graph)
def test_search_captured_in_a_hip_graph_replays_bit_identically(N, D, Q, k, two_pass):
    """VERDICT r2 item 7.  The library never allocates and never synchronises, so a search is capturable: a torch.cuda.graph of
    FlatIPIndex.search replays the eager result bit for bit, also with new queries written into the captured input buffer (score-free chain,
    GEMM main pass for 129..256 queries, k = 1000, plain path).  The one allocation on the way is the index's lazily sized workspace: made
    under capture it would belong to the graph's private pool -- search() refuses that (the cause of round 2's memory-access fault:
    tools/exp/graph_probe.py shows the workspace address handed out again once the graph is gone)."""
    from lightretriever_amd import FlatIPIndex, _lib
    g = torch.Generator(device="cuda").manual_seed(N + Q)
    idx = FlatIPIndex(D, capacity=N)
    idx.two_pass = two_pass
    slot = idx.append_slot(N)
    slot.copy_(torch.nn.functional.normalize(torch.randn(N, D, generator=g, device="cuda"), dim=-1))
    idx.commit(N)
    q = torch.nn.functional.normalize(torch.randn(Q, D, generator=g, device="cuda"), dim=-1)
    cold = torch.cuda.CUDAGraph()
    with pytest.raises(_lib.LrxError, match="workspace must exist"):
        with torch.cuda.graph(cold):                                  # first search of this shape under capture: refused, nothing launched
            idx.search(q, k)
    del cold
    De, Ie = idx.search(q, k)
    De, Ie = De.clone(), Ie.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        idx.search(q, k)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        Dg, Ig = idx.search(q, k)
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(Ig, Ie) and torch.equal(Dg, De)
    q2 = torch.nn.functional.normalize(torch.randn(Q, D, generator=g, device="cuda"), dim=-1)
    D2, I2 = idx.search(q2, k)
    D2, I2 = D2.clone(), I2.clone()
    q.copy_(q2)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(Ig, I2) and torch.equal(Dg, D2)
    check_against_oracle(Dg, Ig, q2.cpu().numpy(), idx.vectors.cpu().numpy(), k)


@pytest.mark.parametrize("N,D,Q,k", [(50000, 128, 100, 10), (33001, 2048, 300, 5), (20011, 64, 1, 50), (70000, 1024, 256, 100), (9000, 192, 40, 7),
                                     # N mod 256 in (0, 128]: a score-matrix launch covers one 128-row block more than the tiled shadow holds
                                     (4200, 64, 40, 5), (70001, 128, 3, 20)])
def test_shadow_built_in_pieces_and_across_reallocations_gives_the_same_bits(N, D, Q, k, search_mode):
    """Rows that arrive in pieces (add), through a shard that is re-allocated with committed rows in it, and shards that end inside a 128-row
    block: same fp16 shadow values, same hits on both filters."""
    from lightretriever_amd import FlatIPIndex
    rng = np.random.default_rng(N + Q)
    X = O.l2_normalize(rng.standard_normal((N, D)).astype(np.float32)) * rng.uniform(0.3, 2.0, size=(N, 1)).astype(np.float32)
    q = O.l2_normalize(rng.standard_normal((Q, D)).astype(np.float32))
    res = {}
    idx = FlatIPIndex(D, capacity=N // 3)                           # grows twice: the shadow is re-allocated with committed rows in it
    for s in range(0, N, 7001):
        idx.add(X[s:s + 7001])
    assert torch.equal(idx.shadow_rows(), torch.from_numpy(X).cuda().to(torch.float16))
    for mode in (1, 2):
        search_mode(mode)
        res[mode] = idx.search(q, k)
    search_mode(0)
    assert torch.equal(res[1][0], res[2][0]) and torch.equal(res[1][1], res[2][1])
    check_against_oracle(*res[2], q, X, k)
