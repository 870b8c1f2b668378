"""torch.ops.lrx.* (TORCH_LIBRARY binding) against the ctypes binding of the same C ABI and the oracle."""
from dataclasses import asdict

import numpy as np
import pytest
import torch

from oracle import lrx_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lrx():
    from lightretriever_amd import torch_ops
    return torch_ops.load()


def test_encode_packed_op_equals_ctypes_path_and_oracle(lrx):
    from lightretriever_amd import EncoderConfig, LrxEncoder
    cfg_o = O.EncoderConfig(vocab_size=500, hidden_size=256, num_layers=2, num_q_heads=4, num_kv_heads=2, head_dim=64,
                            intermediate_size=512, rope_type="llama3", rope_original_max_position=64, max_positions=256)
    w = O.random_weights(cfg_o, seed=1, std=0.04)
    enc = LrxEncoder(EncoderConfig(**asdict(cfg_o)), {k: torch.from_numpy(v) for k, v in w.items()})
    rng = np.random.default_rng(0)
    lens = [70, 1, 33, 128, 5]
    ids = rng.integers(0, 500, size=sum(lens)).astype(np.int32)
    cu = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    tid, tcu = torch.from_numpy(ids).cuda(), torch.from_numpy(cu).cuda()
    out = torch.empty(len(lens), 256, device="cuda")
    assert lrx.encode_packed(tid, tcu, max(lens), enc.handle, out) is None
    ref = enc.encode_packed(tid, tcu, max(lens))
    assert torch.equal(out, ref)
    want = O.encode_passage(cfg_o, w, ids, cu)
    assert ((out.cpu().numpy() * want).sum(-1)).min() > 1 - 5e-3
    mrl = torch.empty(len(lens), 64, device="cuda")
    lrx.encode_packed(tid, tcu, max(lens), enc.handle, mrl, 64, True)
    assert torch.equal(mrl, enc.encode_packed(tid, tcu, max(lens), out_dim=64))
    # `out` = committed rows of an index shard -- the op takes the shard's shadow + bounds like the ctypes path does by itself, so the
    # rows' fp16 shadow and the bounds follow an in-place re-encode (without them the caller must run shard_commit_rows afterwards)
    from lightretriever_amd import FlatIPIndex
    idx = FlatIPIndex(256, capacity=64)
    idx.add(torch.randn(10, 256, device="cuda") * 0.01)
    idx._bounds.zero_()
    lrx.encode_packed(tid, tcu, max(lens), enc.handle, idx._x[3:3 + len(lens)], 0, True, idx._xb, 3, idx._bounds)
    assert torch.equal(idx._x[3:8], ref) and torch.equal(idx.shadow_rows()[3:8], ref.to(torch.float16)) and 1.0 <= float(idx._bounds[0]) < 1.00001
    with pytest.raises(RuntimeError, match="tiled"):
        lrx.encode_packed(tid, tcu, max(lens), enc.handle, idx._x[3:8], 0, True, idx._xb.view(-1, 256), 3, idx._bounds)      # not the 1-D tiled shadow
    # errors surface as RuntimeError with liblrx's message
    with pytest.raises(RuntimeError, match="liblrx error|must be"):
        lrx.encode_packed(tid.long(), tcu, max(lens), enc.handle, out)
    with pytest.raises(RuntimeError, match="liblrx error"):
        lrx.encode_packed(tid, tcu, 100000, enc.handle, out)            # beyond the RoPE table


def test_search_ops(lrx):
    from lightretriever_amd import FlatIPIndex, merge_topk
    rng = np.random.default_rng(1)
    N, D, Q, k = 40000, 128, 50, 10
    X = O.l2_normalize(rng.standard_normal((N, D)).astype(np.float32))
    q = O.l2_normalize(rng.standard_normal((Q, D)).astype(np.float32))
    Xd, qd = torch.from_numpy(X).cuda(), torch.from_numpy(q).cuda()
    Do, Io = O.flat_ip_topk(q, X, k)
    D1, I1 = lrx.flat_ip_topk(qd, Xd, k)
    np.testing.assert_array_equal(I1.cpu().numpy(), Io)
    np.testing.assert_allclose(D1.cpu().numpy(), Do, atol=2e-6)
    xb = torch.empty(-(-N // 128) * 128 * D, dtype=torch.float16, device="cuda")       # the tiled fp16 shadow (include/lrx.h), whole 128-row blocks
    bounds = torch.zeros(2, device="cuda")
    lrx.shard_commit_rows(Xd, xb, bounds)
    idx = FlatIPIndex(D, id_base=7)
    idx.add(X)
    n_full = (N // 128) * 128 * D                                   # (the padding rows of the last, partly filled block are never written)
    assert torch.equal(xb[:n_full], idx._xb[:n_full]) and torch.equal(idx.shadow_rows(), Xd.to(torch.float16)) and 1.0 <= float(bounds[0]) < 1.00001
    assert torch.equal(bounds, idx._bounds)
    D2, I2 = lrx.flat_ip_topk_bounded(qd, Xd, xb, bounds, k, 7)
    assert torch.equal(I2, I1 + 7) and torch.equal(D2, D1)
    for flags in (1, 2):                                            # the filter choice travels with the call (no process-wide switch)
        Df, If = lrx.flat_ip_topk_bounded(qd, Xd, xb, bounds, k, 7, flags)
        assert torch.equal(If, I2) and torch.equal(Df, D2)
    D1b, I1b = lrx.flat_ip_topk(qd, Xd, k, 0, bounds)               # plain path with the shard's bounds handed over: no extra pass, same hits
    assert torch.equal(I1b, I1) and torch.equal(D1b, D1)
    D3, I3 = idx.search(qd, k)                                       # the ctypes path: same bits
    assert torch.equal(I3, I2) and torch.equal(D3, D2)
    Dm, Im = lrx.merge_topk(torch.stack([D1[:, :5], D1[:, 5:]]).contiguous(), torch.stack([I1[:, :5], I1[:, 5:]]).contiguous())
    Dm2, Im2 = merge_topk(torch.stack([D1[:, :5], D1[:, 5:]]), torch.stack([I1[:, :5], I1[:, 5:]]))
    assert torch.equal(Dm, Dm2) and torch.equal(Im, Im2) and torch.equal(Im, I1[:, :5])
    # the sharded form: the search's last kernel writes the exchange words; two "ranks" (halves of the rows, whole 128-row blocks each) gathered
    # and merged == the whole shard
    from lightretriever_amd.sharded import pack_pairs
    row_map = torch.arange(N, device="cuda", dtype=torch.int64) * 2 + 1
    Dw, Iw, W = lrx.flat_ip_topk_bounded_wire(qd, Xd, xb, bounds, k, 7, row_map)
    assert torch.equal(Dw, D2) and torch.equal(Iw, I2) and torch.equal(W, pack_pairs(D2, row_map[I2 - 7]))
    half = (N // 2) // 128 * 128
    parts = []
    for a, b in ((0, half), (half, N)):
        bnd = torch.zeros(2, device="cuda")
        xs = torch.empty(-(-(b - a) // 128) * 128 * D, dtype=torch.float16, device="cuda")
        lrx.shard_commit_rows(Xd[a:b], xs, bnd)
        parts.append(lrx.flat_ip_topk_bounded_wire(qd, Xd[a:b], xs, bnd, k, a)[2])
    Dg, Ig = lrx.merge_topk_packed(torch.stack(parts).contiguous())
    assert torch.equal(Ig, I1) and torch.equal(Dg, D1)
    with pytest.raises(RuntimeError, match="liblrx error"):
        lrx.flat_ip_topk(qd, Xd, 5000)


def test_query_and_unit_kernel_ops(lrx):
    from lightretriever_amd import ops
    rng = np.random.default_rng(2)
    table = torch.from_numpy(rng.standard_normal((300, 64)).astype(np.float32)).cuda()
    ids = torch.from_numpy(rng.integers(0, 300, size=40)).cuda()
    offs = torch.tensor([0, 7, 7, 30], device="cuda")
    got = lrx.embedding_bag_mean(table, ids, offs, -1, 0, True)
    assert torch.equal(got, ops.embedding_bag_mean(table, ids, offs, normalize=True))
    np.testing.assert_allclose(got.cpu().numpy(), O.encode_query_emb(table.cpu().numpy(), ids.cpu().numpy(), offs.cpu().numpy()), atol=1e-6)
    x = torch.randn(100, 256, device="cuda").bfloat16()
    w = torch.randn(256, device="cuda").bfloat16()
    assert torch.equal(lrx.rmsnorm(x, w, 1e-5), ops.rmsnorm(x, w, 1e-5))
    a = (torch.randn(300, 128, device="cuda") * 0.5).bfloat16()
    wgu = (torch.randn(512, 128, device="cuda") * 0.1).bfloat16()
    assert torch.equal(lrx.swiglu_gemm(a, wgu), ops.gemm_bf16_nt(a, wgu, epilogue=2))
    # fused qkv + rope and attention: the op pair equals the ctypes pair
    nq, nkv, d, T = 4, 2, 64, 300
    wqkv = (torch.randn((nq + 2 * nkv) * d, 128, device="cuda") * 0.1).bfloat16()
    cu = torch.tensor([0, 100, 101, 300], dtype=torch.int32, device="cuda")
    pos = ops.build_positions(cu, T)
    cos = torch.cos(torch.arange(512, device="cuda")[:, None] * 0.01 * torch.arange(d // 2, device="cuda")[None, :]).contiguous()
    sin = torch.sin(torch.arange(512, device="cuda")[:, None] * 0.01 * torch.arange(d // 2, device="cuda")[None, :]).contiguous()
    qkv = lrx.rope_qkv_gemm(a, wqkv, None, pos, cos, sin, nq, nkv, d)
    assert torch.equal(qkv, ops.gemm_qkv_rope(a, wqkv, pos, cos, sin, nq, nkv, d))
    assert torch.equal(lrx.attn_varlen(qkv, cu, 200, nq, nkv, d), ops.attn_varlen_causal(qkv, cu, 200, nq, nkv, d))
