"""The attention launch on a prebuilt work list (include/lrx.h, ABI 7: lrx_attn_items_bytes / lrx_attn_build_items /
lrx_attn_varlen_causal_items) against the launch without one: the same bits, whatever the head layout, the batch size (grouped lists once the
items outnumber the workgroup slots) and the mode.  Oracle parity of the list kernel itself: tests/test_gpu_kernels.py (ops.attn_varlen_causal
builds a list by default).  Replaces the FA2 varlen call of utils/nested_input.py:137-146."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _lens(seed, n, hi):
    return [int(x) for x in np.random.default_rng(seed).integers(1, hi + 1, n)]


CASES = [
    ("8b_uniform", 32, 8, 128, [512] * 64, False),
    ("8b_ragged_grouped", 32, 8, 128, _lens(1, 300, 512), False),          # 300 x 8 x 8 items > 256 slots: XCD-grouped lists
    ("8b_last_tile", 32, 8, 128, _lens(2, 300, 512), True),
    ("d128_g1", 8, 8, 128, _lens(3, 64, 300), False),
    ("d128_g2_long", 8, 4, 128, _lens(4, 64, 700), False),
    ("d128_g3", 12, 4, 128, _lens(5, 64, 300), False),
    ("d128_g6_two_parts", 12, 2, 128, _lens(6, 64, 300), False),
    ("d128_g8_two_parts", 16, 2, 128, _lens(7, 64, 300), False),
    ("d64_g4_long", 32, 8, 64, _lens(8, 40, 1500), False),
    ("d64_g7_long", 14, 2, 64, _lens(9, 40, 900), False),
    ("d64_g16_two_parts", 16, 1, 64, _lens(10, 12, 700), False),
    # MHA / g = 2 at d = 64: four slots per CU -- 1024 workgroups on a 256-CU part, the one-block builder's limit (attn_plan clamps there)
    ("d64_mha_g1_long_full_grid", 16, 16, 64, _lens(13, 24, 1100), False),
    ("d64_g2_long_full_grid", 16, 8, 64, _lens(14, 40, 900), False),
    ("d64_mha_last_tile", 16, 16, 64, _lens(15, 200, 512), True),
    ("d64_resident_ignores_the_list", 32, 8, 64, _lens(11, 50, 512), False),
    ("many_short_sequences", 8, 2, 128, [3, 70, 1, 129] * 1250, False),     # 5000 sequences: 10 000 (sequence, kv head) pairs through the one-block builder
    ("many_short_last_tile", 8, 2, 128, [3, 70, 1, 129] * 1250, True),
    ("one_long_sequence", 8, 2, 128, [8192, 5], False),                      # 128 q tiles: groups of 32 workgroups per (sequence, kv head)
    ("d64_one_long_sequence", 8, 2, 64, [6000], False),
    ("tiny", 32, 8, 128, [1, 2, 63, 64, 65], False),
    ("one_token", 32, 8, 128, [1], True),
]


def _inputs(nq, nkv, d, lens, seed=7):
    g = torch.Generator(device="cuda").manual_seed(seed)
    qkv = torch.randn(sum(lens), (nq + 2 * nkv) * d, generator=g, device="cuda").to(torch.float16)
    cu = torch.tensor([0] + lens, dtype=torch.int64).cumsum(0).to(torch.int32).cuda()
    return qkv, cu


def _no_overflow():
    import ctypes as C
    from lightretriever_amd import _lib
    n = C.c_int(-1)
    assert _lib.lib().lrx_debug_attn_items_overflow(C.byref(n)) == 0
    assert n.value == 0, "the work-list builder ran out of list slots"


@pytest.mark.parametrize("name,nq,nkv,d,lens,last", CASES, ids=[c[0] for c in CASES])
def test_work_list_launch_equals_the_launch_without_a_list(name, nq, nkv, d, lens, last):
    from lightretriever_amd import ops
    qkv, cu = _inputs(nq, nkv, d, lens)
    walker = ops.attn_varlen_causal(qkv, cu, max(lens), nq, nkv, d, last_tile_only=last, work_list=False)
    listed = ops.attn_varlen_causal(qkv, cu, max(lens), nq, nkv, d, last_tile_only=last)
    assert torch.equal(walker, listed)          # (last-tile mode: untouched rows are zeros in both)
    assert not torch.isnan(listed.float()).any()
    _no_overflow()


def test_one_list_serves_every_launch_of_a_batch_and_a_loose_max_seqlen():
    """The encoder's use: one list per batch, read by the launch of every layer (different q|k|v each time); max_seqlen may be any upper
    bound of the longest sequence as long as list and launch agree on it."""
    from lightretriever_amd import ops
    nq, nkv, d, lens = 32, 8, 128, _lens(12, 280, 400)
    _, cu = _inputs(nq, nkv, d, lens)
    for msl in (max(lens), 512, 1000):
        wl = ops.attn_work_list(cu, sum(lens), msl, nq, nkv, d)
        for seed in (1, 2):
            qkv, _ = _inputs(nq, nkv, d, lens, seed=seed)
            assert torch.equal(ops.attn_varlen_causal(qkv, cu, msl, nq, nkv, d, work_list=wl),
                               ops.attn_varlen_causal(qkv, cu, max(lens), nq, nkv, d, work_list=False))
    _no_overflow()


def test_list_size_is_monotone_and_bounded_by_the_batch():
    """lrx_encode_workspace_bytes sizes the lists at max_positions: the size must not shrink with max_seqlen or total_tokens, and it is
    bounded by the q tiles the batch can hold ((total_tokens / 64 + n_seqs) x kv heads), not by n_seqs x the longest sequence's tiles."""
    from lightretriever_amd import _lib
    L = _lib.lib()
    for n_seqs in (1, 7, 256, 4096):
        for last in (0, 1):
            T = n_seqs * 300
            sizes = [L.lrx_attn_items_bytes(n_seqs, T, s, 32, 8, 128, last) for s in (1, 64, 65, 512, 513, 2048, 8192, 131072)]
            assert all(a <= b for a, b in zip(sizes, sizes[1:])), (n_seqs, last, sizes)
            assert sizes[-1] <= 16 * ((T // 64 + n_seqs) * 8 + 1024 + 3) + 4 * 1028, (n_seqs, last, sizes)
            by_t = [L.lrx_attn_items_bytes(n_seqs, t, 512, 32, 8, 128, last) for t in (n_seqs, 10 * n_seqs, 100 * n_seqs, 512 * n_seqs)]
            assert all(a <= b for a, b in zip(by_t, by_t[1:])), (n_seqs, last, by_t)
    assert L.lrx_attn_items_bytes(0, 10, 512, 32, 8, 128, 0) == 0 and L.lrx_attn_items_bytes(4, 100, 512, 32, 8, 96, 0) == 0


def test_list_arguments_are_checked():
    from lightretriever_amd import _lib, ops
    L = _lib.lib()
    nq, nkv, d, lens = 32, 8, 128, [100, 200]
    qkv, cu = _inputs(nq, nkv, d, lens)
    wl = ops.attn_work_list(cu, sum(lens), 200, nq, nkv, d)
    out = torch.empty(sum(lens), nq * d, dtype=torch.bfloat16, device="cuda")
    args = (_lib.ptr(qkv), _lib.ptr(cu))
    assert L.lrx_attn_varlen_causal_items(*args, _lib.ptr(wl), 16, 2, sum(lens), 200, nq, nkv, d, _lib.ptr(out), 0, None) == -1     # list too small
    assert L.lrx_attn_varlen_causal_items(*args, None, wl.numel(), 2, sum(lens), 200, nq, nkv, d, _lib.ptr(out), 0, None) == -1
    assert L.lrx_attn_build_items(_lib.ptr(cu), 2, sum(lens), 200, nq, nkv, d, 0, _lib.ptr(wl), 16, None) == -1
    assert L.lrx_attn_build_items(_lib.ptr(cu), 2, sum(lens), 200, nq, 5, d, 0, _lib.ptr(wl), wl.numel(), None) == -1                            # nq % nkv
    assert L.lrx_attn_varlen_causal_items(*args, _lib.ptr(wl), wl.numel(), 0, 0, 0, nq, nkv, d, _lib.ptr(out), 0, None) == 0          # empty batch
