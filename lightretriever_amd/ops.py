"""Thin torch-tensor wrappers over the individual liblrx kernels (unit tests, query side, EmbeddingBag construction)."""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib


def _s():
    return _lib.current_stream()


def embedding_gather(table: torch.Tensor, ids: torch.Tensor) -> torch.Tensor:
    out = torch.empty(ids.numel(), table.shape[1], dtype=torch.bfloat16, device=table.device)
    _lib.check(_lib.lib().lrx_embedding_gather(_lib.ptr(table), _lib.ptr(ids), ids.numel(), table.shape[1], table.shape[0], _lib.ptr(out), _s()))
    return out


def rmsnorm(x: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:
    y = torch.empty_like(x)
    _lib.check(_lib.lib().lrx_rmsnorm(_lib.ptr(x), _lib.ptr(w), _lib.ptr(y), x.shape[0], x.shape[1], eps, _s()))
    return y


def gemm_bf16_nt(A: torch.Tensor, B: torch.Tensor, bias: Optional[torch.Tensor] = None, resid: Optional[torch.Tensor] = None,
                 epilogue: int = 0, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    M, K = A.shape
    N = B.shape[0]
    assert B.shape[1] == K and A.is_contiguous() and B.is_contiguous()
    if out is None:
        out = torch.empty(M, N // 2 if epilogue == 2 else N, dtype=torch.bfloat16, device=A.device)
    _lib.check(_lib.lib().lrx_gemm_bf16_nt(_lib.ptr(A), _lib.ptr(B), _lib.ptr(out), _lib.ptr(bias), _lib.ptr(resid), M, N, K, epilogue, _s()))
    return out


def gemm_bf16_nt_fused(A: torch.Tensor, B: torch.Tensor, bias: Optional[torch.Tensor] = None, resid: Optional[torch.Tensor] = None, epilogue: int = 0,
                       rscale: Optional[torch.Tensor] = None, want_ss: bool = False):
    """lrx_gemm_bf16_nt_fused: -> (out bf16, ss_part fp32 [ceil(N/256), M] or None)."""
    M, K = A.shape
    N = B.shape[0]
    out = torch.empty(M, N // 2 if epilogue == 2 else N, dtype=torch.bfloat16, device=A.device)
    ss = torch.full(((N + 255) // 256, M), float("nan"), dtype=torch.float32, device=A.device) if want_ss else None
    _lib.check(_lib.lib().lrx_gemm_bf16_nt_fused(_lib.ptr(A), _lib.ptr(B), _lib.ptr(out), _lib.ptr(bias), _lib.ptr(resid), M, N, K, epilogue,
                                                 _lib.ptr(rscale), _lib.ptr(ss), _s()))
    return out, ss


def row_rscale(x: torch.Tensor, eps: float) -> torch.Tensor:
    out = torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().lrx_row_rscale(_lib.ptr(x), x.shape[0], x.shape[1], float(eps), _lib.ptr(out), _s()))
    return out


def finalize_rscale(ss_part: torch.Tensor, hidden_size: int, eps: float) -> torch.Tensor:
    out = torch.empty(ss_part.shape[1], dtype=torch.float32, device=ss_part.device)
    _lib.check(_lib.lib().lrx_finalize_rscale(_lib.ptr(ss_part), ss_part.shape[0], ss_part.shape[1], hidden_size, float(eps), _lib.ptr(out), _s()))
    return out


def rotary_pair_order(nq: int, nkv: int, d: int) -> torch.Tensor:
    """Row permutation of the fused qkv projection (include/lrx.h, lrx_layer_weights.wqkv): physical row 32 g + 16 i + t of every q and k head
    = logical row i * d/2 + 16 g + t; v rows stay in place.  `W_physical = W_logical[perm]`; the same index applies to the bias and, read
    the other way (`x_logical[..., perm] = x_physical`), to the q | k columns the fused kernel writes."""
    t = torch.arange(d)
    g, i, r = t // 32, (t % 32) // 16, t % 16
    head = i * (d // 2) + 16 * g + r
    rot = torch.cat([h * d + head for h in range(nq + nkv)])
    return torch.cat([rot, torch.arange((nq + nkv) * d, (nq + 2 * nkv) * d)])


def gemm_qkv_rope(A: torch.Tensor, Wqkv: torch.Tensor, positions: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, nq: int, nkv: int,
                  d: int, bias: Optional[torch.Tensor] = None, rscale: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Wqkv / bias rows in rotary-pair order (rotary_pair_order); -> fp16 [M, (nq+2nkv)d], q | k columns in the same order."""
    M, K = A.shape
    out = torch.empty(M, (nq + 2 * nkv) * d, dtype=torch.float16, device=A.device)
    _lib.check(_lib.lib().lrx_gemm_qkv_rope_fused(_lib.ptr(A), _lib.ptr(Wqkv), _lib.ptr(out), _lib.ptr(bias), _lib.ptr(positions), _lib.ptr(cos),
                                                  _lib.ptr(sin), M, K, nq, nkv, d, _lib.ptr(rscale), _s()))
    return out


def gemm_resid32(A: torch.Tensor, B: torch.Tensor, x32: torch.Tensor, gamma: Optional[torch.Tensor] = None, want_a16: bool = True, want_ss: bool = False):
    """lrx_gemm_bf16_nt_resid32: x32 (fp32 [M,N], in place) += A . B^T -> (a16 bf16 [M,N] = bf16(x32 * gamma) or None, ss_part or None)."""
    M, K = A.shape
    N = B.shape[0]
    a16 = torch.empty(M, N, dtype=torch.bfloat16, device=A.device) if want_a16 else None
    ss = torch.full(((N + 255) // 256, M), float("nan"), dtype=torch.float32, device=A.device) if want_ss else None
    _lib.check(_lib.lib().lrx_gemm_bf16_nt_resid32(_lib.ptr(A), _lib.ptr(B), _lib.ptr(x32), _lib.ptr(a16), _lib.ptr(gamma), M, N, K, _lib.ptr(ss), _s()))
    return a16, ss


def embed_stream32(table: torch.Tensor, ids: torch.Tensor, gamma: torch.Tensor, eps: float):
    """lrx_embed_stream32 -> (x32 fp32 [T,H], a16 bf16 [T,H], rscale fp32 [T])."""
    T, H = ids.numel(), table.shape[1]
    x32 = torch.empty(T, H, dtype=torch.float32, device=table.device)
    a16 = torch.empty(T, H, dtype=torch.bfloat16, device=table.device)
    rs = torch.empty(T, dtype=torch.float32, device=table.device)
    _lib.check(_lib.lib().lrx_embed_stream32(_lib.ptr(table), _lib.ptr(ids), T, H, table.shape[0], _lib.ptr(gamma), _lib.ptr(x32), _lib.ptr(a16), _lib.ptr(rs),
                                             float(eps), _s()))
    return x32, a16, rs


def rmsnorm_f32(x: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:
    y = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    _lib.check(_lib.lib().lrx_rmsnorm_f32(_lib.ptr(x), _lib.ptr(w), _lib.ptr(y), x.shape[0], x.shape[1], float(eps), _s()))
    return y


def build_positions(cu_seqlens: torch.Tensor, total_tokens: int) -> torch.Tensor:
    pos = torch.empty(total_tokens, dtype=torch.int32, device=cu_seqlens.device)
    _lib.check(_lib.lib().lrx_build_positions(_lib.ptr(cu_seqlens), cu_seqlens.numel() - 1, total_tokens, _lib.ptr(pos), _s()))
    return pos


def attn_work_list(cu_seqlens: torch.Tensor, total_tokens: int, max_seqlen: int, nq: int, nkv: int, d: int, last_tile_only: bool = False) -> torch.Tensor:
    """The work list of `attn_varlen_causal` for this batch layout (include/lrx.h, ABI 7): build once, pass to every layer's call."""
    L = _lib.lib()
    n_seqs = cu_seqlens.numel() - 1
    nbytes = L.lrx_attn_items_bytes(n_seqs, total_tokens, max_seqlen, nq, nkv, d, int(last_tile_only))
    items = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=cu_seqlens.device)
    _lib.check(L.lrx_attn_build_items(_lib.ptr(cu_seqlens), n_seqs, total_tokens, max_seqlen, nq, nkv, d, int(last_tile_only), _lib.ptr(items),
                                      items.numel(), _s()))
    return items


def attn_varlen_causal(qkv: torch.Tensor, cu_seqlens: torch.Tensor, max_seqlen: int, nq: int, nkv: int, d: int,
                       last_tile_only: bool = False, work_list=None) -> torch.Tensor:
    """work_list: None -> built here (one extra small launch); a tensor from `attn_work_list` with the SAME layout arguments; False -> the
    launch without a list (`lrx_attn_varlen_causal`: the kernel derives every item itself; same bits, ~10 % longer on the tiled path)."""
    if qkv.dtype != torch.float16:
        raise TypeError("attn_varlen_causal: qkv must be fp16 (the fused QKV + RoPE projection writes fp16)")
    T = qkv.shape[0]
    out = (torch.zeros if last_tile_only else torch.empty)(T, nq * d, dtype=torch.bfloat16, device=qkv.device)
    n_seqs = cu_seqlens.numel() - 1
    if work_list is False:
        _lib.check(_lib.lib().lrx_attn_varlen_causal(_lib.ptr(qkv), _lib.ptr(cu_seqlens), n_seqs, T, max_seqlen, nq, nkv, d,
                                                     _lib.ptr(out), int(last_tile_only), _s()))
        return out
    if T == 0 or n_seqs == 0:
        return out
    if work_list is None:
        work_list = attn_work_list(cu_seqlens, T, max_seqlen, nq, nkv, d, last_tile_only)
    _lib.check(_lib.lib().lrx_attn_varlen_causal_items(_lib.ptr(qkv), _lib.ptr(cu_seqlens), _lib.ptr(work_list), work_list.numel(), n_seqs, T, max_seqlen,
                                                       nq, nkv, d, _lib.ptr(out), int(last_tile_only), _s()))
    return out


def attn_prefix_suffix(qkv: torch.Tensor, prefix_kv: torch.Tensor, n_seqs: int, suffix_len: int, nq: int, nkv: int, d: int) -> torch.Tensor:
    """qkv [n_seqs*suffix_len, (nq+2nkv)d] fp16, prefix_kv [P1, 2*nkv*d] fp16 (k | v) -> [n_seqs*suffix_len, nq*d] bf16."""
    if qkv.dtype != torch.float16 or prefix_kv.dtype != torch.float16:
        raise TypeError("attn_prefix_suffix: qkv and prefix_kv must be fp16")
    if qkv.shape != (n_seqs * suffix_len, (nq + 2 * nkv) * d) or prefix_kv.shape[1:] != (2 * nkv * d,):
        raise ValueError("attn_prefix_suffix: operand shapes do not match the head layout")
    out = torch.empty(n_seqs * suffix_len, nq * d, dtype=torch.bfloat16, device=qkv.device)
    _lib.check(_lib.lib().lrx_attn_prefix_suffix(_lib.ptr(qkv), _lib.ptr(prefix_kv) if prefix_kv.shape[0] else None, n_seqs, suffix_len,
                                                 prefix_kv.shape[0], nq, nkv, d, _lib.ptr(out), _s()))
    return out


def uniform_layout(n_seqs: int, length: int, position_offset: int, device) -> tuple[torch.Tensor, torch.Tensor]:
    cu = torch.empty(n_seqs + 1, dtype=torch.int32, device=device)
    pos = torch.empty(n_seqs * length, dtype=torch.int32, device=device)
    _lib.check(_lib.lib().lrx_uniform_layout(_lib.ptr(cu), _lib.ptr(pos), n_seqs, length, position_offset, _s()))
    return cu, pos


def gather_last_rows(src: torch.Tensor, cu_seqlens: torch.Tensor) -> torch.Tensor:
    B = cu_seqlens.numel() - 1
    dst = torch.empty(B, src.shape[1], dtype=torch.bfloat16, device=src.device)
    _lib.check(_lib.lib().lrx_gather_last_rows(_lib.ptr(src), _lib.ptr(cu_seqlens), B, src.shape[1], _lib.ptr(dst), _s()))
    return dst


def pool_norm(hidden: torch.Tensor, w: torch.Tensor, cu_seqlens: torch.Tensor, eps: float, out_dim: Optional[int] = None,
              normalize: bool = True, pooling: str = "lasttoken") -> torch.Tensor:
    """hidden bf16 [T,H] (HF's bf16 norm arithmetic) or fp32 [T,H] (the precise stream: fp32 norm); pooling: finetune/dense_pooling.py:12-82
    ('lasttoken', 'cls', 'mean', 'second_to_last', 'third_to_last')."""
    B, H = cu_seqlens.numel() - 1, hidden.shape[1]
    D = out_dim or H
    out = torch.empty(B, D, dtype=torch.float32, device=hidden.device)
    _lib.check(_lib.lib().lrx_pool_norm_mode(_lib.ptr(hidden), _lib.ptr(w), _lib.ptr(cu_seqlens), B, H, eps, _lib.POOLING[pooling], _lib.ptr(out), D, D,
                                             int(normalize), None, 0, None, int(hidden.dtype == torch.float32), _s()))
    return out


def embedding_bag_mean(table: torch.Tensor, ids: torch.Tensor, offsets: torch.Tensor, padding_idx: Optional[int] = None,
                       out_dim: Optional[int] = None, normalize: bool = False) -> torch.Tensor:
    """table fp32 [V,H] (device), ids/offsets int64 (device) -> fp32 [n_bags, out_dim]."""
    V, H = table.shape
    D = out_dim or H
    nb = offsets.numel()
    out = torch.empty(nb, D, dtype=torch.float32, device=table.device)
    _lib.check(_lib.lib().lrx_embedding_bag_mean(_lib.ptr(table), V, H, _lib.ptr(ids), ids.numel(), _lib.ptr(offsets), nb,
                                                 -1 if padding_idx is None else int(padding_idx), _lib.ptr(out), D, D, int(normalize), _s()))
    return out


def flat_ip_scores(X: torch.Tensor, q: torch.Tensor) -> torch.Tensor:
    lib = _lib.lib()
    N, D = X.shape
    ld = int(lib.lrx_flat_ip_score_ld(N))
    s = torch.empty(q.shape[0], ld, dtype=torch.float32, device=X.device)
    _lib.check(lib.lrx_flat_ip_scores(_lib.ptr(X), N, X.stride(0), D, _lib.ptr(q), q.shape[0], _lib.ptr(s), _s()))
    return s


# -- sparse document vectors (N2) ------------------------------------------------------------------------------------------
def sparse_max_aggregate(hidden: torch.Tensor, lm_head: torch.Tensor, cu_seqlens: torch.Tensor, tok_mask: Optional[torch.Tensor] = None,
                         bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """hidden bf16 [T,H], lm_head bf16 [V,H], tok_mask uint8 [T] (None = drop first/last token) -> fp32 [B,V] running maxima."""
    T, H = hidden.shape
    V, B = lm_head.shape[0], cu_seqlens.numel() - 1
    if lm_head.shape[1] != H or hidden.dtype != torch.bfloat16 or lm_head.dtype != torch.bfloat16:
        raise ValueError("sparse_max_aggregate: hidden [T,H] / lm_head [V,H] must be bf16 with matching H")
    if tok_mask is not None and (tok_mask.dtype != torch.uint8 or tok_mask.numel() != T):
        raise ValueError("sparse_max_aggregate: tok_mask must be uint8 [T]")
    out = torch.empty(B, V, dtype=torch.float32, device=hidden.device)
    seg = torch.empty(T, dtype=torch.int32, device=hidden.device)
    _lib.check(_lib.lib().lrx_sparse_max_aggregate(_lib.ptr(hidden), _lib.ptr(lm_head), _lib.ptr(bias) if bias is not None else None,
                                                   _lib.ptr(cu_seqlens), _lib.ptr(tok_mask) if tok_mask is not None else None, B, T, H, V,
                                                   _lib.ptr(out), out.stride(0), _lib.ptr(seg), _s()))
    return out


def sparsify_(reps: torch.Tensor, relu: bool = True, log1p: bool = True, round_bf16: bool = True, top_k: int = 0, min_tokens_to_keep: int = 8) -> torch.Tensor:
    if reps.dtype != torch.float32 or reps.dim() != 2 or reps.stride(1) != 1:
        raise ValueError("sparsify_: fp32 [B,V] with unit inner stride")
    _lib.check(_lib.lib().lrx_sparsify(_lib.ptr(reps), reps.shape[0], reps.shape[1], reps.stride(0), int(relu), int(log1p), int(round_bf16),
                                       int(top_k), int(min_tokens_to_keep), _s()))
    return reps


def sparse_compact(reps: torch.Tensor, quantization_factor: int = 100, capacity: Optional[int] = None):
    """-> (ids int32 [B,cap], weights int32 [B,cap], counts int32 [B]); ascending token ids, weights = rint(max(x,0)*q) != 0."""
    B, V = reps.shape
    cap = int(capacity or V)
    ids = torch.empty(B, cap, dtype=torch.int32, device=reps.device)
    w = torch.empty(B, cap, dtype=torch.int32, device=reps.device)
    cnt = torch.empty(B, dtype=torch.int32, device=reps.device)
    _lib.check(_lib.lib().lrx_sparse_compact(_lib.ptr(reps), B, V, reps.stride(0), int(quantization_factor), cap, _lib.ptr(ids), _lib.ptr(w),
                                             _lib.ptr(cnt), _s()))
    return ids, w, cnt
