"""Hit-list fusion on the GPU: `fuse_scores_rrf` / `fuse_scores_linear` with the reference's names, arguments and
dict-of-dicts results (retriever/score_fuse_utils.py:3-91), plus the array form `fuse_hits` the searchers use so that fused
results never become Python dicts before the final hand-over.  Arithmetic is IEEE double on the device, documents'
contributions are summed in system order: scores are bit-identical to the reference's numpy float64 results."""
from __future__ import annotations

from typing import Optional, Sequence

import torch

from . import _lib


def _s():
    return _lib.current_stream()


def fuse_hits(systems: Sequence[tuple[torch.Tensor, torch.Tensor]], method: str = "rrf", k: int = 60, weights: Optional[Sequence[float]] = None,
              eps: float = 1e-8):
    """systems: [(scores [Q, k_i] float, ids [Q, k_i] int64 with -1 = empty slot)], same Q, on the GPU.
    -> (fused scores f64 [Q, sum k_i], ids i64 [Q, sum k_i], counts i32 [Q]); rows sorted by fused score, -inf / -1 padding."""
    if method not in ("rrf", "linear"):
        raise NotImplementedError(f"score_fuse_method {method} is not supported.")
    if method == "linear":
        assert weights is not None and len(weights) == len(systems)
    lib = _lib.lib()
    Q = systems[0][0].shape[0]
    dev = systems[0][0].device
    ids_all, con_all = [], []
    for si, (sc, ids) in enumerate(systems):
        if sc.shape != ids.shape or sc.shape[0] != Q:
            raise ValueError("fuse_hits: every system needs scores/ids of the same [Q, k] shape")
        sc = sc.to(dev, dtype=torch.float64).contiguous()
        ids = ids.to(dev, dtype=torch.int64).contiguous()
        con = torch.empty_like(sc)
        p0, p1 = (float(k), 0.0) if method == "rrf" else (float(weights[si]), float(eps))
        _lib.check(lib.lrx_hit_contributions(_lib.ptr(sc), _lib.ptr(ids), Q, sc.shape[1], sc.stride(0), 0 if method == "rrf" else 1, p0, p1,
                                             _lib.ptr(con), con.stride(0), _s()))
        ids_all.append(ids)
        con_all.append(con)
    ids_cat, con_cat = torch.cat(ids_all, 1).contiguous(), torch.cat(con_all, 1).contiguous()
    n = ids_cat.shape[1]
    out_s = torch.empty(Q, n, dtype=torch.float64, device=dev)
    out_i = torch.empty(Q, n, dtype=torch.int64, device=dev)
    cnt = torch.empty(Q, dtype=torch.int32, device=dev)
    _lib.check(lib.lrx_hit_union(_lib.ptr(ids_cat), _lib.ptr(con_cat), Q, n, ids_cat.stride(0), _lib.ptr(out_s), _lib.ptr(out_i), _lib.ptr(cnt), _s()))
    return out_s, out_i, cnt


def _dicts_to_arrays(results_list):
    qids: dict[str, int] = {}
    pids: dict[str, int] = {}
    for res in results_list:
        for q, passages in res.items():
            qids.setdefault(str(q), len(qids))
            for p in passages:
                pids.setdefault(str(p), len(pids))
    systems = []
    for res in results_list:
        kmax = max([len(v) for v in res.values()] + [1])
        sc = torch.zeros(len(qids), kmax, dtype=torch.float64)
        ids = torch.full((len(qids), kmax), -1, dtype=torch.int64)
        for q, passages in res.items():
            r = qids[str(q)]
            if passages:
                sc[r, :len(passages)] = torch.tensor([float(v) for v in passages.values()], dtype=torch.float64)
                ids[r, :len(passages)] = torch.tensor([pids[str(p)] for p in passages], dtype=torch.int64)
        systems.append((sc, ids))
    return list(qids), list(pids), systems


def _fuse_dicts(results_list, **kw):
    _lib.require_gpu()
    qids, pids, systems = _dicts_to_arrays(results_list)
    if not qids:
        return {}
    dev = torch.device("cuda", torch.cuda.current_device())
    sc, ids, cnt = fuse_hits([(s.to(dev), i.to(dev)) for s, i in systems], **kw)
    sc, ids, cnt = sc.cpu().tolist(), ids.cpu().tolist(), cnt.cpu().tolist()
    return {q: {pids[ids[r][j]]: sc[r][j] for j in range(cnt[r])} for r, q in enumerate(qids)}


def fuse_scores_rrf(results_list: list[dict], k: int = 60) -> dict:
    """Reciprocal Rank Fusion (retriever/score_fuse_utils.py:3-45): dict[query_id -> dict[passage_id -> sum 1/(k + rank)]]."""
    return _fuse_dicts(results_list, method="rrf", k=k)


def fuse_scores_linear(results_list: list[dict], weights: Sequence[float] = (0.7, 0.3), eps: float = 1e-8) -> dict:
    """Min-max normalise each system's scores per query, weight, add (retriever/score_fuse_utils.py:47-91)."""
    assert len(results_list) == len(weights)
    return _fuse_dicts(results_list, method="linear", weights=list(weights), eps=eps)
