"""Row-sharded corpus across the GPUs of one node: one process per GPU (torchrun), no collective on the encode path,
one small all-gather of per-shard top-k per search call, merge on device.

Replaces the reference's GPU mode of Faiss (retriever/faiss_index.py:60-70, index_cpu_to_all_gpus(shard=True): rows split
across GPUs, per-shard top-k merged on the host) and its RPC fan-out of batches (inference/exact_search_torchrpc.py:243-328):
here batch j of the longest-first sorted corpus is owned by rank j % R and its embeddings never leave that rank's HBM."""
from __future__ import annotations

import os
from typing import Optional

import torch
import torch.distributed as dist


def batches_for_rank(n_docs: int, batch_size: int, rank: int, world: int) -> list[tuple[int, int]]:
    """Static interleaved assignment: batch j = docs [j*bs, (j+1)*bs) of the sorted corpus goes to rank j % world."""
    out = []
    for j, s in enumerate(range(0, n_docs, batch_size)):
        if j % world == rank:
            out.append((s, min(s + batch_size, n_docs)))
    return out


def local_to_global_rows(n_docs: int, batch_size: int, rank: int, world: int) -> torch.Tensor:
    """global (sorted-order) row of every local shard row, int64, in local insertion order."""
    parts = [torch.arange(s, e, dtype=torch.int64) for s, e in batches_for_rank(n_docs, batch_size, rank, world)]
    return torch.cat(parts) if parts else torch.zeros(0, dtype=torch.int64)


def pack_pairs(D: torch.Tensor, I: torch.Tensor) -> torch.Tensor:
    """(score f32, global row i64 < 2^32 - 1 or -1) -> one int64 per hit: score bits in the high word, row in the low word.  Torch form of
    lrx_pack_topk for host tensors (tests); raises on rows the 32-bit field cannot carry instead of truncating them."""
    if I.numel() and int(I.max()) >= 2 ** 32 - 1:
        raise ValueError("pack_pairs: global row ids must be < 2^32 - 1")
    return (D.contiguous().view(torch.int32).to(torch.int64) << 32) | (I & 0xFFFFFFFF)


def unpack_pairs(P: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
    D = (P >> 32).to(torch.int32).view(torch.float32)
    I = (P & 0xFFFFFFFF).to(torch.int64)
    I = torch.where(I >= 0x80000000, I - 0x100000000, I)   # sign-extend the -1 sentinel
    return D, I


def exchange_topk(D: torch.Tensor, I: torch.Tensor, group: Optional[dist.ProcessGroup] = None,
                  force_collective: bool = False) -> tuple[torch.Tensor, torch.Tensor]:
    """all-gather the packed [Q,k] lists of every rank -> ([R,Q,k] scores, [R,Q,k] global rows) on every rank, UNMERGED (host-side
    tests and tools; the product path is exchange_merge, which keeps the wire words on the device end to end).
    Payload R*Q*k*8 bytes (640 KB at R=8,Q=100,k=100): latency-bound on xGMI, one collective, no pipelining."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    force_collective = force_collective or os.environ.get("LRX_FORCE_COLLECTIVE") == "1"     # one-rank rehearsal of the exchange
    if world == 1 and not (force_collective and dist.is_initialized()):
        return D.unsqueeze(0), I.unsqueeze(0)
    return unpack_pairs(_all_gather_words(pack_pairs(D, I), world, group))


def _all_gather_words(packed: torch.Tensor, world: int, group) -> torch.Tensor:
    """[Q, k] int64 wire words of this rank -> [R, Q, k] on the same device (one all_gather_into_tensor)."""
    Q = packed.shape[0]
    dev = packed.device
    if packed.is_cuda and dist.get_backend(group) == "gloo":      # CPU-side process group (tests, debugging): stage the 8-byte pairs through the host
        packed = packed.cpu()
    out = torch.empty((world * Q,) + tuple(packed.shape[1:]), dtype=torch.int64, device=packed.device)   # concat form: nccl + gloo
    dist.all_gather_into_tensor(out, packed.contiguous(), group=group)
    return out.view((world, Q) + tuple(packed.shape[1:])).to(dev)


def _collective(group, force_collective: bool) -> bool:
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    force_collective = force_collective or os.environ.get("LRX_FORCE_COLLECTIVE") == "1"
    return world > 1 or (force_collective and dist.is_initialized())


def exchange_merge(D: torch.Tensor, I: torch.Tensor, group: Optional[dist.ProcessGroup] = None, row_map: Optional[torch.Tensor] = None,
                   id_base: int = 0, force_collective: bool = False, words: Optional[torch.Tensor] = None,
                   on_stage=None) -> tuple[torch.Tensor, torch.Tensor]:
    """This shard's device-resident (scores f32 [Q,k], ids i64 [Q,k]) -> the global top-k on every rank:
    lrx_pack_topk (row map applied, one 64-bit word per hit) -> ONE RCCL all-gather -> lrx_merge_topk_packed.  No torch kernels.
    words: the wire words when the search has already written them (FlatIPIndex.search(wire_out=...)): the packing launch is skipped.
    on_stage (measurement aid, bench.py): called with "gathered" / "merged" on the caller's stream right after the all-gather / the merge
    were enqueued -- HIP events recorded there split a pass into local search, exchange and merge."""
    from . import _lib
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    force_collective = force_collective or os.environ.get("LRX_FORCE_COLLECTIVE") == "1"
    collective = world > 1 or (force_collective and dist.is_initialized())
    if not collective and row_map is None:
        return D, I
    lib = _lib.lib()
    Q, k = D.shape
    D, I = D.contiguous(), I.contiguous()
    if words is None:
        words = torch.empty(Q, k, dtype=torch.int64, device=D.device)
        _lib.check(lib.lrx_pack_topk(_lib.ptr(D), _lib.ptr(I), _lib.ptr(row_map), int(id_base), Q * k, _lib.ptr(words), _lib.current_stream()))
    allw = _all_gather_words(words, world, group) if collective else words.unsqueeze(0)
    if on_stage is not None:
        on_stage("gathered")
    Dm = torch.empty(Q, k, dtype=torch.float32, device=D.device)
    Im = torch.empty(Q, k, dtype=torch.int64, device=D.device)
    _lib.check(lib.lrx_merge_topk_packed(_lib.ptr(allw), allw.shape[0], Q, k, _lib.ptr(Dm), _lib.ptr(Im), _lib.current_stream()))
    if on_stage is not None:
        on_stage("merged")
    return Dm, Im


class ShardedFlatIPIndex:
    """This rank's FlatIPIndex shard + the exchange.  `row_map` (optional) maps local rows to global rows; without it the
    shard covers the contiguous global range starting at id_base."""

    def __init__(self, shard, row_map: Optional[torch.Tensor] = None, group=None):
        self.shard = shard
        self.row_map = None if row_map is None else row_map.to(device=shard.device, dtype=torch.int64).contiguous()
        self.group = group
        # the wire word carries the global row in 32 bits (0xFFFFFFFF = none)
        top = int(self.row_map.max()) if (self.row_map is not None and self.row_map.numel()) else shard.id_base + max(shard._x.shape[0], shard.ntotal)
        if top >= 2 ** 32 - 1:
            raise ValueError(f"ShardedFlatIPIndex: global row {top} does not fit the 32-bit field of the exchange word")

    def search(self, q: torch.Tensor, k: int, lane: int = 0):
        return self.finish(*self.local_search(q, k, lane=lane))

    def local_search(self, q: torch.Tensor, k: int, lane: int = 0):
        """This rank's part: (scores, ids, wire words or None).  When an exchange will follow, the search's last kernel writes the wire
        words itself (no packing launch between the local search and the all-gather)."""
        if not (_collective(self.group, False) or self.row_map is not None):
            return (*self.shard.search(q, k, lane=lane), None)
        words = torch.empty(q.shape[0], k, dtype=torch.int64, device=self.shard.device)
        D, I = self.shard.search(q, k, wire_out=words, row_map=self.row_map, lane=lane)
        return D, I, words

    def finish(self, D: torch.Tensor, I: torch.Tensor, words: Optional[torch.Tensor] = None, on_stage=None):
        """local (scores, ids) of this shard -> global top-k on every rank (row map, all-gather, on-device merge)."""
        return exchange_merge(D, I, self.group, row_map=self.row_map, id_base=self.shard.id_base, words=words, on_stage=on_stage)
