"""HBM-resident flat inner-product index shard: the faiss.IndexFlatIP surface the reference uses
(retriever/faiss_index.py:20-73: add / search / reset / ntotal), backed by lrx_flat_ip_search."""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib


class FlatIPIndex:
    def __init__(self, d: int, capacity: int = 0, device: Optional[torch.device] = None, id_base: int = 0):
        _lib.require_gpu()
        if d % 32 != 0:
            raise ValueError(f"FlatIPIndex: d={d} must be a multiple of 32")
        self.lib = _lib.lib()
        self.d = d
        self.device = device or torch.device("cuda", torch.cuda.current_device())
        self.ntotal = 0
        self.id_base = id_base  # added to local row numbers (global row of this shard's row 0)
        self._x = torch.empty(max(capacity, 0), d, dtype=torch.float32, device=self.device)
        self._ws = None
        # max |row| over the committed rows, kept on the device (no host sync): the error bound of the bf16 filter pass of
        # lrx_flat_ip_search_bounded scales with it.  two_pass = False forces the six-product path for every search.
        self._norm_bound = torch.zeros(1, dtype=torch.float32, device=self.device)
        self.two_pass = True
        # bf16 shadow of the rows (round-to-nearest-even), maintained by commit(): the filter pass of the two-pass search streams
        # it instead of the fp32 rows (half the bytes; the exact rescoring still reads fp32).  +50 % index memory; False = no shadow.
        self.shadow_bf16 = True
        self.max_workspace_bytes = 12 << 30  # search(): cap of the [queries, rows] score workspace; larger query batches are chunked
        self._xb: Optional[torch.Tensor] = None

    # -- storage -------------------------------------------------------------------------------------------------
    def reserve(self, n_rows: int):
        if n_rows > self._x.shape[0]:
            new = torch.empty(n_rows, self.d, dtype=torch.float32, device=self.device)
            if self.ntotal:
                new[:self.ntotal].copy_(self._x[:self.ntotal])
            self._x = new

    def append_slot(self, n_rows: int) -> torch.Tensor:
        """Rows [ntotal, ntotal+n) of the shard as a writable view (the encoder writes embeddings straight into it);
        call commit(n) afterwards."""
        if self.ntotal + n_rows > self._x.shape[0]:
            self.reserve(max(self.ntotal + n_rows, int(self._x.shape[0] * 1.5) + 1))
        return self._x[self.ntotal:self.ntotal + n_rows]

    def _shadow_rows(self, a: int, b: int):
        if not self.shadow_bf16 or self.d % 64 != 0:
            return
        if self._xb is None or self._xb.shape[0] < self._x.shape[0]:
            xb = torch.empty(self._x.shape[0], self.d, dtype=torch.bfloat16, device=self.device)
            if self._xb is not None and self.ntotal:
                xb[:self.ntotal].copy_(self._xb[:self.ntotal])
            self._xb = xb
        for s in range(a, b, 262144):
            e = min(s + 262144, b)
            self._xb[s:e].copy_(self._x[s:e])          # fp32 -> bf16, round-to-nearest-even

    def commit(self, n_rows: int):
        if n_rows > 0:
            new = self._x[self.ntotal:self.ntotal + n_rows]
            torch.maximum(self._norm_bound, torch.linalg.vector_norm(new, dim=1).max().reshape(1) * (1.0 + 1e-6), out=self._norm_bound)
            self._shadow_rows(self.ntotal, self.ntotal + n_rows)
        self.ntotal += n_rows

    def add(self, x):
        """faiss add(x f32[n,d]); accepts torch (any device) or numpy."""
        if not isinstance(x, torch.Tensor):
            x = torch.from_numpy(x)
        if x.ndim != 2 or x.shape[1] != self.d:
            raise ValueError(f"add: expected [n,{self.d}], got {tuple(x.shape)}")
        slot = self.append_slot(x.shape[0])
        slot.copy_(x.to(dtype=torch.float32))
        self.commit(x.shape[0])

    def refresh_norm_bound(self):
        """Recompute max |row| (and the bf16 shadow) over all committed rows: call after writing into committed rows in place."""
        self._shadow_rows(0, self.ntotal)
        self._norm_bound.zero_()
        for s in range(0, self.ntotal, 262144):
            e = min(s + 262144, self.ntotal)
            torch.maximum(self._norm_bound, torch.linalg.vector_norm(self._x[s:e], dim=1).max().reshape(1) * (1.0 + 1e-6), out=self._norm_bound)

    def reset(self):
        self.ntotal = 0
        self._norm_bound.zero_()

    # -- persistence (faiss.write_index / read_index of an IndexFlatIP, see index_io.py) --------------------------
    def save(self, fname: str, chunk_rows: int = 262144):
        from .index_io import write_flat_ip
        write_flat_ip(fname, (self._x[s:min(s + chunk_rows, self.ntotal)].cpu().numpy() for s in range(0, self.ntotal, chunk_rows)),
                      self.d, self.ntotal)

    @classmethod
    def load(cls, fname: str, device: Optional[torch.device] = None, id_base: int = 0, chunk_rows: int = 262144) -> "FlatIPIndex":
        from .index_io import read_flat_ip
        import numpy as np
        mm = read_flat_ip(fname)
        idx = cls(mm.shape[1], capacity=mm.shape[0], device=device, id_base=id_base)
        for s in range(0, mm.shape[0], chunk_rows):
            e = min(s + chunk_rows, mm.shape[0])
            idx._x[s:e].copy_(torch.from_numpy(np.array(mm[s:e], copy=True)), non_blocking=False)
        idx.commit(mm.shape[0])
        return idx

    @property
    def vectors(self) -> torch.Tensor:
        return self._x[:self.ntotal]

    # -- search --------------------------------------------------------------------------------------------------
    def search(self, q, k: int):
        """-> (D f32[Q,k], I i64[Q,k]) device tensors, descending scores, ids = id_base + row, ties -> lower id,
        (-FLT_MAX, -1) padding when k > ntotal."""
        if not isinstance(q, torch.Tensor):
            q = torch.from_numpy(q)
        q = q.to(device=self.device, dtype=torch.float32).contiguous()
        if q.ndim != 2 or q.shape[1] != self.d:
            raise ValueError(f"search: expected [Q,{self.d}], got {tuple(q.shape)}")
        Q = q.shape[0]
        D = torch.empty(Q, k, dtype=torch.float32, device=self.device)
        I = torch.empty(Q, k, dtype=torch.int64, device=self.device)
        if Q == 0:
            return D, I
        # the score workspace is [queries, rows] fp32: large query batches over a large shard go through in chunks of queries that keep
        # it under max_workspace_bytes (multiples of the 256-query filter pass; results do not depend on the chunking)
        per_query = max(1, int(self.lib.lrx_flat_ip_bounded_workspace_bytes(self.ntotal, self.d, 2, k))
                        - int(self.lib.lrx_flat_ip_bounded_workspace_bytes(self.ntotal, self.d, 1, k)))
        chunk = max(1, min(Q, int(self.max_workspace_bytes) // per_query))
        if chunk < Q:
            chunk = chunk // 256 * 256 if chunk >= 256 else (128 if chunk >= 128 else chunk)
        need = int(self.lib.lrx_flat_ip_bounded_workspace_bytes(self.ntotal, self.d, chunk, k))
        if self._ws is None or self._ws.numel() < need:
            self._ws = None
            self._ws = torch.zeros(need, dtype=torch.uint8, device=self.device)
        ldx = self._x.stride(0) if self._x.shape[0] else self.d
        xb = self._xb if (self.two_pass and self.shadow_bf16 and self._xb is not None) else None
        for s in range(0, Q, chunk):
            qc, Dc, Ic = q[s:s + chunk], D[s:s + chunk], I[s:s + chunk]
            if self.two_pass:
                _lib.check(self.lib.lrx_flat_ip_search_bounded(_lib.ptr(self._x), self.ntotal, ldx, self.d, _lib.ptr(xb) if xb is not None else None,
                                                               xb.stride(0) if xb is not None else 0, _lib.ptr(self._norm_bound), _lib.ptr(qc),
                                                               qc.shape[0], k, self.id_base, _lib.ptr(Dc), _lib.ptr(Ic), _lib.ptr(self._ws),
                                                               self._ws.numel(), _lib.current_stream()))
            else:
                _lib.check(self.lib.lrx_flat_ip_search(_lib.ptr(self._x), self.ntotal, ldx, self.d, _lib.ptr(qc), qc.shape[0], k, self.id_base,
                                                       _lib.ptr(Dc), _lib.ptr(Ic), _lib.ptr(self._ws), self._ws.numel(), _lib.current_stream()))
        return D, I


def merge_topk(D_parts: torch.Tensor, I_parts: torch.Tensor):
    """[R,Q,k] per-shard lists -> ([Q,k], [Q,k]) with the same ordering rule (score desc, id asc)."""
    lib = _lib.lib()
    R, Q, k = D_parts.shape
    D_parts, I_parts = D_parts.contiguous(), I_parts.contiguous()
    D = torch.empty(Q, k, dtype=torch.float32, device=D_parts.device)
    I = torch.empty(Q, k, dtype=torch.int64, device=D_parts.device)
    _lib.check(lib.lrx_merge_topk(_lib.ptr(D_parts), _lib.ptr(I_parts), R, Q, k, _lib.ptr(D), _lib.ptr(I), _lib.current_stream()))
    return D, I
