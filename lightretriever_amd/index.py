"""HBM-resident flat inner-product index shard: the faiss.IndexFlatIP surface the reference uses
(retriever/faiss_index.py:20-73: add / search / reset / ntotal), backed by lrx_flat_ip_search."""
from __future__ import annotations

import ctypes as C
import weakref
from typing import Optional

import torch

from . import _lib


_SHARDS = weakref.WeakValueDictionary()   # fp32 storage pointer -> FlatIPIndex (lets the encoder recognise a shard slot it writes into)


def shard_of(out: torch.Tensor):
    """(index, first_row) when `out` is a view of whole rows of a live FlatIPIndex's fp32 storage, else None.  LrxEncoder.encode_packed
    uses it to hand the shard's fp16 shadow rows and bounds to the encoder's last kernel (lrx_encode_packed_shard)."""
    if not _SHARDS or out.dtype != torch.float32 or out.ndim != 2:
        return None
    idx = _SHARDS.get(out.untyped_storage().data_ptr())
    if idx is None or out.shape[1] != idx.d or out.stride(0) != idx._x.stride(0) or out.stride(1) != 1:
        return None
    off = out.data_ptr() - idx._x.data_ptr()
    row_bytes = idx._x.stride(0) * 4
    if off < 0 or off % row_bytes:
        return None
    return idx, off // row_bytes


class FlatIPIndex:
    """One HBM-resident shard.  Memory: 4 B/element fp32 rows + 2 B/element tiled fp16 shadow + the search workspace -- per lane in use,
    for a chunk of up to 256 queries (up to 1024 where the main pass runs on the GEMM kernel: shadow, d >= 1024, more than 256 queries in the
    call): candidate lists of max(64 Ki, 64 k rounded up to a power of two) 8-byte entries per query (512 KiB per
    query up to k = 1024: 134 MB per 256-query chunk, 0.5 GB per 1024), the compact sample scores and the [128, ntotal] fp32 region of the gated fallback;
    `lrx_flat_ip_bounded_workspace_bytes` is the exact figure.  A search of MORE queries than one library chunk forks its chunks over
    `chunk_lanes` (2) internal HIP streams with one workspace each (so up to 2 x the figure above) unless that would exceed
    `max_workspace_bytes`, in which case it runs the chunks one after the other on the caller's stream.  Lanes != 0 (pipeline.SearchLanes)
    add one workspace each.  NOT thread-safe: an index keeps search state (workspaces, internal streams, the statistics of the last search);
    two host threads must not call search() on the same index at the same time (concurrent searches from ONE thread go through
    SearchLanes, one lane per search in flight)."""
    # lrx_flat_ip_search_bounded flags (_lib.SEARCH_FILTER_*): which filter the bounded search runs.  A class-level default that tests and A/B
    # tools override (per index or for all); the hits do not depend on it.  (Round 2 had a process-global switch inside the library.)
    search_flags = _lib.SEARCH_FILTER_AUTO

    def __init__(self, d: int, capacity: int = 0, device: Optional[torch.device] = None, id_base: int = 0):
        _lib.require_gpu()
        if d % 32 != 0:
            raise ValueError(f"FlatIPIndex: d={d} must be a multiple of 32")
        self.lib = _lib.lib()
        self.d = d
        self.device = device or torch.device("cuda", torch.cuda.current_device())
        self.ntotal = 0
        self.id_base = id_base  # added to local row numbers (global row of this shard's row 0)
        self._ws = None                      # search workspace of lane 0 (the only one unless searches are pipelined over lanes)
        self._lane_ws: dict = {}             # lane != 0 -> its own workspace: searches in flight on different HIP streams must not share one
        self.chunk_lanes = 2                 # a search of more queries than one library chunk (256) alternates its chunks over this many internal streams
        self._chunk_streams = None
        self._last_search = None             # what last_list_counts() needs to find the statistics of the last two-pass search
        # {max |row|, max |row - fp16(row)|} over the committed rows, kept on the device (no host sync): the error bound of the fp16
        # filter pass of lrx_flat_ip_search_bounded is built from them.  two_pass = False forces the six-product path for every search.
        self._bounds = torch.zeros(2, dtype=torch.float32, device=self.device)
        self.two_pass = True
        # fp16 shadow of the rows (round-to-nearest-even, saturating): the filter pass of the two-pass search streams it instead of the fp32
        # rows (half the bytes; the exact rescoring still reads fp32).  +50 % index memory; False = no shadow.  Shadow rows and bounds are
        # written by the kernel that produces the fp32 rows (the encoder's last kernel for slots, lrx_shard_commit_rows for add()).
        # Layout: [128-row block][64-wide k-slice] tiles of 16 KiB, fragment-major inside (include/lrx.h): a wave of the filter pass loads
        # its MFMA operand with one coalesced 1-KiB request, straight into registers.
        self.shadow_f16 = True
        self.max_workspace_bytes = 12 << 30  # search(): cap of the search workspace; larger query batches are chunked
        self._xb: Optional[torch.Tensor] = None
        self._shadow_rows = 0                # committed rows [0, _shadow_rows) have valid shadow rows (== ntotal while shadow_f16 stays on)
        self._fused: list = []               # row intervals whose shadow + bounds the encoder has already written
        self._x = torch.empty(0, d, dtype=torch.float32, device=self.device)
        self._set_storage(torch.empty(max(capacity, 0), d, dtype=torch.float32, device=self.device))

    @property
    def _norm_bound(self) -> torch.Tensor:
        return self._bounds[:1]

    def _wants_shadow(self) -> bool:
        return self.shadow_f16 and self.d % 64 == 0

    def _set_storage(self, x: torch.Tensor):
        if self._x.numel():
            _SHARDS.pop(self._x.untyped_storage().data_ptr(), None)
        self._x = x
        if x.numel():
            _SHARDS[x.untyped_storage().data_ptr()] = self

    # -- storage -------------------------------------------------------------------------------------------------
    def reserve(self, n_rows: int):
        if n_rows > self._x.shape[0]:
            new = torch.empty(n_rows, self.d, dtype=torch.float32, device=self.device)
            if self.ntotal:
                new[:self.ntotal].copy_(self._x[:self.ntotal])
            self._set_storage(new)
        self._ensure_shadow()

    def _ensure_shadow(self):
        if not self._wants_shadow():
            return
        cap = self._x.shape[0]
        need = -(-cap // 128) * 128 * self.d                           # whole 128-row blocks, flat
        if self._xb is None or self._xb.numel() < need:
            xb = torch.empty(need, dtype=torch.float16, device=self.device)   # (padding rows of the last block are masked by the kernels)
            if self._xb is not None and self._shadow_rows:
                n_old = min(self._xb.numel(), -(-self._shadow_rows // 128) * 128 * self.d)   # the blocks that hold shadowed rows
                xb[:n_old].copy_(self._xb[:n_old])
            else:
                self._shadow_rows = 0
            self._xb = xb
        if self._shadow_rows < self.ntotal:
            # committed rows without a shadow (shadow_f16 switched on after rows were added, or switched off for a while and on again:
            # commits made meanwhile maintained the bounds only): build their shadow from the fp32 rows before anything streams it
            a, self._shadow_rows = self._shadow_rows, self.ntotal
            self._maintain(a, self.ntotal)

    def shadow_rows(self, n: Optional[int] = None) -> torch.Tensor:
        """The shadow as a row-major [n, d] fp16 tensor (a copy: the stored layout is tiled): tests and tools."""
        n = self.ntotal if n is None else n
        if self._xb is None:
            raise ValueError("this index keeps no fp16 shadow")
        nb = -(-n // 128)
        # tile = [16-row group w][k-step ks][fq][fi][8]  ->  row 128 b + 16 w + fi, column 64 s + 32 ks + 8 fq + j
        t = self._xb[:nb * 128 * self.d].view(nb, self.d // 64, 8, 2, 4, 16, 8)          # b, s, w, ks, fq, fi, j
        return t.permute(0, 2, 5, 1, 3, 4, 6).reshape(nb * 128, self.d)[:n]

    def append_slot(self, n_rows: int) -> torch.Tensor:
        """Rows [ntotal, ntotal+n) of the shard as a writable view (the encoder writes embeddings straight into it, together with
        their shadow rows and the bounds); call commit(n) afterwards."""
        if self.ntotal + n_rows > self._x.shape[0]:
            self.reserve(max(self.ntotal + n_rows, int(self._x.shape[0] * 1.5) + 1))
        self._ensure_shadow()
        # whoever receives these rows may write them with anything: an earlier encoder write into them no longer vouches for
        # their shadow / bounds (commit() maintains whatever is not re-recorded by shard_sink() after this point)
        a, b = self.ntotal, self.ntotal + n_rows
        self._fused = [iv for s, e in self._fused for iv in ((s, min(e, a)), (max(s, b), e)) if iv[1] > iv[0]]
        return self._x[a:b]

    def shard_sink(self, row0: int, n_rows: int):
        """(tiled shadow tensor or None, first shadow row, bounds) for rows [row0, row0 + n) and a note that their producer maintains them."""
        self._ensure_shadow()
        if row0 + n_rows > self.ntotal:               # (rows already committed need no bookkeeping: their producer keeps them valid)
            self._fused.append((row0, row0 + n_rows))
        if not (self._wants_shadow() and self._xb is not None):
            return None, 0, self._bounds
        return self._xb, row0, self._bounds

    def _maintain(self, a: int, b: int):
        """Shadow + bounds of rows [a, b) by lrx_shard_commit_rows (one read of the fp32 rows)."""
        if b <= a:
            return
        self._ensure_shadow()
        xb = self._xb if self._wants_shadow() else None
        _lib.check(self.lib.lrx_shard_commit_rows(_lib.ptr(self._x[a:]), self._x.stride(0), b - a, self.d, _lib.ptr(xb), a, _lib.ptr(self._bounds),
                                                  _lib.current_stream()))

    def commit(self, n_rows: int):
        if n_rows > 0:
            a, b = self.ntotal, self.ntotal + n_rows
            # rows the encoder wrote through shard_sink() are done; anything else in [a, b) gets its shadow + bounds now
            pos = a
            for s, e in sorted(self._fused):
                s, e = max(s, a), min(e, b)
                if e <= pos:
                    continue
                self._maintain(pos, min(s, b))
                pos = max(pos, e)
            self._maintain(pos, b)
        # an interval vouches for ONE commit: rows beyond b that are handed out again (append_slot) or written by something else are
        # maintained by the commit that covers them
        self._fused = []
        self.ntotal += n_rows
        if self._wants_shadow() and self._xb is not None and self._shadow_rows >= self.ntotal - n_rows:
            self._shadow_rows = self.ntotal

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
        """Recompute the bounds and the fp16 shadow over all committed rows: needed only after writing into committed rows in place
        with something other than the encoder (which maintains both itself)."""
        self._bounds.zero_()
        self._maintain(0, self.ntotal)

    def reset(self):
        self.ntotal = 0
        self._shadow_rows = 0
        self._bounds.zero_()
        self._fused = []

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
    def _lane_workspace(self, lane: int, need: int, user_stream: Optional["torch.cuda.Stream"] = None) -> torch.Tensor:
        """The workspace of `lane`, grown to `need` bytes.  user_stream: the stream its kernels will run on when that is not the stream
        current now (the internal chunk streams): the caching allocator is told, so that a later regrow / free cannot hand the block to the
        caller's stream while kernels of that stream still use it."""
        ws = self._ws if lane == 0 else self._lane_ws.get(lane)
        if ws is None or ws.numel() < need:
            if torch.cuda.is_current_stream_capturing():
                # an allocation made under HIP-graph capture lives in the graph's private pool: the index would keep pointing at memory that
                # goes back to the allocator with the graph (the memory-access fault of round 2's capture probe)
                raise _lib.LrxError("FlatIPIndex.search under graph capture: the search workspace must exist before the capture starts -- "
                                    "run one eager search with the same number of queries and k first")
            ws = None
            if lane == 0:
                self._ws = None
                ws = self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
            else:
                self._lane_ws.pop(lane, None)
                ws = self._lane_ws[lane] = torch.empty(need, dtype=torch.uint8, device=self.device)
        if user_stream is not None and user_stream != torch.cuda.current_stream():
            ws.record_stream(user_stream)
        return ws

    def search(self, q, k: int, wire_out: Optional[torch.Tensor] = None, row_map: Optional[torch.Tensor] = None, lane: int = 0):
        """-> (D f32[Q,k], I i64[Q,k]) device tensors, descending scores, ids = id_base + row, ties -> lower id,
        (-FLT_MAX, -1) padding when k > ntotal.  wire_out (int64 [Q,k], optional): also filled with the exchange words of a row-sharded
        search (lrx_pack_topk's format, `row_map` applied) by the last kernel of the search itself.  lane: which of the index's search
        workspaces to use -- searches that may be in flight at the same time (different HIP streams: pipeline.SearchLanes) take different lanes."""
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
        if wire_out is not None and not (wire_out.is_cuda and wire_out.dtype == torch.int64 and wire_out.is_contiguous() and tuple(wire_out.shape) == (Q, k)):
            raise ValueError("wire_out must be a contiguous int64 CUDA tensor [Q, k]")
        # the bounded search needs a workspace that stops growing at 256 queries; the six-product path (two_pass = False) a
        # [queries, rows] fp32 score matrix.  Either way the queries go through in chunks that keep it under max_workspace_bytes
        # (results do not depend on the chunking).
        flags = int(self.search_flags)
        ws_bytes = (lambda n, d, nq, kk: self.lib.lrx_flat_ip_bounded_workspace_bytes(n, d, nq, kk, flags)) if self.two_pass else self.lib.lrx_flat_ip_workspace_bytes
        chunk = Q
        while chunk > 1 and int(ws_bytes(self.ntotal, self.d, chunk, k)) > int(self.max_workspace_bytes):
            chunk = 256 if chunk > 256 else (128 if chunk > 128 else chunk // 2)
        ldx = self._x.stride(0) if self._x.shape[0] else self.d
        if self.two_pass and self.shadow_f16:
            self._ensure_shadow()                       # (no-op unless rows were committed while the shadow was switched off)
        xb = self._xb if (self.two_pass and self._wants_shadow() and self._xb is not None and self._shadow_rows >= self.ntotal) else None
        # the library walks a call's queries in chunks of this size: 256 over the shadow (128 without) or, where its main pass runs on the GEMM
        # kernel (D >= 1024), ONE pass over the shadow per up to 1024 queries (lrx_flat_ip_bounded_chunk_queries, round 6)
        has_xb = xb is not None and self.d % 64 == 0
        lib_chunk_of = lambda n: int(self.lib.lrx_flat_ip_bounded_chunk_queries(self.ntotal, self.d, n, k, flags, int(has_xb))) if self.two_pass else 128
        lib_chunk = lib_chunk_of(min(Q, chunk))
        # More queries than one library chunk: the chunks are independent searches over the same rows, so they alternate between two
        # internal HIP streams (own workspaces), forked from and joined back into the caller's stream inside this call -- the short
        # latency-bound kernels that frame one chunk's passes overlap the other chunk's passes (Q = 1000, top-1000 over a 100 k-row
        # chunk, the reference's operating point: 2.45 -> ~2.2 ms; the stream semantics of the call do not change).  Not under graph capture.
        fork = (self.two_pass and self.chunk_lanes > 1 and Q > lib_chunk and lane == 0 and not torch.cuda.is_current_stream_capturing())
        if fork and int(ws_bytes(self.ntotal, self.d, min(chunk, lib_chunk), k)) * self.chunk_lanes > int(self.max_workspace_bytes):
            fork = False                               # one workspace per internal stream would exceed the cap: chunks one after the other
        if fork:
            chunk = min(chunk, lib_chunk)
        need = int(ws_bytes(self.ntotal, self.d, chunk, k))
        cur = torch.cuda.current_stream()
        if fork:
            if self._chunk_streams is None:
                self._chunk_streams = [torch.cuda.Stream(device=self.device) for _ in range(self.chunk_lanes)]
            # (lane 0's own workspace serves the first internal stream, lanes -1, -2, ... the others)
            lane_ws = [self._lane_workspace(-j, need, user_stream=self._chunk_streams[j]) for j in range(self.chunk_lanes)]
            ws = lane_ws[0]
            start = torch.cuda.Event()
            start.record(cur)
        else:
            ws = self._lane_workspace(lane, need)
        try:
            self._run_chunks(q, D, I, k, chunk, fork, lane_ws if fork else None, ws, start if fork else None, xb, ldx, flags, row_map, wire_out)
        finally:
            if fork:                                   # the side streams are joined back whatever happened in the loop
                for st in self._chunk_streams:
                    cur.wait_stream(st)
        if wire_out is not None and not self.two_pass:
            _lib.check(self.lib.lrx_pack_topk(_lib.ptr(D), _lib.ptr(I), _lib.ptr(row_map), int(self.id_base), Q * k, _lib.ptr(wire_out), _lib.current_stream()))
        # (nq of the last library chunk of the last host chunk, the ntotal / mode the workspace was planned for, and the workspace itself)
        last_ws = lane_ws[((Q - 1) // chunk) % self.chunk_lanes] if fork else ws
        n_last_call = (Q - 1) % chunk + 1                                    # queries of the last library call ...
        self._last_search = ((n_last_call - 1) % lib_chunk_of(n_last_call) + 1, k, flags, xb is not None, last_ws, self.ntotal, bool(self.two_pass))   # ... and of its last chunk
        return D, I

    def _run_chunks(self, q, D, I, k, chunk, fork, lane_ws, ws, start, xb, ldx, flags, row_map, wire_out):
        Q = q.shape[0]
        for j, s in enumerate(range(0, Q, chunk)):
            qc, Dc, Ic = q[s:s + chunk], D[s:s + chunk], I[s:s + chunk]
            if fork:
                st = self._chunk_streams[j % self.chunk_lanes]
                if j < self.chunk_lanes:
                    st.wait_event(start)
                ws = lane_ws[j % self.chunk_lanes]
                stream = C.c_void_p(st.cuda_stream)
            else:
                stream = _lib.current_stream()
            if self.two_pass:
                _lib.check(self.lib.lrx_flat_ip_search_bounded_wire(
                    _lib.ptr(self._x), self.ntotal, ldx, self.d, _lib.ptr(xb), _lib.ptr(self._bounds), _lib.ptr(qc), qc.shape[0], k, self.id_base,
                    _lib.ptr(Dc), _lib.ptr(Ic), _lib.ptr(row_map), _lib.ptr(wire_out[s:s + chunk]) if wire_out is not None else None,
                    _lib.ptr(ws), ws.numel(), flags, stream))
            else:
                _lib.check(self.lib.lrx_flat_ip_search(_lib.ptr(self._x), self.ntotal, ldx, self.d, _lib.ptr(self._bounds), _lib.ptr(qc), qc.shape[0], k,
                                                       self.id_base, _lib.ptr(Dc), _lib.ptr(Ic), _lib.ptr(ws), ws.numel(), stream))

    def last_list_counts(self) -> torch.Tensor:
        """uint32-valued int64 tensor [q]: candidate-list entries per query of the last chunk of the last two-pass search (the rows that
        passed the filter threshold and reached the refine step) -- statistics for tools and bench legs.  Zeros when the last search was
        not a two-pass search (no candidate lists exist); raises before the first search."""
        if self._last_search is None:
            raise _lib.LrxError("last_list_counts(): no search has run on this index yet")
        nq, k, flags, has_shadow, ws, ntotal, two_pass = self._last_search
        out = torch.zeros(nq, dtype=torch.int32, device=self.device)
        if two_pass:                    # the workspace layout is the one planned for the ntotal of THAT search (add() / reset() since do not matter)
            _lib.check(self.lib.lrx_flat_ip_bounded_list_counts(_lib.ptr(ws), ntotal, self.d, nq, k, flags, int(has_shadow), _lib.ptr(out),
                                                                _lib.current_stream()))
        return out.to(torch.int64)


def merge_topk(D_parts: torch.Tensor, I_parts: torch.Tensor):
    """[R,Q,k] per-shard lists -> ([Q,k], [Q,k]) with the same ordering rule (score desc, id asc)."""
    lib = _lib.lib()
    R, Q, k = D_parts.shape
    D_parts, I_parts = D_parts.contiguous(), I_parts.contiguous()
    D = torch.empty(Q, k, dtype=torch.float32, device=D_parts.device)
    I = torch.empty(Q, k, dtype=torch.int64, device=D_parts.device)
    _lib.check(lib.lrx_merge_topk(_lib.ptr(D_parts), _lib.ptr(I_parts), R, Q, k, _lib.ptr(D), _lib.ptr(I), _lib.current_stream()))
    return D, I
