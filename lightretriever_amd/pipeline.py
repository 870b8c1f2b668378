"""Searches in flight on several HIP streams ("lanes").

One search of the bounded chain is a few HBM-bound passes framed by short latency-bound kernels (query packing, threshold selection, band
rescoring, merge, the idle gated fallback -- each ~5 us of a dependent launch).  On ONE stream those frames leave the
memory system idle: 35 of the 155 us of a 125 k-row shard search, the per-rank shard of an 8-GPU run.  A serving loop that has the next
query batch at hand can keep two searches in flight: the frames of one overlap the streaming passes of the other.  Nothing inside the
library changes -- the C ABI launches on the stream it is handed; each lane has its own search workspace (FlatIPIndex.search(lane=...)).

    lanes = SearchLanes(index_or_sharded_index, lanes=2)
    h = lanes.submit(q, k)            # or submit(lambda: ops.embedding_bag_mean(...), k): the query producer runs on the lane's stream too
    ...                               # submit the next batch before asking for this one's result
    D, I = h.result()                 # the caller's current stream now waits for that search; tensors safe to use on it

Replaces nothing in the reference (faiss searches are synchronous calls, retriever/faiss_index.py:27-40); it is how this build turns the
per-search latency of a small shard into throughput.  Results are bit-identical to the same searches issued one after the other."""
from __future__ import annotations

from typing import Callable, Union

import torch


class _Pending:
    def __init__(self, out, event: torch.cuda.Event):
        self._out, self._event = out, event

    def result(self):
        cur = torch.cuda.current_stream()
        cur.wait_event(self._event)
        for t in self._out:
            if torch.is_tensor(t) and t.is_cuda:
                t.record_stream(cur)          # allocated on the lane's stream, consumed on this one
        return self._out


class SearchLanes:
    def __init__(self, target, lanes: int = 2):
        """target: FlatIPIndex or ShardedFlatIPIndex (anything with search(q, k, lane=...))."""
        if lanes < 1:
            raise ValueError("lanes >= 1")
        self.target = target
        dev = getattr(target, "device", None) or target.shard.device
        self.streams = [torch.cuda.Stream(device=dev) for _ in range(lanes)]
        self._next = 0

    def submit(self, q: Union[torch.Tensor, Callable[[], torch.Tensor]], k: int) -> _Pending:
        lane = self._next % len(self.streams)
        self._next += 1
        st = self.streams[lane]
        st.wait_stream(torch.cuda.current_stream())      # whatever produced the inputs on the caller's stream is done first
        with torch.cuda.stream(st):
            qq = q() if callable(q) else q
            out = self.target.search(qq, k, lane=lane + 1)  # lane 0 stays the workspace of plain search() calls on the caller's stream
            ev = torch.cuda.Event()
            ev.record(st)
        if torch.is_tensor(qq) and qq.is_cuda:
            qq.record_stream(st)
        return _Pending(out, ev)

    def drain(self):
        """Make the caller's current stream wait for everything submitted so far."""
        for st in self.streams:
            torch.cuda.current_stream().wait_stream(st)
