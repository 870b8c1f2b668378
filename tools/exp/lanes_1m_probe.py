#!/usr/bin/env python3
"""Two searches in flight at 1M x 2048 through pipeline.SearchLanes: fixed queries vs the EmbeddingBag producer on the lane, 1 / 2 / 3 lanes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lightretriever_amd import FlatIPIndex, ops
from lightretriever_amd.pipeline import SearchLanes
N, D, Q, K = 1_000_000, 2048, 100, 100
g = torch.Generator(device="cuda").manual_seed(7)
idx = FlatIPIndex(D, capacity=N)
slot = idx.append_slot(N)
for s in range(0, N, 65536):
    e = min(s + 65536, N)
    slot[s:e] = torch.nn.functional.normalize(torch.randn(e - s, D, generator=g, device="cuda"), dim=-1)
idx.commit(N)
table = torch.randn(128256, D, generator=g, device="cuda")
lens = torch.randint(8, 33, (Q,), generator=g, device="cuda")
offs = (torch.cumsum(lens, 0) - lens).to(torch.int64)
ids = torch.randint(1000, 127000, (int(lens.sum()),), generator=g, device="cuda")
q_fixed = ops.embedding_bag_mean(table, ids, offs, normalize=True)
mk = lambda: ops.embedding_bag_mean(table, ids, offs, normalize=True)
def run(n_lanes, producer, n=60):
    sl = SearchLanes(idx, lanes=n_lanes)
    for _ in range(6): sl.submit(mk if producer else q_fixed, K)
    sl.drain(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    pend = [sl.submit(mk if producer else q_fixed, K) for _ in range(n)]
    sl.drain(); torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n
for producer in (False, True):
    for n_lanes in (1, 2, 3, 1, 2):
        print(f"1M x 2048 Q=100: lanes={n_lanes} producer_on_lane={producer}: {run(n_lanes, producer):.4f} ms per search", flush=True)
