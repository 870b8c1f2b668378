#!/usr/bin/env python3
"""Host-side cost of one index.search() call (launch path) vs its GPU time: enqueue 200 searches without synchronising."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lightretriever_amd import FlatIPIndex, ops

N, D, Q = 1_000_000, 2048, 100
g = torch.Generator(device="cuda").manual_seed(7)
idx = FlatIPIndex(D, capacity=N)
slot = idx.append_slot(N)
for s in range(0, N, 65536):
    e = min(s + 65536, N)
    slot[s:e] = torch.nn.functional.normalize(torch.randn(e - s, D, generator=g, device="cuda"), dim=-1)
idx.commit(N)
q = torch.nn.functional.normalize(torch.randn(Q, D, generator=g, device="cuda"), dim=-1)
for _ in range(5):
    idx.search(q, 100)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    idx.search(q, 100)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"enqueue {1e3 * (t1 - t0) / 200:.3f} ms per search (host), drained after {1e3 * (t2 - t0) / 200:.3f} ms per search (device-bound if larger)")

# the bench loop's shape: EmbeddingBag lookup + search per pass, with and without a pair of timing events per pass
V, H = 128256, D
table = torch.randn(V, H, generator=g, device="cuda")
lens = torch.randint(8, 33, (Q,), generator=g, device="cuda")
offs = torch.cat([torch.zeros(1, dtype=torch.int64, device="cuda"), lens.cumsum(0)[:-1]])
ids = torch.randint(1000, 127000, (int(lens.sum()),), generator=g, device="cuda")
def loop(events, embed=True, K=200):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2 * K)]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    qq = q
    for i in range(K):
        if embed: qq = ops.embedding_bag_mean(table, ids, offs, normalize=True)
        if events: ev[2 * i].record()
        idx.search(qq, 100)
        if events: ev[2 * i + 1].record()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / K
for name, kw in (("search only", dict(events=False, embed=False)), ("+ embedding bag", dict(events=False)), ("+ embedding bag + 2 events per pass", dict(events=True))):
    loop(**kw, K=20)
    print(f"{name}: {loop(**kw):.3f} ms per pass")
