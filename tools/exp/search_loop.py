"""Steady load for tools/exp/search_power.sh: LOOPS back-to-back searches of Q queries over an N x D shard (env N, D, Q, K, LOOPS)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lightretriever_amd import FlatIPIndex
N, D, Q, K, LOOPS = (int(os.environ.get(k, v)) for k, v in (("N", 1_000_000), ("D", 2048), ("Q", 1000), ("K", 100), ("LOOPS", 2000)))
g = torch.Generator(device="cuda").manual_seed(7)
idx = FlatIPIndex(D, capacity=N)
slot = idx.append_slot(N)
for s in range(0, N, 65536):
    e = min(s + 65536, N)
    slot[s:e] = torch.nn.functional.normalize(torch.randn(e - s, D, generator=g, device="cuda"), dim=-1)
idx.commit(N)
q = torch.nn.functional.normalize(torch.randn(Q, D, generator=g, device="cuda"), dim=-1)
idx.search(q, K)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(LOOPS):
    idx.search(q, K)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("Q=%d over %d x %d: %.3f ms per search, %.1f TFLOP/s (f16 MFMA)" % (Q, N, D, 1e3 * dt / LOOPS, 2.0 * Q * D * N * LOOPS / dt / 1e12))
