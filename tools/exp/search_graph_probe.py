import sys, os, statistics
sys.path.insert(0, "/root/repo")
import torch
from lightretriever_amd import FlatIPIndex
N, D, Q, k = 1_000_000, 2048, 100, 100
g = torch.Generator(device="cuda").manual_seed(7)
idx = FlatIPIndex(D, capacity=N)
slot = idx.append_slot(N)
for s in range(0, N, 65536):
    e = min(s + 65536, N)
    slot[s:e] = torch.nn.functional.normalize(torch.randn(e - s, D, generator=g, device="cuda"), dim=-1)
idx.commit(N)
q = torch.nn.functional.normalize(torch.randn(Q, D, generator=g, device="cuda"), dim=-1)
for _ in range(3): Dr, Ir = idx.search(q, k)
torch.cuda.synchronize()
def timeit(fn, n=20):
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return statistics.median(ts)
print("eager  %.3f ms" % timeit(lambda: idx.search(q, k)))
graph = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2): idx.search(q, k)
torch.cuda.current_stream().wait_stream(s)
with torch.cuda.graph(graph):
    Dg, Ig = idx.search(q, k)
graph.replay(); torch.cuda.synchronize()
print("graph matches eager:", torch.equal(Dg, Dr), torch.equal(Ig, Ir))
print("graph  %.3f ms" % timeit(lambda: graph.replay()))
