import os, sys, statistics
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from lightretriever_amd import FlatIPIndex
N, D = 1_000_000, 2048
g = torch.Generator(device="cuda").manual_seed(7)
idx = FlatIPIndex(D, capacity=N)
slot = idx.append_slot(N)
for s in range(0, N, 65536):
    e = min(s + 65536, N)
    slot[s:e] = torch.nn.functional.normalize(torch.randn(e - s, D, generator=g, device="cuda"), dim=-1)
idx.commit(N)
idx.two_pass = False
for Q in (64, 100):
    q = torch.nn.functional.normalize(torch.randn(Q, D, generator=g, device="cuda"), dim=-1)
    for _ in range(2): idx.search(q, 100)
    ts = []
    for _ in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); idx.search(q, 100); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    print(f"six-product exact path Q={Q}: {statistics.median(ts):.3f} ms")
