#!/usr/bin/env python3
"""First run of the persistent sample-pass kernel (k_filter_xreg_store): shapes with more than two sample block pairs per CU, results
compared bit for bit with the score-matrix filter.  Run under `timeout`: a barrier mismatch would hang."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lightretriever_amd import FlatIPIndex

for (N, D, Q, k) in [(300000, 256, 20, 1000), (300001, 512, 100, 1000), (280000, 256, 128, 2000), (2700000, 256, 100, 100)]:
    g = torch.Generator(device="cuda").manual_seed(N + D)
    idx = FlatIPIndex(D, capacity=N)
    slot = idx.append_slot(N)
    for s in range(0, N, 1 << 18):
        e = min(s + (1 << 18), N)
        slot[s:e] = torch.nn.functional.normalize(torch.randn(e - s, D, generator=g, device="cuda"), dim=-1)
    idx.commit(N)
    q = torch.nn.functional.normalize(torch.randn(Q, D, generator=g, device="cuda"), dim=-1)
    setattr(idx, "search_flags", 2)
    D2, I2 = idx.search(q, k)
    torch.cuda.synchronize()
    setattr(idx, "search_flags", 1)
    D1, I1 = idx.search(q, k)
    torch.cuda.synchronize()
    setattr(idx, "search_flags", 0)
    print((N, D, Q, k), "equal" if torch.equal(D1, D2) and torch.equal(I1, I2) else "DIFFERENT", flush=True)
    del idx
