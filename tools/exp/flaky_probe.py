#!/usr/bin/env python3
"""Replay of search_stress.py seed 3, shape (400000, 1024), round 21: Q = 256, k = 2048, one query differs between mode 3 and mode 1."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from lightretriever_amd import FlatIPIndex
rng = np.random.default_rng(3)
shapes = [(1_000_000, 256), (600_000, 512), (3_000_001, 256), (400_000, 1024)]
rounds = 40
for (N, D) in shapes[:3]:
    for r in range(rounds):
        rng.choice([1, 2, 7, 16, 33, 64, 100, 113, 128, 129, 200, 256, 300]); rng.choice([1, 10, 100, 500, 1000, 2048]); rng.choice([0, 2, 3])
N, D = shapes[3]
g = torch.Generator(device="cuda").manual_seed(N + D)
idx = FlatIPIndex(D, capacity=N)
slot = idx.append_slot(N)
for s in range(0, N, 1 << 18):
    e = min(s + (1 << 18), N)
    slot[s:e] = torch.nn.functional.normalize(torch.randn(e - s, D, generator=g, device="cuda"), dim=-1) * (0.5 + torch.rand(e - s, 1, generator=g, device="cuda"))
idx.commit(N)
for r in range(rounds):
    Q = int(rng.choice([1, 2, 7, 16, 33, 64, 100, 113, 128, 129, 200, 256, 300])); k = int(rng.choice([1, 10, 100, 500, 1000, 2048])); mode = int(rng.choice([0, 2, 3]))
    q = torch.randn(Q, D, generator=g, device="cuda")
    if r != 21: continue
    print("round", r, Q, k, mode)
    setattr(idx, "search_flags", mode); D2, I2 = idx.search(q, k)
    setattr(idx, "search_flags", 1); D1, I1 = idx.search(q, k)
    setattr(idx, "search_flags", 0)
    bad = ((D1 != D2) | (I1 != I2)).any(dim=1).nonzero().flatten().tolist()
    print("bad queries", bad)
    for b in bad[:2]:
        pos = ((D1[b] != D2[b]) | (I1[b] != I2[b])).nonzero().flatten().tolist()
        print(" query", b, "differing positions", pos[:10], "count", len(pos))
        p0 = pos[0]
        for name, Dx, Ix in (("mode%d" % mode, D2, I2), ("mode1", D1, I1)):
            print("  ", name, "pos", p0 - 1, "..", p0 + 2, [(int(Ix[b, j]), float(Dx[b, j])) for j in range(max(0, p0 - 1), min(k, p0 + 3))])
        ex = (q[b].double() @ idx.vectors.double().T)
        s1, s2 = set(I1[b].tolist()), set(I2[b].tolist())
        print("   only in mode1:", sorted(s1 - s2)[:5], "only in mode%d:" % mode, sorted(s2 - s1)[:5])
        for row in list(s1 - s2)[:3] + list(s2 - s1)[:3]:
            print("     row", row, "exact fp64 score", float(ex[row]), "fp32 dot", float((q[b] * idx.vectors[row]).sum()))
        kth = ex.topk(k).values[-1]
        print("   fp64 k-th score", float(kth), "k+1-th", float(ex.topk(k + 1).values[-1]))
    break
