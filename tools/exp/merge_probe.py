#!/usr/bin/env python3
"""lrx_merge_topk_packed: time per call by number of gathered shards R (Q = 100, k = 100 / 1000), HIP events around 200 back-to-back calls."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lightretriever_amd import _lib
lib = _lib.lib()
for k in (100, 1000):
    for R in (1, 2, 4, 8):
        Q = 100
        g = torch.Generator(device="cuda").manual_seed(1)
        sc = torch.randn(R, Q, k, generator=g, device="cuda").sort(dim=-1, descending=True).values
        # rows ascending along each list: exact score ties (a few per 100 k draws) then sit in (score desc, row asc) order, as a search returns them
        ids = (torch.arange(k, device="cuda")[None, None, :] * R + torch.arange(R, device="cuda")[:, None, None]).expand(R, Q, k).contiguous()
        words = ((sc.view(torch.int32).to(torch.int64) << 32) | ids).contiguous()
        D = torch.empty(Q, k, device="cuda"); I = torch.empty(Q, k, dtype=torch.int64, device="cuda")
        fn = lambda: lib.lrx_merge_topk_packed(_lib.ptr(words), R, Q, k, _lib.ptr(D), _lib.ptr(I), _lib.current_stream())
        for _ in range(10): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200): fn()
        e1.record(); torch.cuda.synchronize()
        print(f"merge_topk_packed R={R} Q={Q} k={k}: {1e3 * e0.elapsed_time(e1) / 200:.2f} us per call", flush=True)
