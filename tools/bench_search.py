#!/usr/bin/env python3
"""Dev micro-benchmark of the flat-IP search kernels (score pass + select) at the BASELINE index size."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lightretriever_amd import FlatIPIndex

def main():
    N, D = int(os.environ.get("N", 1_000_000)), int(os.environ.get("D", 2048))
    g = torch.Generator(device="cuda").manual_seed(7)
    idx = FlatIPIndex(D, capacity=N)
    slot = idx.append_slot(N)
    for s in range(0, N, 65536):
        e = min(s + 65536, N)
        slot[s:e] = torch.nn.functional.normalize(torch.randn(e - s, D, generator=g, device="cuda"), dim=-1)
    idx.commit(N)
    for Q in [int(x) for x in os.environ.get("QS", "1,16,32,48,100,128").split(",")]:
        q = torch.nn.functional.normalize(torch.randn(Q, D, generator=g, device="cuda"), dim=-1)
        for _ in range(2):
            idx.search(q, 100)
        ts = []
        for _ in range(8):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); idx.search(q, 100); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        med = statistics.median(ts)
        bpe = 2 if (idx._xb is not None and idx.two_pass) else 4
        print(f"Q={Q:4d}: {med:.3f} ms  -> {N*D*bpe/med/1e6:.0f} GB/s corpus stream ({bpe} B/element), {Q/med*1e3:.0f} q/s, {2*Q*D*N/med/1e9:.1f} TFLOP/s", flush=True)

if __name__ == "__main__":
    main()
