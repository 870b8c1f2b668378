#!/usr/bin/env python3
"""Does a second search in flight on another HIP stream hide the latency-bound kernels of the chain?  Two FlatIPIndex objects over the SAME
rows / shadow / bounds (separate workspaces), searches issued alternately on two streams, against the same searches on one stream.
env N, D, Q, K."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lightretriever_amd import FlatIPIndex

N, D, Q, K = (int(os.environ.get(k, v)) for k, v in (("N", 125000), ("D", 2048), ("Q", 100), ("K", 100)))
g = torch.Generator(device="cuda").manual_seed(7)
idx = FlatIPIndex(D, capacity=N)
slot = idx.append_slot(N)
for s in range(0, N, 65536):
    e = min(s + 65536, N)
    slot[s:e] = torch.nn.functional.normalize(torch.randn(e - s, D, generator=g, device="cuda"), dim=-1)
idx.commit(N)
lanes = [idx]
for _ in range(int(os.environ.get("LANES", 2)) - 1):
    v = FlatIPIndex(D)
    v._x, v._xb, v._bounds, v.ntotal, v._shadow_rows = idx._x, idx._xb, idx._bounds, N, N
    lanes.append(v)
qs = [torch.nn.functional.normalize(torch.randn(Q, D, generator=g, device="cuda"), dim=-1) for _ in range(4)]
streams = [torch.cuda.Stream() for _ in lanes]
ref = [idx.search(q, K) for q in qs]
torch.cuda.synchronize()

def run(n, n_lanes):
    outs = []
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for st in streams[:n_lanes]:
        st.wait_stream(torch.cuda.current_stream())
    e0.record()
    for st in streams[:n_lanes]:
        st.wait_event(e0)
    for i in range(n):
        l = i % n_lanes
        with torch.cuda.stream(streams[l]):
            outs.append(lanes[l].search(qs[i % 4], K))
    for st in streams[:n_lanes]:
        torch.cuda.current_stream().wait_stream(st)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, outs

for n_lanes in (1, 2, len(lanes)):
    run(8, n_lanes)
    ms, outs = run(80, n_lanes)
    ok = all(torch.equal(o[0], ref[i % 4][0]) and torch.equal(o[1], ref[i % 4][1]) for i, o in enumerate(outs))
    print(f"N={N} D={D} Q={Q} K={K}: {n_lanes} lane(s): {ms:.4f} ms per search ({Q / ms * 1e3:.0f} q/s), results identical: {ok}", flush=True)
