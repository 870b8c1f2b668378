#!/usr/bin/env python3
"""HIP-graph capture + replay of one FlatIPIndex.search (round 2's probe ended in a GPU memory-access fault and was
deleted).  Prints a line before every stage so that a fault names its stage; compares replayed results with the eager ones bit for bit
and times both.   N=... D=... Q=... K=... python tools/exp/graph_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lightretriever_amd import FlatIPIndex

def log(*a):
    print(*a, flush=True)

def main():
    N, D, Q, K = (int(os.environ.get(k, d)) for k, d in (("N", 200000), ("D", 256), ("Q", 100), ("K", 100)))
    g = torch.Generator(device="cuda").manual_seed(3)
    idx = FlatIPIndex(D, capacity=N)
    slot = idx.append_slot(N)
    for s in range(0, N, 65536):
        e = min(s + 65536, N)
        slot[s:e] = torch.nn.functional.normalize(torch.randn(e - s, D, generator=g, device="cuda"), dim=-1)
    idx.commit(N)
    q = torch.nn.functional.normalize(torch.randn(Q, D, generator=g, device="cuda"), dim=-1)
    log("eager warm-up")
    for _ in range(3):
        De, Ie = idx.search(q, K)
    torch.cuda.synchronize()
    De, Ie = De.clone(), Ie.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):                     # torch's own recipe: warm up on the capture stream first
        for _ in range(2):
            idx.search(q, K)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    log("capture")
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        Dg, Ig = idx.search(q, K)
    torch.cuda.synchronize()
    log("captured; replay 1")
    graph.replay()
    torch.cuda.synchronize()
    log("replay 1 done: ids equal", torch.equal(Ig, Ie), "scores equal", torch.equal(Dg, De))
    q2 = torch.nn.functional.normalize(torch.randn(Q, D, generator=g, device="cuda"), dim=-1)
    D2, I2 = idx.search(q2, K)
    D2, I2 = D2.clone(), I2.clone()
    q.copy_(q2)                                       # new queries in the captured input buffer
    graph.replay()
    torch.cuda.synchronize()
    log("replay 2 (new queries): ids equal", torch.equal(Ig, I2), "scores equal", torch.equal(Dg, D2))
    for name, fn in (("eager", lambda: idx.search(q, K)), ("graph", graph.replay)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(50):
            fn()
        e1.record()
        torch.cuda.synchronize()
        log("%s: %.4f ms per search (events), %.4f ms wall" % (name, e0.elapsed_time(e1) / 50, (time.perf_counter() - t0) * 1e3 / 50))
    # ---- why round 2's probe faulted: the FIRST search of a shape allocates the index's workspace; under capture that allocation comes from
    #      the graph's private pool and returns to the allocator with the graph, while the index keeps the pointer (no kernel is launched on
    #      the dangling pointer here -- addresses only)
    idx2 = FlatIPIndex(D, capacity=4096)
    idx2.add(torch.nn.functional.normalize(torch.randn(4096, D, generator=g, device="cuda"), dim=-1))
    need = int(idx2.lib.lrx_flat_ip_bounded_workspace_bytes(idx2.ntotal, D, Q, K, 0))
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2):
        ws = torch.empty(need, dtype=torch.uint8, device="cuda")      # what search() did under capture before it refused to
    p_ws = ws.data_ptr()
    del ws, g2
    torch.cuda.empty_cache()
    again = [torch.empty(need, dtype=torch.uint8, device="cuda") for _ in range(4)]
    log("workspace allocated under capture at 0x%x; after the graph is gone the allocator hands out %s -> %s" % (
        p_ws, ["0x%x" % t.data_ptr() for t in again], "SAME ADDRESS: a later eager search would have used freed memory" if any(
            t.data_ptr() == p_ws for t in again) else "no reuse observed in this run"))
    log("GRAPH PROBE OK")

main()
