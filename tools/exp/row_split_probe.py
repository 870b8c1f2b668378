#!/usr/bin/env python3
"""One search over a per-rank shard as P row-wise sub-searches on P internal HIP streams + one merge of the P result lists: the latency-bound
kernels that frame one sub-search's passes (selection, refine, merge, launches) overlap the other's streaming passes -- the overlap two
searches in flight buy (pipeline.SearchLanes), inside ONE search.  Eager launches and a captured HIP graph, same box, same index.

    python tools/exp/row_split_probe.py [rows=125000] [dim=2048] [Q=100] [k=100]
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import bench
from lightretriever_amd import _lib


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 125_000
    D = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
    Q = int(sys.argv[3]) if len(sys.argv) > 3 else 100
    k = int(sys.argv[4]) if len(sys.argv) > 4 else 100
    dev = torch.device("cuda", 0)
    idx, g = bench.synthetic_index(N, D, dev, 41)
    q = torch.nn.functional.normalize(torch.randn(Q, D, generator=g, device=dev), dim=-1)
    lib = _lib.lib()
    D0, I0 = idx.search(q, k)
    base_ms, base_med = bench.time_search(lambda: idx.search(q, k), 200)
    print("%d x %d, Q = %d, k = %d: one chain %.4f ms (median %.4f)" % (N, D, Q, k, base_ms, base_med), flush=True)
    flags = int(idx.search_flags)
    x, xb, bounds = idx._x, idx._xb, idx._bounds
    ldx = x.stride(0)
    for P in (2, 3, 4):
        nblk = (N + 127) // 128
        cuts = [min(N, ((nblk * p) // P) * 128) for p in range(P + 1)]
        cuts[-1] = N
        parts = [(cuts[p], cuts[p + 1] - cuts[p]) for p in range(P)]
        streams = [torch.cuda.Stream(device=dev) for _ in range(P)]
        wss = [torch.empty(int(lib.lrx_flat_ip_bounded_workspace_bytes(n, D, Q, k, flags)), dtype=torch.uint8, device=dev) for _, n in parts]
        Dp = torch.empty(P, Q, k, dtype=torch.float32, device=dev)
        Ip = torch.empty(P, Q, k, dtype=torch.int64, device=dev)
        Do = torch.empty(Q, k, dtype=torch.float32, device=dev)
        Io = torch.empty(Q, k, dtype=torch.int64, device=dev)

        def run():
            cur = torch.cuda.current_stream()
            ev = torch.cuda.Event()
            ev.record(cur)
            for p, (r0, n) in enumerate(parts):
                st = streams[p]
                st.wait_event(ev)
                _lib.check(lib.lrx_flat_ip_search_bounded(
                    C.c_void_p(x.data_ptr() + r0 * ldx * 4), n, ldx, D, C.c_void_p(xb.data_ptr() + r0 * D * 2), _lib.ptr(bounds), _lib.ptr(q), Q, k,
                    int(idx.id_base) + r0, _lib.ptr(Dp[p]), _lib.ptr(Ip[p]), _lib.ptr(wss[p]), wss[p].numel(), flags, C.c_void_p(st.cuda_stream)))
            for st in streams:
                cur.wait_stream(st)
            _lib.check(lib.lrx_merge_topk(_lib.ptr(Dp), _lib.ptr(Ip), P, Q, k, _lib.ptr(Do), _lib.ptr(Io), C.c_void_p(cur.cuda_stream)))

        run()
        torch.cuda.synchronize()
        same = torch.equal(Do, D0) and torch.equal(Io, I0)
        ms, med = bench.time_search(run, 200)
        # the same as a captured graph (no host cost between the launches)
        gms = gmed = float("nan")
        try:
            side = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(side):
                run()
                torch.cuda.synchronize()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=side):
                    run()
            torch.cuda.synchronize()
            graph.replay()
            torch.cuda.synchronize()
            same_g = torch.equal(Do, D0) and torch.equal(Io, I0)
            gms, gmed = bench.time_search(graph.replay, 200)
        except Exception as e:  # noqa: BLE001
            same_g = "capture failed: %r" % (e,)
        print("  %d row parts on %d streams + merge: eager %.4f ms (median %.4f), bit-identical %s; graph replay %.4f ms (median %.4f), bit-identical %s" % (
            P, P, ms, med, same, gms, gmed, same_g), flush=True)
    # one chain as a graph, for the same comparison
    try:
        side = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(side):
            idx.search(q, k)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):
                Dg, Ig = idx.search(q, k)
        torch.cuda.synchronize()
        gms, gmed = bench.time_search(graph.replay, 200)
        print("  one chain, graph replay: %.4f ms (median %.4f)" % (gms, gmed))
    except Exception as e:  # noqa: BLE001
        print("  one chain, graph capture failed: %r" % (e,))


if __name__ == "__main__":
    main()
