#!/usr/bin/env python3
"""Phase times of k_refine_topk (diagnostic build -DSEARCH_TRACE): LRX_LIB_DEV_VARIANT=.../liblrx_strace.so python tools/refine_timeline.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lightretriever_amd import FlatIPIndex, _lib

def main():
    lib = _lib.lib()
    lib.lrx_debug_read_search_trace.restype = C.c_int
    lib.lrx_debug_read_search_trace.argtypes = [C.c_void_p, C.c_size_t]
    N, D, Q = 1_000_000, 2048, 100
    g = torch.Generator(device="cuda").manual_seed(7)
    idx = FlatIPIndex(D, capacity=N)
    slot = idx.append_slot(N)
    for s in range(0, N, 65536):
        e = min(s + 65536, N)
        slot[s:e] = torch.nn.functional.normalize(torch.randn(e - s, D, generator=g, device="cuda"), dim=-1)
    idx.commit(N)
    q = torch.nn.functional.normalize(torch.randn(Q, D, generator=g, device="cuda"), dim=-1)
    for _ in range(3):
        idx.search(q, 100)
    torch.cuda.synchronize()
    buf = np.zeros(8 * 1024, np.int64)
    assert lib.lrx_debug_read_search_trace(buf.ctypes.data, buf.nbytes) == 0
    t = buf.reshape(-1, 8)[:Q]
    us = lambda a: a / 100.0
    names = ["|q| + kth", "block-max scan", "candidate gather", "exact rescoring (+publish)", "merge + sort + write (last part)"]
    for i, n in enumerate(names):
        d = us(t[:, i + 1] - t[:, i])
        print(f"{n:18s} median {np.median(d):7.2f} us   p90 {np.percentile(d, 90):7.2f}")
    print("candidates per query: median", int(np.median(t[:, 6])), "max", int(t[:, 6].max()), "| qualifying blocks: median", int(np.median(t[:, 7])))
    print("whole kernel (first start -> last end):", us(t[:, 5].max() - t[:, 0].min()), "us; per-block total median", np.median(us(t[:, 5] - t[:, 0])))

if __name__ == "__main__":
    main()
