#!/usr/bin/env python3
"""Host-side cost of the search call path on a TINY shard (GPU time << host time): what one FlatIPIndex.search, one ShardedFlatIPIndex.search
with the 1-rank RCCL exchange, and the bare C call cost the Python thread.  When the per-rank shard search takes 0.15 ms on the GPU, this is
the other bound on queries/s."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.distributed as dist
from lightretriever_amd import FlatIPIndex, _lib
from lightretriever_amd.sharded import ShardedFlatIPIndex

N, D, Q, K = 20000, 64, 100, 100
g = torch.Generator(device="cuda").manual_seed(7)
idx = FlatIPIndex(D, capacity=N)
idx.add(torch.nn.functional.normalize(torch.randn(N, D, generator=g, device="cuda"), dim=-1))
q = torch.nn.functional.normalize(torch.randn(Q, D, generator=g, device="cuda"), dim=-1)

def host_us(fn, n=300):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return 1e6 * (t1 - t0) / n, 1e6 * (t2 - t0) / n

print("FlatIPIndex.search                 host %.1f us per call (drained %.1f)" % host_us(lambda: idx.search(q, K)))
Dd, Ii = idx.search(q, K)
ws = idx._ws
lib = _lib.lib()
st = _lib.current_stream()
args = (_lib.ptr(idx._x), idx.ntotal, idx._x.stride(0), idx.d, _lib.ptr(idx._xb), _lib.ptr(idx._bounds), _lib.ptr(q), Q, K, 0, _lib.ptr(Dd), _lib.ptr(Ii), None, None,
        _lib.ptr(ws), ws.numel(), 0, st)
print("bare lrx_flat_ip_search_bounded_wire host %.1f us per call (drained %.1f)" % host_us(lambda: lib.lrx_flat_ip_search_bounded_wire(*args)))
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29900 + os.getpid() % 90), HSA_ENABLE_IPC_MODE_LEGACY="0", LRX_FORCE_COLLECTIVE="1")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
sh = ShardedFlatIPIndex(idx)
print("ShardedFlatIPIndex.search (1-rank) host %.1f us per call (drained %.1f)" % host_us(lambda: sh.search(q, K)))
w = torch.empty(Q, K, dtype=torch.int64, device="cuda")
out = torch.empty(Q, K, dtype=torch.int64, device="cuda")
print("dist.all_gather_into_tensor alone  host %.1f us per call (drained %.1f)" % host_us(lambda: dist.all_gather_into_tensor(out, w)))
dist.destroy_process_group()
