#!/usr/bin/env python3
"""Index persistence: save / load time of a FlatIPIndex shard in the Faiss flat-index file layout, and that the
reloaded shard returns the same hits.  N, D from the environment (default 500000 x 2048 = 4.1 GB)."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lightretriever_amd import FlatIPIndex

def main():
    N, D = int(os.environ.get("N", 500_000)), int(os.environ.get("D", 2048))
    g = torch.Generator(device="cuda").manual_seed(7)
    idx = FlatIPIndex(D, capacity=N)
    slot = idx.append_slot(N)
    for s in range(0, N, 65536):
        e = min(s + 65536, N)
        slot[s:e] = torch.nn.functional.normalize(torch.randn(e - s, D, generator=g, device="cuda"), dim=-1)
    idx.commit(N)
    q = torch.nn.functional.normalize(torch.randn(100, D, generator=g, device="cuda"), dim=-1)
    D0, I0 = idx.search(q, 100)
    gb = N * D * 4 / 1e9
    with tempfile.TemporaryDirectory(dir=os.environ.get("TMPDIR", "/tmp")) as td:
        f = os.path.join(td, "shard.flat.faiss")
        torch.cuda.synchronize(); t0 = time.perf_counter()
        idx.save(f)
        t_save = time.perf_counter() - t0
        t0 = time.perf_counter()
        idx2 = FlatIPIndex.load(f)
        torch.cuda.synchronize()
        t_load = time.perf_counter() - t0
        t0 = time.perf_counter()
        idx3 = FlatIPIndex.load(f)            # second load: file in the page cache
        torch.cuda.synchronize()
        t_load2 = time.perf_counter() - t0
    D1, I1 = idx2.search(q, 100)
    assert torch.equal(D0, D1) and torch.equal(I0, I1)
    print(f"{N} x {D} fp32 = {gb:.2f} GB: save {t_save:.2f} s ({gb / t_save:.2f} GB/s), load {t_load:.2f} s ({gb / t_load:.2f} GB/s), "
          f"load from page cache {t_load2:.2f} s ({gb / t_load2:.2f} GB/s); hits identical after reload", flush=True)

if __name__ == "__main__":
    main()
