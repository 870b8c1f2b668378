"""Exact rescoring grouped by row against the per-query gather (lrx_flat_ip_search_bounded flags LRX_SEARCH_REFINE_ROWS_ALWAYS / _NEVER) at the
reference's evaluation point and around it; same box, alternating.  python3 tools/exp/refine_rows_ab.py"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from lightretriever_amd import FlatIPIndex, _lib


def run(N, D, Q, k):
    g = torch.Generator(device="cuda").manual_seed(1)
    X = torch.randn(N, D, generator=g, device="cuda")
    X /= X.norm(dim=1, keepdim=True)
    q = torch.randn(Q, D, generator=g, device="cuda")
    idx = FlatIPIndex(D, capacity=N)
    idx.shadow_f16 = True
    idx.add(X)
    out = {}
    for rnd in range(2):
        for name, fl in (("gather", _lib.SEARCH_REFINE_ROWS_NEVER), ("rows", _lib.SEARCH_REFINE_ROWS_ALWAYS), ("rule", 0)):
            FlatIPIndex.search_flags = fl
            for _ in range(2):
                idx.search(q, k)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                r = idx.search(q, k)
            e1.record(); torch.cuda.synchronize()
            out.setdefault(name, []).append(e0.elapsed_time(e1) / 5)
            out.setdefault(name + "_res", r)
    FlatIPIndex.search_flags = 0
    same = torch.equal(out["gather_res"][0], out["rows_res"][0]) and torch.equal(out["gather_res"][1], out["rows_res"][1])
    print("%7d x %4d  Q=%4d k=%4d : gather %s ms | by row %s ms | rule %s ms | same bits %s" % (
        N, D, Q, k, "/".join("%.3f" % v for v in out["gather"]), "/".join("%.3f" % v for v in out["rows"]), "/".join("%.3f" % v for v in out["rule"]), same), flush=True)


for cfg in [(100000, 2048, 1000, 1000), (100000, 2048, 256, 1000), (100000, 2048, 100, 1000), (100000, 1024, 1000, 1000), (100000, 4096, 1000, 1000),
            (50000, 2048, 1000, 1000), (200000, 2048, 1000, 1000), (100000, 2048, 1000, 100), (1000000, 2048, 1000, 1000)]:
    run(*cfg)
