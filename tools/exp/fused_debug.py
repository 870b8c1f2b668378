"""Dev: one bounded search on a small shard with the fused launch, checked against the fp64 host evaluation (bisecting aid: LRX_FUSED_PHASES).
Needs a -DLRX_DEV_KNOBS build of the library (`. tools/dev_lib.sh` builds it and exports LRX_LIB_DEV_VARIANT): the shipping liblrx.so reads no
environment variable."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch
from lightretriever_amd import FlatIPIndex
N, D, Q, K = int(os.environ.get("N", 40000)), int(os.environ.get("D", 256)), int(os.environ.get("Q", 100)), int(os.environ.get("K", 100))
g = torch.Generator(device="cuda").manual_seed(3)
idx = FlatIPIndex(D, capacity=N)
idx.add(torch.nn.functional.normalize(torch.randn(N, D, generator=g, device="cuda"), dim=-1))
q = torch.nn.functional.normalize(torch.randn(Q, D, generator=g, device="cuda"), dim=-1)
print("launch", flush=True)
t0 = time.time()
Dg, Ig = idx.search(q, K)
torch.cuda.synchronize()
print("done in %.3f s" % (time.time() - t0), flush=True)
S = (q.double() @ idx.vectors.double().T)
Dw, Iw = S.topk(K, dim=1)
print("ids equal:", bool((Iw == Ig).all()), "max score diff", float((Dw.float() - Dg).abs().max()), "fallbacks", idx.lib.lrx_search_fallback_count(1), flush=True)
print("hits per query: mean %.0f max %d" % (idx.last_list_counts().float().mean().item(), idx.last_list_counts().max().item()))
