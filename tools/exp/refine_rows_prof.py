"""One shape of tools/exp/refine_rows_ab.py for `rocprofv3 --kernel-trace --stats`: 10 searches of 1000 queries, k = 1000, over 100 k x 2048 with the rule's path."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from lightretriever_amd import FlatIPIndex, _lib
N, D, Q, k = 100000, 2048, 1000, 1000
g = torch.Generator(device="cuda").manual_seed(1)
X = torch.randn(N, D, generator=g, device="cuda"); X /= X.norm(dim=1, keepdim=True)
q = torch.randn(Q, D, generator=g, device="cuda")
idx = FlatIPIndex(D, capacity=N); idx.shadow_f16 = True; idx.add(X)
FlatIPIndex.search_flags = int(os.environ.get("FLAGS", "0"))
for _ in range(10):
    idx.search(q, k)
torch.cuda.synchronize()
