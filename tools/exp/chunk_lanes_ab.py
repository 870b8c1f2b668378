import sys, torch
sys.path.insert(0, '/root/repo')
from bench import synthetic_index, time_search
dev = torch.device('cuda', 0)
idx, g = synthetic_index(100_000, 2048, dev, 31)
for Q in (1000, 512, 257):
    q = torch.nn.functional.normalize(torch.randn(Q, 2048, generator=g, device=dev), dim=-1)
    for rep in range(2):
        for lanes in (1, 2):
            idx.chunk_lanes = lanes
            ms, med = time_search(lambda: idx.search(q, 1000), 8)
            print("100k x 2048, Q=%d, k=1000, chunk_lanes=%d: %.4f ms (median %.4f)" % (Q, lanes, ms, med), flush=True)
