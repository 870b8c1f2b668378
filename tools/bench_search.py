#!/usr/bin/env python3
"""Dev micro-benchmark of the flat-IP search kernels (score pass + select) at the BASELINE index size."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lightretriever_amd import FlatIPIndex

def main():
    N, D = int(os.environ.get("N", 1_000_000)), int(os.environ.get("D", 2048))
    g = torch.Generator(device="cuda").manual_seed(7)
    idx = FlatIPIndex(D, capacity=N)
    if os.environ.get("LAYOUT"):      # shadow layout: tiled (default) | rows
        raise SystemExit("LAYOUT: the row-major shadow is gone (round 3): the shadow is the tiled fp16 layout of include/lrx.h")
    slot = idx.append_slot(N)
    for s in range(0, N, 65536):
        e = min(s + 65536, N)
        slot[s:e] = torch.nn.functional.normalize(torch.randn(e - s, D, generator=g, device="cuda"), dim=-1)
    idx.commit(N)
    if os.environ.get("MODE"):        # 1 = score-matrix filter, 2 = score-free filter (FlatIPIndex.search_flags -> the flags argument of lrx_flat_ip_search_bounded)
        setattr(idx, "search_flags", int(os.environ["MODE"]))
    K = int(os.environ.get("K", 100))
    for Q in [int(x) for x in os.environ.get("QS", "1,16,32,48,100,128").split(",")]:
        q = torch.nn.functional.normalize(torch.randn(Q, D, generator=g, device="cuda"), dim=-1)
        for _ in range(2):
            idx.search(q, K)
        ts = []
        for _ in range(8):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); idx.search(q, K); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        med = statistics.median(ts)
        if os.environ.get("STATS") and idx._xb is not None and Q >= 4:
            # what the score-free filter sees for a few queries: rows reaching the sample threshold T' - 2 eps, rows in the final band
            xb, k, ss = idx.shadow_rows(), K, 20
            nb = (N + 127) // 128
            samp = torch.arange(0, nb, ss, device="cuda").repeat_interleave(128) * 128 + torch.arange(128, device="cuda").repeat((nb + ss - 1) // ss)
            samp = samp[samp < N]
            for qi in range(4):
                qb = q[qi].to(torch.float16)
                sc = (xb @ qb).float()
                eps = float((q[qi] - qb.float()).norm() * idx._bounds[0] + qb.float().norm() * (idx._bounds[1] + (D + 32) * 2.0 ** -23 * idx._bounds[0]))
                tp = float(sc[samp].topk(k).values[-1]); kth = float(sc.topk(k).values[-1])
                print(f"   q{qi}: eps {eps:.5f}  T' {tp:.4f} kth~ {kth:.4f}  rows >= T'-2eps {int((sc >= tp - 2 * eps).sum())}  band rows {int((sc >= kth - 2 * eps).sum())}", flush=True)
        bpe = 2 if (idx._xb is not None and idx.two_pass) else 4
        print(f"Q={Q:4d}: {med:.3f} ms  -> {N*D*bpe/med/1e6:.0f} GB/s corpus stream ({bpe} B/element), {Q/med*1e3:.0f} q/s, {2*Q*D*N/med/1e9:.1f} TFLOP/s", flush=True)

if __name__ == "__main__":
    main()
