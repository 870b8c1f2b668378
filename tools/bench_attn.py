#!/usr/bin/env python3
"""Dev micro-benchmark of the attention launch at the encoder's shapes: on a prebuilt work list (as the encoder calls it; default) or,
with WALKER=1, without one (lrx_attn_varlen_causal).  B / S / SHAPES / REPS from the environment."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lightretriever_amd import ops

def main():
    B, S = int(os.environ.get("B", 256)), int(os.environ.get("S", 512))
    shapes = [(32, 8, 64), (32, 8, 128), (12, 2, 128), (28, 4, 128), (12, 2, 64)]
    if os.environ.get("SHAPES"):      # e.g. SHAPES=32-8-128,12-2-128
        shapes = [tuple(int(v) for v in sh.split("-")) for sh in os.environ["SHAPES"].split(",")]
    for (nq, nkv, d) in shapes:
        T = B * S
        g = torch.Generator(device="cuda").manual_seed(0)
        qkv = torch.randn(T, (nq + 2 * nkv) * d, generator=g, device="cuda").to(torch.float16)
        cu = (torch.arange(B + 1, device="cuda") * S).to(torch.int32)
        wl = False if os.environ.get("WALKER") == "1" else ops.attn_work_list(cu, T, S, nq, nkv, d)
        for _ in range(2):
            ops.attn_varlen_causal(qkv, cu, S, nq, nkv, d, work_list=wl)
        ts = []
        for _ in range(int(os.environ.get("REPS", 8))):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); ops.attn_varlen_causal(qkv, cu, S, nq, nkv, d, work_list=wl); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        med = statistics.median(ts)
        fl = 4.0 * d * nq * B * (S * (S + 1) / 2)
        print(f"attn{' (walker)' if wl is False else ''} nq={nq} nkv={nkv} d={d} B={B} S={S}: {med:.3f} ms = {fl/med/1e9:.1f} TF/s (causal flops)", flush=True)

if __name__ == "__main__":
    main()
