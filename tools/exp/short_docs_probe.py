"""Token throughput of lrx_encode_packed at SHORT documents (real corpora: MS MARCO passages ~ 80 tokens): 131 072 tokens per call, documents of
L tokens each, Llama-3.2-1B dims.  Does the per-sequence work (attention workgroups per (sequence, kv head), pooled tail of B rows) cost tokens/s?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lightretriever_amd import EncoderConfig, LrxEncoder
cfg = EncoderConfig.llama32_1b(512)
enc = LrxEncoder.random_init(cfg, seed=0)
T = 131072
for L in [int(x) for x in os.environ.get("LS", "512,256,128,64,32,16").split(",")]:
    B = T // L
    ids = torch.randint(1000, 127000, (B * L,), device="cuda", dtype=torch.int64).to(torch.int32)
    cu = (torch.arange(B + 1, device="cuda", dtype=torch.int64) * L).to(torch.int32)
    out = torch.empty(B, cfg.hidden_size, device="cuda")
    enc.encode_packed(ids, cu, L, out=out)
    enc.set_profiling(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        enc.encode_packed(ids, cu, L, out=out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    prof = enc.get_profile()
    enc.set_profiling(False)
    print("L=%4d B=%5d: %.1f ms per call, %.0f k tokens/s, %.0f docs/s; per class ms: %s" % (
        L, B, dt * 1e3, B * L / dt / 1e3, B / dt, {k: round(v["ms"] / 3, 2) for k, v in prof.items() if v["ms"] > 0}), flush=True)
