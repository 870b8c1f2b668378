"""EmbeddingBag table build (N1): reference-literal full-sequence loop vs shared-prefix build, Llama-3.2-1B dims.
usage: python tools/bench_embbag.py [--vocab 128256] [--prefix 21] [--model llama32_1b] [--full-rows 20000]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lightretriever_amd.encoder import EncoderConfig, LrxEncoder


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--vocab", type=int, default=128256)
    ap.add_argument("--prefix", type=int, default=21)          # [bos] + a 20-token instruction
    ap.add_argument("--model", default="llama32_1b")
    ap.add_argument("--full-rows", type=int, default=20000)    # rows timed on the full path (extrapolated linearly)
    ap.add_argument("--batch", type=int, default=5000)
    a = ap.parse_args()
    cfg = getattr(EncoderConfig, a.model)(512)
    enc = LrxEncoder.random_init(cfg, seed=0)
    dev = enc.device
    pre = torch.randint(5, 1000, (a.prefix,), dtype=torch.int32, device=dev)
    V, H, L = a.vocab, cfg.hidden_size, a.prefix + 2

    def build_fast():
        table = torch.empty(V, H, dtype=torch.float32, device=dev)
        step = a.batch * max(1, L // 2)
        for s in range(0, V, step):
            e = min(s + step, V)
            suf = torch.empty(e - s, 2, dtype=torch.int32, device=dev)
            suf[:, 0] = torch.arange(s, e, dtype=torch.int32, device=dev)
            suf[:, 1] = 2
            enc.encode_prefixed(pre, suf, out=table[s:e])
        return table

    def build_full(rows):
        table = torch.empty(rows, H, dtype=torch.float32, device=dev)
        base = torch.empty(a.batch, L, dtype=torch.int32, device=dev)
        base[:, :a.prefix] = pre
        base[:, -1] = 2
        for s in range(0, rows, a.batch):
            e = min(s + a.batch, rows)
            n = e - s
            base[:n, -2] = torch.arange(s, e, dtype=torch.int32, device=dev)
            cu = (torch.arange(n + 1, device=dev, dtype=torch.int64) * L).to(torch.int32)
            enc.encode_packed(base[:n].reshape(-1), cu, L, out=table[s:e], normalize=False)
        return table

    def timed(fn, *args):
        fn(*args); torch.cuda.synchronize()
        t0 = time.perf_counter(); r = fn(*args); torch.cuda.synchronize()
        return time.perf_counter() - t0, r

    tf, fast = timed(build_fast)
    rows = min(a.full_rows, V)
    tl, full = timed(build_full, rows)
    cos = torch.nn.functional.cosine_similarity(fast[:rows], full, dim=1)
    enc.set_profiling(True)
    suf = torch.stack([torch.arange(0, a.batch * (L // 2), dtype=torch.int32, device=dev), torch.full((a.batch * (L // 2),), 2, dtype=torch.int32, device=dev)], 1).contiguous()
    enc.encode_prefixed(pre, suf)
    prof = enc.get_profile()
    enc.set_profiling(False)
    print(json.dumps({"model": a.model, "vocab": V, "prefix_len": a.prefix, "shared_prefix_s": round(tf, 4),
                      "full_rows_timed": rows, "full_s_extrapolated": round(tl * V / rows, 3), "speedup": round(tl * V / rows / tf, 2),
                      "min_cos_fast_vs_full": float(cos.min()), "one_call_ms": {k: round(v["ms"], 3) for k, v in prof.items()}}))


if __name__ == "__main__":
    main()
