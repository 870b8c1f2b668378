"""Sparse document vectors (N2) at Llama-3.2-1B dims: LM-head max-aggregation GEMM alone and the dense+sparse encode call.
usage: python tools/bench_sparse.py [--docs 256] [--seq 512] [--iters 5]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lightretriever_amd import ops
from lightretriever_amd.encoder import EncoderConfig, LrxEncoder


def timed(fn, iters):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--docs", type=int, default=256)
    ap.add_argument("--seq", type=int, default=512)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--model", default="llama32_1b")
    a = ap.parse_args()
    cfg = getattr(EncoderConfig, a.model)(512)
    enc = LrxEncoder.random_init(cfg, seed=0)
    dev = enc.device
    B, S, H, V = a.docs, a.seq, cfg.hidden_size, cfg.vocab_size
    T = B * S
    ids = torch.randint(5, V, (T,), dtype=torch.int32, device=dev)
    cu = (torch.arange(B + 1, device=dev, dtype=torch.int64) * S).to(torch.int32)
    hid = torch.randn(T, H, device=dev).to(torch.bfloat16)
    ms_agg = timed(lambda: ops.sparse_max_aggregate(hid, enc.embed, cu, None), a.iters)
    reps = ops.sparse_max_aggregate(hid, enc.embed, cu, None)
    ms_spf = timed(lambda: ops.sparsify_(reps.clone(), True, True, True, 0, 8), a.iters) - timed(lambda: reps.clone(), a.iters)
    ms_topk = timed(lambda: ops.sparsify_(reps.clone(), True, True, True, 256, 8), a.iters) - timed(lambda: reps.clone(), a.iters)
    sp = ops.sparsify_(reps.clone(), True, True, True, 256, 8)
    ms_cmp = timed(lambda: ops.sparse_compact(sp, 100, 512), a.iters)
    ms_dense = timed(lambda: enc.encode_packed(ids, cu, S), a.iters)
    ms_both = timed(lambda: enc.encode_packed_sparse(ids, cu, S), a.iters)
    enc.set_profiling(True)
    enc.encode_packed_sparse(ids, cu, S)
    prof = enc.get_profile()
    enc.set_profiling(False)
    flops = 2.0 * T * V * H
    print(json.dumps({"model": a.model, "docs": B, "seq": S, "max_aggregate_ms": round(ms_agg, 3), "max_aggregate_tflops": round(flops / ms_agg / 1e9, 1),
                      "sparsify_relu_log1p_ms": round(ms_spf, 3), "sparsify_top256_ms": round(ms_topk, 3), "compact_ms": round(ms_cmp, 3),
                      "dense_only_ms": round(ms_dense, 2), "dense_plus_sparse_ms": round(ms_both, 2),
                      "docs_per_s_dense": round(B / ms_dense * 1e3, 1), "docs_per_s_dense_plus_sparse": round(B / ms_both * 1e3, 1),
                      "profile_ms": {k: round(v["ms"], 2) for k, v in prof.items()}}))


if __name__ == "__main__":
    main()
