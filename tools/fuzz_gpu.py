#!/usr/bin/env python3
"""Randomised shape fuzz of the kernels against plain torch fp32 / the oracle (run on the GPU box): GEMM epilogues, varlen attention,
two-pass search, encoder vs oracle.  Prints one line per failure and a summary; exit code 1 on any failure.
usage: python tools/fuzz_gpu.py [--seed 0] [--rounds 40]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import lrx_oracle as O
from lightretriever_amd import ops, FlatIPIndex
from lightretriever_amd.encoder import interleave_gate_up

def bf(a): return torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda().to(torch.bfloat16).contiguous()
def hf(a): return torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda().to(torch.float16).contiguous()   # q | k | v activations are fp16

def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--seed", type=int, default=0); ap.add_argument("--rounds", type=int, default=40)
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    fails = 0
    def check(name, ok, info):
        nonlocal fails
        if not ok:
            fails += 1
            print("FAIL", name, info, flush=True)
    for r in range(a.rounds):
        # ---- GEMM
        M = int(rng.choice([1, 7, 255, 256, 257, 300, 511, 1000, 2049])); K = 64 * int(rng.integers(1, 40)); epi = int(rng.integers(0, 3))
        N = 8 * int(rng.integers(1, 130)) if epi != 2 else 32 * int(rng.integers(1, 40))
        A = O.round_bf16(rng.standard_normal((M, K)).astype(np.float32)); B = O.round_bf16(rng.standard_normal((N, K)).astype(np.float32) * 0.05)
        At, Bt = bf(A), bf(B)
        ref = At.float() @ Bt.float().T
        if epi == 0:
            bias = bf(O.round_bf16(rng.standard_normal(N).astype(np.float32)))
            got = ops.gemm_bf16_nt(At, Bt, bias=bias, epilogue=0).float(); want = (ref + bias.float()).to(torch.bfloat16).float()
        elif epi == 1:
            R = bf(O.round_bf16(rng.standard_normal((M, N)).astype(np.float32)))
            got = ops.gemm_bf16_nt(At, Bt, resid=R, epilogue=1).float(); want = (ref.to(torch.bfloat16).float() + R.float()).to(torch.bfloat16).float()
        else:
            g, u = ref[:, :N // 2], ref[:, N // 2:]
            Bi = interleave_gate_up(Bt[:N // 2].float().cpu(), Bt[N // 2:].float().cpu()).cuda().to(torch.bfloat16).contiguous()
            got = ops.gemm_bf16_nt(At, Bi, epilogue=2).float(); want = (torch.nn.functional.silu(g) * u).to(torch.bfloat16).float()
        tol = 2.0 ** -7 * want.abs() + 2.0 ** -7 * ref.abs().max() * 0.02 + 1e-3
        bad = ((got - want).abs() > tol).float().mean().item()
        # (a rounding-boundary flip of the bf16 intermediate of the residual epilogue is worth one element: with M = 1 a single one is > 2e-3 of the row)
        check("gemm", (bad < 2e-3 or bad * got.numel() <= 2.5) and got.shape == want.shape, (M, N, K, epi, bad))
        # ---- attention
        d = int(rng.choice([64, 128])); nkv = int(rng.choice([1, 2, 4])); grp = int(rng.choice([1, 2, 4, 6, 7, 8])); nq = nkv * grp
        lens = [int(x) for x in rng.integers(1, 300, size=int(rng.integers(1, 6)))]
        T = sum(lens); W = (nq + 2 * nkv) * d
        qkv = O.round_bf16(rng.standard_normal((T, W)).astype(np.float32)); cu = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        got = ops.attn_varlen_causal(hf(qkv), torch.from_numpy(cu).cuda(), max(lens), nq, nkv, d).float().cpu().numpy()
        q = qkv[:, :nq * d].reshape(T, nq, d); k = qkv[:, nq * d:(nq + nkv) * d].reshape(T, nkv, d); v = qkv[:, (nq + nkv) * d:].reshape(T, nkv, d)
        want = np.zeros((T, nq, d), np.float32)
        for b in range(len(lens)):
            s, e = cu[b], cu[b + 1]; L = e - s; causal = np.tril(np.ones((L, L), bool))
            for h in range(nq):
                sc = np.where(causal, (q[s:e, h] @ k[s:e, h // grp].T) * np.float32(d ** -0.5), -np.inf)
                p = np.exp(sc - sc.max(-1, keepdims=True)); want[s:e, h] = (p / p.sum(-1, keepdims=True)) @ v[s:e, h // grp]
        err = np.abs(got - want.reshape(T, nq * d)).max()
        check("attn", err < 3e-2, (d, nq, nkv, lens, float(err)))
        # ---- search
        N_ = int(rng.choice([300, 4097, 5000, 20000, 70001])); D = int(rng.choice([32, 64, 96, 128, 192, 256, 320, 512, 1024, 1088, 1536, 2048])); Q = int(rng.choice([1, 3, 17, 33, 100, 130, 150, 200, 256, 257, 300, 513, 700, 1024, 1100])); kk = int(rng.choice([1, 5, 100, 257]))      # (D >= 1024 with more than 256 queries: wide chunks, both passes on the GEMM kernel)
        X = O.l2_normalize(rng.standard_normal((N_, D)).astype(np.float32)) * rng.uniform(0.2, 2.0, size=(N_, 1)).astype(np.float32)
        qq = rng.standard_normal((Q, D)).astype(np.float32)
        idx = FlatIPIndex(D, capacity=N_); idx.shadow_f16 = bool(rng.integers(0, 4)); idx.add(X)
        setattr(idx, "search_flags", int(rng.integers(0, 4)))       # 0 auto, 1 score-matrix filter, 2 score-free filter, 3 same without the GEMM pass: same hits
        Dg, Ig = idx.search(qq, kk)
        setattr(idx, "search_flags", 1)
        D1, I1 = idx.search(qq, kk)
        setattr(idx, "search_flags", 0)
        check("search modes", bool(torch.equal(Dg, D1) and torch.equal(Ig, I1)), (N_, D, Q, kk))
        Dg, Ig = Dg.cpu().numpy(), Ig.cpu().numpy()
        Do, Io = O.flat_ip_topk(qq, X, kk)
        valid = Io >= 0
        ok = np.allclose(Dg[valid], Do[valid], atol=2e-5, rtol=2e-5) and (Ig[~valid] == -1).all()
        mism = (Ig != Io) & valid
        if ok and mism.any():
            qi, ri = np.nonzero(mism)
            ok = np.abs(np.einsum("ij,ij->i", qq[qi], X[Ig[qi, ri]]) - Do[qi, ri]).max() < 5e-5
        check("search", ok, (N_, D, Q, kk, float(mism.mean())))
    # ---- EmbeddingBag lookup (bits of the sequential fp32 sum) and the shard merge
    from lightretriever_amd.index import merge_topk
    for r in range(a.rounds):
        V = int(rng.integers(5, 400)); H = 4 * int(rng.integers(1, 600)) if rng.random() < 0.8 else int(rng.integers(1, 300)); nb = int(rng.integers(1, 40))
        table = rng.standard_normal((V, H)).astype(np.float32)
        lens = rng.integers(0, 300 if rng.random() < 0.1 else 24, size=nb)
        ids = rng.integers(0, V, size=int(lens.sum())).astype(np.int64)
        pad = int(rng.integers(0, V)) if rng.random() < 0.5 else None
        offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int64)
        od = H if rng.random() < 0.6 else max(1, (H // 2) & ~3) if H >= 8 else H
        got = ops.embedding_bag_mean(torch.from_numpy(table).cuda(), torch.from_numpy(ids).cuda(), torch.from_numpy(offs).cuda(), padding_idx=pad, out_dim=od).cpu().numpy()
        want = O.embedding_bag_mean(table, ids, offs, pad)[:, :od]
        check("embedding_bag", np.array_equal(got, want), (V, H, nb, od, pad))
        R, Qm, km = int(rng.integers(1, 9)), int(rng.integers(1, 20)), int(rng.integers(1, 300))
        Dp = rng.standard_normal((R, Qm, km)).astype(np.float32); Dp[rng.random(Dp.shape) < 0.1] = np.float32(0.25)      # ties
        Ip = np.stack([np.stack([rng.choice(100000, size=km, replace=False) for _ in range(Qm)]) for _ in range(R)]).astype(np.int64)
        Ip[rng.random(Ip.shape) < 0.05] = -1
        Dp = -np.sort(-Dp, axis=2)
        gD, gI = merge_topk(torch.from_numpy(Dp).cuda(), torch.from_numpy(Ip).cuda())
        wD, wI = O.merge_topk(list(Dp), list(Ip), km)
        check("merge_topk", np.array_equal(gI.cpu().numpy(), wI) and np.array_equal(gD.cpu().numpy()[wI >= 0], wD[wI >= 0]), (R, Qm, km))
    # ---- encoder end to end vs the oracle (small random architectures)
    from dataclasses import asdict
    from lightretriever_amd import EncoderConfig, LrxEncoder
    for r in range(max(1, a.rounds // 6)):
        d = int(rng.choice([64, 128])); nkv = int(rng.choice([1, 2])); grp = int(rng.choice([1, 2, 4])); nq = nkv * grp
        H = int(rng.choice([128, 256, 512])); I = 64 * int(rng.integers(1, 9)); L = int(rng.integers(1, 4))
        cfg = O.EncoderConfig(vocab_size=400, hidden_size=H, num_layers=L, num_q_heads=nq, num_kv_heads=nkv, head_dim=d, intermediate_size=I,
                              rms_eps=1e-5 if rng.random() < 0.5 else 1e-6, rope_theta=float(rng.choice([10000.0, 500000.0, 1e6])),
                              rope_type=str(rng.choice(["default", "llama3"])), rope_factor=8.0, rope_original_max_position=64,
                              qkv_bias=bool(rng.random() < 0.5), max_positions=256)
        w = O.random_weights(cfg, seed=int(rng.integers(0, 1 << 30)), std=0.05)
        pool = str(rng.choice(O.POOLING_STRATEGIES))                # (the x-to-last strategies need >= 3 tokens: the reference asserts on shorter rows)
        lens = [int(x) for x in rng.integers(3 if "to_last" in pool else 1, 200, size=int(rng.integers(1, 7)))]
        ids = rng.integers(0, 400, size=sum(lens)).astype(np.int32); cu = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        fold = bool(rng.random() < 0.7)
        prec = bool(rng.random() < 0.6)                             # the fp32 residual stream of deep backbones
        ops16 = str(rng.choice(["fp16", "fp16_qkv", "bf16"])) if prec else None     # the fp32 stream's GEMM operands
        enc = LrxEncoder(EncoderConfig(**asdict(cfg), fold_norm=fold, precise_stream=prec, operand_dtype=ops16), {k: torch.from_numpy(v) for k, v in w.items()})
        shrink = int(rng.choice([H, H // 2]))
        got = enc.encode_packed(torch.from_numpy(ids).cuda(), torch.from_numpy(cu).cuda(), max(lens), out_dim=shrink, pooling=pool).cpu().numpy()
        want = O.encode_passage(cfg, w, ids, cu, dense_shrink_dim=shrink, pooling=pool)
        cos = (got * want).sum(-1)
        check("encoder", cos.min() > 1 - 6e-3 and np.isfinite(got).all(), (H, I, L, nq, nkv, d, cfg.rope_type, cfg.qkv_bias, fold, prec, ops16, pool, lens, float(1 - cos.min())))
    # ---- sparse max aggregation, hit-list fusion, shared-prefix encode
    from lightretriever_amd.score_fuse_utils import fuse_hits
    BF16_ULP = 2.0 ** -7
    for r in range(max(1, a.rounds // 4)):
        lens = [int(x) for x in rng.integers(1, 400, size=int(rng.integers(1, 9)))]
        T = sum(lens); H = 64 * int(rng.integers(1, 9)); V = int(rng.integers(20, 700))
        hid = O.round_bf16(rng.standard_normal((T, H)).astype(np.float32)); W = O.round_bf16(rng.standard_normal((V, H)).astype(np.float32) * 0.1)
        cu = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32); tm = rng.random(T) < rng.uniform(0.1, 1.0)
        got = ops.sparse_max_aggregate(bf(hid), bf(W), torch.from_numpy(cu).cuda(), torch.from_numpy(tm.astype(np.uint8)).cuda()).cpu().numpy()
        want = O.max_aggregate_packed(hid, cu, tm, W, None, bf16=True)
        empty = want == O.BF16_MIN
        ok = np.array_equal(got == O.BF16_MIN, empty) and (np.abs(got[~empty] - want[~empty]) <= BF16_ULP * np.abs(want[~empty]) + 1e-6).all()
        check("maxagg", ok, (lens, H, V))
        Qf, k1, k2, Nf = int(rng.integers(1, 12)), int(rng.integers(1, 900)), int(rng.integers(1, 900)), 3000
        sysl = []
        for kk_ in (k1, k2):
            ids_ = np.stack([rng.choice(Nf, size=kk_, replace=False) for _ in range(Qf)]).astype(np.int64)
            sc_ = rng.standard_normal((Qf, kk_)).astype(np.float32)
            ids_[rng.random((Qf, kk_)) < 0.05] = -1
            sysl.append((sc_, ids_))
        method = str(rng.choice(["rrf", "linear"]))
        fs, fi, fc = fuse_hits([(torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()) for x, y in sysl], method=method, k=60, weights=[0.7, 0.3])
        fs, fi, fc = fs.cpu().numpy(), fi.cpu().numpy(), fc.cpu().numpy()
        dicts = [{str(q_): {str(int(p_)): float(v_) for p_, v_ in zip(y[q_], x[q_]) if p_ >= 0} for q_ in range(Qf) if (y[q_] >= 0).any()} for x, y in sysl]
        wantf = O.fuse_scores_rrf(dicts, k=60) if method == "rrf" else O.fuse_scores_linear(dicts, [0.7, 0.3])
        okf = True
        for q_ in range(Qf):
            wq = wantf.get(str(q_), {})
            gq = {str(int(p_)): float(v_) for p_, v_ in zip(fi[q_, :fc[q_]], fs[q_, :fc[q_]])}
            okf &= gq == wq
        check("fuse", okf, (Qf, k1, k2, method))
    for r in range(max(1, a.rounds // 15)):
        d = int(rng.choice([64, 128])); nkv = int(rng.choice([1, 2])); grp = int(rng.choice([1, 2, 4])); nq = nkv * grp; H = nq * d
        cfg = O.EncoderConfig(vocab_size=300, hidden_size=H, num_layers=int(rng.integers(1, 3)), num_q_heads=nq, num_kv_heads=nkv, head_dim=d, intermediate_size=128,
                              rope_type="default", qkv_bias=bool(rng.random() < 0.5), max_positions=128)
        w = O.random_weights(cfg, seed=int(rng.integers(0, 1 << 30)), std=0.05)
        enc = LrxEncoder(EncoderConfig(**asdict(cfg)), {k: torch.from_numpy(v) for k, v in w.items()})
        P1, S2, n = int(rng.integers(0, 40)), int(rng.integers(1, 4)), int(rng.integers(1, 50))
        pre = rng.integers(0, 300, size=P1).astype(np.int32); suf = rng.integers(0, 300, size=(n, S2)).astype(np.int32)
        got = enc.encode_prefixed(torch.from_numpy(pre).cuda(), torch.from_numpy(suf).cuda()).cpu().numpy()
        full = np.concatenate([np.concatenate([pre, suf[i]]) for i in range(n)]).astype(np.int32)
        cuf = (np.arange(n + 1) * (P1 + S2)).astype(np.int32)
        want = enc.encode_packed(torch.from_numpy(full).cuda(), torch.from_numpy(cuf).cuda(), P1 + S2, normalize=False).cpu().numpy()
        cosv = (got * want).sum(-1) / (np.linalg.norm(got, axis=-1) * np.linalg.norm(want, axis=-1))
        check("prefixed", cosv.min() > 0.999, (H, d, nq, nkv, P1, S2, n, float(1 - cosv.min())))
    print("fuzz rounds", a.rounds, "failures", fails)
    sys.exit(1 if fails else 0)

if __name__ == "__main__":
    main()
