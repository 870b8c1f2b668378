#!/usr/bin/env python3
"""End-to-end retrieval agreement: the SAME corpus and the SAME queries through three complete pipelines on one GPU

    lrx       lrx_encode_packed (default stream mode) -> FlatIPIndex           (this build)
    hf_fp32   HF transformers model, fp32 weights and activations -> FlatIPIndex   (the reference's arithmetic, exact)
    hf_bf16   the same model cast to bf16 -> FlatIPIndex                            (what the reference executes under --bf16)

each producing its OWN document embeddings (finetune/modeling_hybrid.py:205-278: LM forward, lasttoken pooling, normalise) and its OWN
query vectors, of both kinds the reference serves:

    emb     asymmetric: EmbeddingBag mean over a table the pipeline built itself from `[bos] prompt tok [eos]` sequences
            (finetune/nonctx_emb_utils.py:239-313, modeling_hybrid.py:472-490)
    dense   symmetric: the query tokens through the LM, lasttoken pooling, normalise (modeling_hybrid.py:363-401)

and the hits of retriever/faiss_index.py:27-40 (exact inner product, top-k).  Reported per query kind: overlap@100, overlap@10, the
fraction of top-10 POSITIONS that hold the same document, top-1 agreement -- of lrx and of hf_bf16 against hf_fp32.  Library of
tests/test_gpu_recall.py and a CLI:

    python tools/recall_probe.py [--docs 20000] [--queries 200] [--preset llama32_1b] [--seed 0] [--out gpurun_out/r05_recall.jsonl]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from parity_margin import hf_model_for


def synthetic_corpus(cfg, n_docs, n_queries, seed, sub_vocab=4096, max_len=512, sink=None):
    """Ragged documents (lengths clip(lognormal(4.8, 0.7), 16, max_len - 1) + the eos the encoders append, sink token first) over a
    sub-vocabulary of `sub_vocab` token ids; every query = 8..32 tokens drawn from one document (so a query shares its bag of tokens with
    at least that document)."""
    rng = np.random.default_rng(4242 + seed)
    lens = np.clip(rng.lognormal(4.8, 0.7, size=n_docs), 16, max_len - 1).astype(np.int64)
    lo = 2000
    docs = [rng.integers(lo, lo + sub_vocab, size=int(l)).astype(np.int64) for l in lens]
    if sink is not None:
        for d in docs:
            d[0] = sink
    src = rng.integers(0, n_docs, size=n_queries)
    queries = []
    for s in src:
        body = docs[s][1:]
        n = int(rng.integers(8, 33))
        queries.append(rng.choice(body, size=min(n, len(body)), replace=False).astype(np.int64))
    return docs, queries, lo


def by_length(seqs):
    """{length: [indices]}: sequences of one length form a batch without padding (no mask: exactly the causal forward)"""
    groups = {}
    for i, s in enumerate(seqs):
        groups.setdefault(len(s), []).append(i)
    return groups


@torch.no_grad()
def hf_encode(hf, seqs, prefix=None, suffix=None, normalize=True, max_batch_tokens=32768):
    """lasttoken-pooled final hidden state of `prefix + seq + suffix` for every seq -> fp32 [n, H] (normalised on request)"""
    dev = next(hf.parameters()).device
    H = hf.config.hidden_size
    out = torch.empty(len(seqs), H, dtype=torch.float32, device=dev)
    pre = [] if prefix is None else list(prefix)
    suf = [] if suffix is None else list(suffix)
    for L, idx in sorted(by_length(seqs).items()):
        Lf = L + len(pre) + len(suf)
        per = max(1, max_batch_tokens // Lf)
        for s in range(0, len(idx), per):
            part = idx[s:s + per]
            ids = torch.tensor(np.stack([np.concatenate([pre, seqs[i], suf]).astype(np.int64) for i in part]), device=dev)
            h = hf(input_ids=ids, use_cache=False).last_hidden_state[:, -1].float()
            out[part] = torch.nn.functional.normalize(h, dim=-1) if normalize else h
    return out


def lrx_encode(enc, seqs, prefix=None, suffix=None, normalize=True, max_batch_tokens=131072, out=None):
    """the same through lrx_encode_packed, `max_batch_tokens` packed tokens per call, rows in input order"""
    pre = np.asarray([] if prefix is None else prefix, dtype=np.int64)
    suf = np.asarray([] if suffix is None else suffix, dtype=np.int64)
    full = [np.concatenate([pre, s, suf]) for s in seqs]
    if out is None:
        out = torch.empty(len(seqs), enc.cfg.hidden_size, dtype=torch.float32, device=enc.device)
    s = 0
    while s < len(full):
        e, tok = s, 0
        while e < len(full) and (e == s or tok + len(full[e]) <= max_batch_tokens):
            tok += len(full[e])
            e += 1
        lens = [len(x) for x in full[s:e]]
        ids = torch.from_numpy(np.concatenate(full[s:e]).astype(np.int32)).to(enc.device)
        cu = torch.tensor(np.concatenate([[0], np.cumsum(lens)]), dtype=torch.int32, device=enc.device)
        enc.encode_packed(ids, cu, max(lens), out=out[s:e], normalize=normalize)
        s = e
    return out


def bag_queries(table, queries, lo):
    """nn.EmbeddingBag(mean) over table rows (token - lo), then F.normalize: modeling_hybrid.py:472-490 with plain torch ops"""
    flat = torch.tensor(np.concatenate(queries) - lo, dtype=torch.int64, device=table.device)
    offs = torch.tensor(np.cumsum([0] + [len(q) for q in queries[:-1]]), dtype=torch.int64, device=table.device)
    return torch.nn.functional.normalize(torch.nn.functional.embedding_bag(flat, table, offs, mode="mean"), dim=-1)


def agreement(I_a, I_ref):
    """hit lists [Q, 100] (ranked) -> overlap@100, overlap@10, same-document-at-position fraction over the top 10, top-1 agreement"""
    a, r = I_a.cpu().numpy(), I_ref.cpu().numpy()
    Q = a.shape[0]
    o100 = np.array([len(set(a[i]) & set(r[i])) / a.shape[1] for i in range(Q)])
    o10 = np.array([len(set(a[i, :10]) & set(r[i, :10])) / 10.0 for i in range(Q)])
    pos10 = (a[:, :10] == r[:, :10]).mean(axis=1)
    return {"overlap_at_100": float(o100.mean()), "overlap_at_100_min": float(o100.min()), "overlap_at_10": float(o10.mean()),
            "top10_same_position": float(pos10.mean()), "top10_identical_order_queries": float((pos10 == 1.0).mean()),
            "top1": float((a[:, 0] == r[:, 0]).mean())}


def measure(preset="llama32_1b", n_docs=20000, n_queries=200, seed=0, profile="trained_like", k=100, sub_vocab=4096, prompt_len=20, mrl_dim=256, log=print) -> dict:
    from lightretriever_amd import EncoderConfig, FlatIPIndex, LrxEncoder, ops
    from lightretriever_amd.synth import sink_token
    cfg = getattr(EncoderConfig, preset)()
    t0 = time.time()
    enc = LrxEncoder.random_init(cfg, seed=seed, profile=profile)
    sink = sink_token(cfg) if profile == "trained_like" else 1
    docs, queries, lo = synthetic_corpus(cfg, n_docs, n_queries, seed, sub_vocab=sub_vocab, sink=sink)
    rng = np.random.default_rng(77 + seed)
    prefix = np.concatenate([[sink], rng.integers(lo, lo + sub_vocab, size=prompt_len)]).astype(np.int64)     # [bos] + a 20-token instruction
    eos = lo - 1
    vocab = [np.array([t], dtype=np.int64) for t in range(lo, lo + sub_vocab)]
    dev = enc.device
    rec = {"preset": preset, "profile": profile, "seed": seed, "docs": n_docs, "queries": n_queries, "k": k, "sub_vocab": sub_vocab,
           "mean_doc_tokens": float(np.mean([len(d) for d in docs])), "stream": "precise_fp32" if enc.precise else "bf16_folded_norm"}
    log("recall: weights + corpus in %.1f s" % (time.time() - t0))

    def search(X, q):
        idx = FlatIPIndex(X.shape[1], capacity=X.shape[0], device=dev)
        idx.add(X)
        D, I = idx.search(q.contiguous(), k)
        return D.clone(), I.clone()

    # ---- lrx: documents straight into the index shard; the table through the shared-prefix build; dense queries through the same operator
    t0 = time.time()
    idx = FlatIPIndex(cfg.hidden_size, capacity=n_docs, device=dev)
    lrx_encode(enc, docs, suffix=[eos], out=idx.append_slot(n_docs))
    idx.commit(n_docs)
    suf = torch.tensor([[t, eos] for t in range(lo, lo + sub_vocab)], dtype=torch.int32, device=dev)
    table_lrx = enc.encode_prefixed(torch.tensor(prefix, dtype=torch.int32, device=dev), suf, normalize=False).clone()
    flat = torch.tensor(np.concatenate(queries) - lo, dtype=torch.int64, device=dev)
    offs = torch.tensor(np.cumsum([0] + [len(q) for q in queries[:-1]]), dtype=torch.int64, device=dev)
    q_emb_lrx = ops.embedding_bag_mean(table_lrx, flat, offs, normalize=True)
    q_den_lrx = lrx_encode(enc, queries, prefix=prefix, suffix=[eos])
    hits = {"lrx": {"emb": idx.search(q_emb_lrx, k)[1].clone(), "dense": idx.search(q_den_lrx, k)[1].clone()}}
    X_lrx = idx.vectors.clone()

    def mrl(x):                                   # dense_shrink_dim: slice, then normalise (modeling_hybrid.py:274-277, :487-490)
        return torch.nn.functional.normalize(x[:, :mrl_dim].float(), dim=-1).contiguous()
    # BASELINE configs[4]: MRL-256 embeddings -- the first 256 dims of the un-normalised pooled state, renormalised; slicing the normalised rows and
    # renormalising is the same vector.  (lrx: the kernel's own out_dim path is held to this by tests/test_gpu_encoder.py.)
    hits["lrx"]["emb_mrl"] = search(mrl(X_lrx), mrl(q_emb_lrx))[1]
    hits["lrx"]["dense_mrl"] = search(mrl(X_lrx), mrl(q_den_lrx))[1]
    torch.cuda.synchronize()
    rec["lrx_seconds"] = round(time.time() - t0, 2)
    sd = enc.hf_state_dict()
    del enc, idx
    torch.cuda.empty_cache()

    # ---- HF fp32, then the same module cast to bf16 (what the reference runs): documents, table, both query kinds, FlatIPIndex
    hf = hf_model_for(cfg)
    missing, unexpected = hf.load_state_dict({k_: v.float() for k_, v in sd.items()}, strict=False)
    assert not unexpected and all("rotary" in m for m in missing), (missing, unexpected)
    del sd
    X = {}
    for name, dt in (("hf_fp32", torch.float32), ("hf_bf16", torch.bfloat16)):
        t0 = time.time()
        hf = hf.to(dt)
        X[name] = hf_encode(hf, docs, suffix=[eos])
        table = hf_encode(hf, vocab, prefix=prefix, suffix=[eos], normalize=False)
        q_emb = bag_queries(table, queries, lo)
        q_den = hf_encode(hf, queries, prefix=prefix, suffix=[eos])
        hits[name] = {"emb": search(X[name], q_emb)[1], "dense": search(X[name], q_den)[1],
                      "emb_mrl": search(mrl(X[name]), mrl(q_emb))[1], "dense_mrl": search(mrl(X[name]), mrl(q_den))[1]}
        if name == "hf_fp32":
            # how hard the ranking problem is: the fp32 pipeline's score gap between its hits k and k + 1, and between hits 10 and 11
            for kind, qq in (("emb", q_emb), ("dense", q_den)):
                idx = FlatIPIndex(cfg.hidden_size, capacity=n_docs, device=dev)
                idx.add(X[name])
                Dg = idx.search(qq.contiguous(), k + 1)[0]
                rec["fp32_gap_" + kind] = {"k_to_k1_median": float((Dg[:, k - 1] - Dg[:, k]).median()), "10_to_11_median": float((Dg[:, 9] - Dg[:, 10]).median()),
                                           "top1_score_median": float(Dg[:, 0].median()), "kth_score_median": float(Dg[:, k - 1].median())}
                del idx
            rec["table_lrx_vs_fp32_max_1mcos"] = float((1 - torch.nn.functional.cosine_similarity(table_lrx.double(), table.double(), dim=-1)).max())
            rec["q_emb_lrx_vs_fp32_max_1mcos"] = float((1 - (q_emb_lrx.double() * q_emb.double()).sum(-1)).max())
            rec["q_dense_lrx_vs_fp32_max_1mcos"] = float((1 - (q_den_lrx.double() * q_den.double()).sum(-1)).max())
        torch.cuda.synchronize()
        rec[name + "_seconds"] = round(time.time() - t0, 2)
        log("recall: %s in %.1f s" % (name, time.time() - t0))
    del hf
    torch.cuda.empty_cache()
    rec["doc_lrx_vs_fp32_max_1mcos"] = float((1 - (X_lrx.double() * X["hf_fp32"].double()).sum(-1)).max())
    rec["doc_hfbf16_vs_fp32_max_1mcos"] = float((1 - (X["hf_bf16"].double() * X["hf_fp32"].double()).sum(-1)).max())
    rec["mrl_dim"] = mrl_dim
    for kind in ("emb", "dense", "emb_mrl", "dense_mrl"):
        rec[kind] = {"lrx_vs_fp32": agreement(hits["lrx"][kind], hits["hf_fp32"][kind]),
                     "hfbf16_vs_fp32": agreement(hits["hf_bf16"][kind], hits["hf_fp32"][kind]),
                     "lrx_vs_hfbf16": agreement(hits["lrx"][kind], hits["hf_bf16"][kind])}
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--preset", default="llama32_1b")
    ap.add_argument("--docs", type=int, default=20000)
    ap.add_argument("--queries", type=int, default=200)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--profile", default="trained_like")
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    rec = measure(a.preset, a.docs, a.queries, a.seed, a.profile)
    line = json.dumps(rec)
    print(line, flush=True)
    if a.out:
        with open(a.out, "a") as f:
            f.write(line + "\n")


if __name__ == "__main__":
    main()
