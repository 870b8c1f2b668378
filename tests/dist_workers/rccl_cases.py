#!/usr/bin/env python3
"""One rank of an R-rank RCCL run of the sharded search path (launched by tests/test_gpu_00_multi_gpu.py through
`python -m torch.distributed.run --nproc-per-node R`; R = 1 on a one-GPU box -- with the exchange forced -- R = 2, 4, 8 where the GPUs exist).

Every case compares, ON EVERY RANK, the result of the distributed path with the result of one process holding everything, bit for bit; a
failing assertion raises, the rank exits non-zero and torch.distributed.run takes the others down and reports the failure.  What runs is
what replaces the reference's sharded GPU index (retriever/faiss_index.py:60-70) and its RPC fan-out (inference/exact_search_torchrpc.py:
243-328): row shards, one all-gather of [Q, k] wire words, on-device merge (lightretriever_amd/sharded.py).

    cases: sharded (contiguous shards of a non-dividing row count + interleaved row maps; k = 100 and 1000)
           lanes   (pipeline.SearchLanes over the communicator: two searches in flight per rank)
           hybrid  (HybridSearch.search through the B1 entry point, chunks sharded over the ranks)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np
import torch
import torch.distributed as dist


def corpus_rows(n, d, seed):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((n, d), dtype=np.float32)
    X /= np.linalg.norm(X, axis=1, keepdims=True)
    X[n // 3] = X[n // 7]                    # an exact duplicate pair in different shards: the tie rule (lower global row first) crosses ranks
    return X


def case_sharded(rank, world, dev):
    from bench import shard_split
    from lightretriever_amd import FlatIPIndex
    from lightretriever_amd.sharded import ShardedFlatIPIndex, local_to_global_rows
    N, D, Q = 50_003, 256, 37                 # 50 003 divides by none of 2, 4, 8
    X = corpus_rows(N, D, 5)
    q = torch.from_numpy(corpus_rows(Q, D, 6)).to(dev)
    q[3] = torch.from_numpy(X[N // 7]).to(dev)   # a query equal to the duplicated row: two hits with the same score
    full = FlatIPIndex(D, capacity=N, device=dev)
    full.add(torch.from_numpy(X))
    for k in (100, 1000):
        Dw, Iw = full.search(q, k)
        # (a) contiguous shards, the remainder rows on the first ranks (bench.shard_split): ids = id_base + local row
        rows, base = shard_split(N, rank, world)
        sh = FlatIPIndex(D, capacity=rows, device=dev, id_base=base)
        sh.add(torch.from_numpy(X[base:base + rows]))
        Dg, Ig = ShardedFlatIPIndex(sh).search(q, k)
        assert torch.equal(Ig, Iw) and torch.equal(Dg, Dw), ("contiguous shards", k, rank, (Ig != Iw).sum().item())
        # (b) interleaved batches (batch j of 64 rows -> rank j % world) with a local -> global row map, the layout of the chunk loop
        gl = local_to_global_rows(N, 64, rank, world)
        sh2 = FlatIPIndex(D, capacity=len(gl), device=dev)
        sh2.add(torch.from_numpy(X[gl.numpy()]))
        Dg2, Ig2 = ShardedFlatIPIndex(sh2, row_map=gl.to(dev)).search(q, k)
        assert torch.equal(Ig2, Iw) and torch.equal(Dg2, Dw), ("interleaved shards", k, rank, (Ig2 != Iw).sum().item())
    assert Iw[3, 0].item() == N // 7 and Iw[3, 1].item() == N // 3          # the duplicate pair really is the top of query 3, lower row first
    # (c) round 6 -- a WIDE chunk over the exchange: 600 queries over a 1024-wide shadow are one GEMM pass per rank (three query n-tiles); the
    # wire words of all 600 queries come from the chain's last kernels (per 128-query group), the row map applied
    N2, D2, Q2, k2 = 40_003, 1024, 600, 50
    X2 = corpus_rows(N2, D2, 8)
    q2 = torch.from_numpy(corpus_rows(Q2, D2, 9)).to(dev)
    full2 = FlatIPIndex(D2, capacity=N2, device=dev)
    full2.add(torch.from_numpy(X2))
    assert int(full2.lib.lrx_flat_ip_bounded_chunk_queries(N2, D2, Q2, k2, 0, 1)) == 608
    Dw2, Iw2 = full2.search(q2, k2)
    gl2 = local_to_global_rows(N2, 64, rank, world)
    sh3 = FlatIPIndex(D2, capacity=len(gl2), device=dev)
    sh3.add(torch.from_numpy(X2[gl2.numpy()]))
    Dg3, Ig3 = ShardedFlatIPIndex(sh3, row_map=gl2.to(dev)).search(q2, k2)
    assert torch.equal(Ig3, Iw2) and torch.equal(Dg3, Dw2), ("wide chunk, interleaved shards", rank, (Ig3 != Iw2).sum().item())
    return "sharded: %d rows over %d rank(s), k = 100 and 1000, contiguous + interleaved: bit-identical to one index" % (N, world)


def case_lanes(rank, world, dev):
    from bench import shard_split
    from lightretriever_amd import FlatIPIndex
    from lightretriever_amd.pipeline import SearchLanes
    from lightretriever_amd.sharded import ShardedFlatIPIndex
    N, D = 40_001, 256
    X = corpus_rows(N, D, 15)
    rows, base = shard_split(N, rank, world)
    sh = FlatIPIndex(D, capacity=rows, device=dev, id_base=base)
    sh.add(torch.from_numpy(X[base:base + rows]))
    shd = ShardedFlatIPIndex(sh)
    qs = [torch.from_numpy(corpus_rows(20 + i, D, 30 + i)).to(dev) for i in range(6)]
    one = [shd.search(q, 50) for q in qs]                                        # one at a time on the caller's stream
    one = [(d.clone(), i.clone()) for d, i in one]
    lanes = SearchLanes(shd, lanes=2)
    pend = [lanes.submit(q, 50) for q in qs]                                     # two in flight: the lanes' all-gathers alternate on the communicator
    for (dw, iw), p in zip(one, pend):
        dg, ig = p.result()
        assert torch.equal(ig, iw) and torch.equal(dg, dw), ("lanes", rank)
    lanes.drain()
    torch.cuda.synchronize()
    return "lanes: 6 searches over 2 lanes and %d rank(s): bit-identical to one at a time" % world


def case_hybrid(rank, world, dev, want):
    from test_gpu_api import build_stack, synth_corpus
    from helpers import load_model_golden
    from lightretriever_amd.retriever import HybridSearch
    cfg_o, w, _, _, _, _ = load_model_golden("llama_small_d64")
    tok, enc, hm, model = build_stack(cfg_o, w)
    corpus = synth_corpus(np.random.default_rng(0), 150)
    queries = {"q0": "capital of france", "q1": "dense retrieval models", "q2": "amd instinct memory", "d7": corpus["d7"]["text"]}
    res = HybridSearch(model, batch_size=16, corpus_chunk_size=70).search(corpus, queries, top_k=12, ignore_identical_ids=True)
    if want is None:
        return res
    assert set(res) == set(want) and "d7" not in res["d7"]
    for qid in want:
        assert set(res[qid]) == set(want[qid]), (qid, rank)
        for pid, sc in want[qid].items():
            assert abs(res[qid][pid] - sc) < 1e-6, (qid, pid, rank)
    return "hybrid: HybridSearch.search over %d rank(s) == single process (ids, scores)" % world


def main():
    cases = sys.argv[1].split(",") if len(sys.argv) > 1 else ["sharded", "lanes", "hybrid"]
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    want = case_hybrid(rank, world, dev, None) if "hybrid" in cases else None       # BEFORE the process group exists: the single-process answer
    dist.init_process_group("nccl", device_id=dev)
    if world == 1:
        os.environ["LRX_FORCE_COLLECTIVE"] = "1"                                     # one-rank rehearsal: the exchange still runs its collective
    notes = []
    try:
        sizes = torch.empty(world, dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(sizes, torch.tensor([rank], dtype=torch.int64, device=dev))
        assert sizes.tolist() == list(range(world))
        if "sharded" in cases:
            notes.append(case_sharded(rank, world, dev))
        if "lanes" in cases:
            notes.append(case_lanes(rank, world, dev))
        if "hybrid" in cases:
            notes.append(case_hybrid(rank, world, dev, want))
        dist.barrier()
        torch.cuda.synchronize()
    finally:
        dist.destroy_process_group()
    if rank == 0:
        for n in notes:
            print("RCCL_CASE_OK", n, flush=True)
        print("RCCL_ALL_OK ranks=%d" % world, flush=True)


if __name__ == "__main__":
    main()
