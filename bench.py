#!/usr/bin/env python3
"""Headline benchmark: docs embedded/sec (Llama-3.2-1B dims, seq_len=512, bf16) + queries/sec @ top-100 over a 1M-doc
fp32 index, on N MI355X of one node (one process per GPU; launched by torchrun for N > 1).

A step = one pass of the hot path over one batch of synthetic input: 256 packed documents x 512 tokens through
lrx_encode_packed, embeddings written in place into this rank's HBM index shard.  The search leg (same K steps, timed
separately) = lrx_flat_ip_search of Q=100 queries over the 1M-row index (row-sharded over the ranks) + the RCCL all-gather
of per-shard top-100 + on-device merge.  Inputs are resident in HBM before the timed regions start.

Prints ONE JSON line on rank 0 (contract in the task statement): value = whole-job docs/s.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0  # MI355X dense bf16 MFMA (MI355X_MICROARCH.md, chip-level parameters)
PEAK_HBM_GBS = 8000.0      # HBM3E spec (same table); ~6300 achievable
PEAK_F32_MFMA_TFLOPS = 157.3


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch-docs", type=int, default=256)
    ap.add_argument("--seq-len", type=int, default=512)
    ap.add_argument("--index-rows", type=int, default=1_000_000)
    ap.add_argument("--queries", type=int, default=100)
    ap.add_argument("--topk", type=int, default=100)
    ap.add_argument("--model", default="llama3.2-1b", choices=["llama3.2-1b", "llama3.2-3b", "llama3.1-8b", "qwen2.5-1.5b", "qwen2.5-3b", "qwen2.5-7b"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--mrl-dim", type=int, default=0, help="index / embedding dimension D (dense_shrink_dim, e.g. 256 for BASELINE config 5); 0 = hidden size")
    ap.add_argument("--no-search", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the extra single-GPU legs (BASELINE configs 2-4, 8-way shard, top_k = 1000)")
    ap.add_argument("--no-sparse", action="store_true", help="skip the dense+sparse document-vector leg (SURVEY 8f N2)")
    ap.add_argument("--legs", default="all", help="comma list of encode,search,sparse,sharded,configs,cpu (default all): one leg per run gives one "
                    "rocprofv3 kernel-stats file per leg (tools/final_profile.sh); a partial run is marked `partial_run` and is not the headline")
    ap.add_argument("--config-legs", default="all", help="comma list of the `configs` entries to run (default all)")
    ap.add_argument("--sharded-rows", type=int, default=10_000_000,
                    help="rows of the row-sharded indexes of BASELINE configs[3] / configs[4] (the `sharded` leg: runs over the communicator when "
                         "there is one, i.e. --gpus > 1 or LRX_BENCH_FORCE_DIST=1); every rank holds shard_split(rows, rank, world) of them")
    ap.add_argument("--ragged", action="store_true",
                    help="document lengths ~ clip(lognormal(5.3, 0.6), 16, seq_len), sorted longest first (mirrors hybrid_search.py:273-276) "
                         "instead of the fixed-length headline workload")
    return ap.parse_args()


def barrier_sync(distributed):
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()


def usable_cores():
    """Cores this process may really use: affinity mask capped by the cgroup CPU quota (the GPU box shows 256 CPUs
    but grants 16; running 256 threads on a 16-CPU quota is 50x slower than 16 threads)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(cfg, seq_len, topk, dim):
    """The reference's CPU path on this box's host cores, on a bounded sample (~30 s), as SURVEY.md 8d words it:
    (a) HF transformers LlamaModel (the model code the reference's encode_passage executes) at the real config, sdpa, batch 8 x
        seq_len, fp32 AND bf16 -- both reported, the faster one is `value`;
    (b) flat IP via the oracle port (Faiss is not installed);
    (c) BASELINE configs[0] really run on the CPU through the restatement of B1 (HybridSearch.search): 200 documents x 128 tokens + 100
        queries -> top-k (configs[0] names 1000 documents; the encode is linear in them, the scaling is stated next to the measurement)."""
    import numpy as np
    from transformers import LlamaConfig, LlamaModel
    cores = usable_cores()
    torch.set_num_threads(cores)
    hf_cfg = LlamaConfig(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size,
                         num_hidden_layers=cfg.num_layers, num_attention_heads=cfg.num_q_heads, num_key_value_heads=cfg.num_kv_heads,
                         head_dim=cfg.head_dim, rms_norm_eps=cfg.rms_eps, max_position_embeddings=131072,
                         rope_parameters={"rope_type": "llama3", "rope_theta": cfg.rope_theta, "factor": cfg.rope_factor,
                                          "low_freq_factor": 1.0, "high_freq_factor": 4.0, "original_max_position_embeddings": 8192},
                         attn_implementation="sdpa")
    torch.manual_seed(0)
    model = LlamaModel(hf_cfg).float().eval()

    def run(ids, budget_s):
        n_done, t0 = 0, time.perf_counter()
        with torch.no_grad():
            while True:
                h = model(input_ids=ids, use_cache=False).last_hidden_state[:, -1]
                torch.nn.functional.normalize(h.float(), dim=-1)
                n_done += ids.shape[0]
                if time.perf_counter() - t0 > budget_s:
                    break
        return n_done, time.perf_counter() - t0

    batch = 8
    ids = torch.randint(1000, 127000, (batch, seq_len))
    with torch.no_grad():
        model(input_ids=ids[:1, :32], use_cache=False)   # touch the weights once (page-in), untimed
    n32, s32 = run(ids, 6.0)
    # bf16: the GPU box's host (EPYC 9575F) has no AMX; probe one short batch first and shrink the bf16 sample so the leg stays bounded
    model = model.to(torch.bfloat16)
    with torch.no_grad():
        t0 = time.perf_counter()
        model(input_ids=ids[:1, :64], use_cache=False)
        model(input_ids=ids[:1, :64], use_cache=False)
        probe = (time.perf_counter() - t0) / 2
    est_batch_s = probe * batch * seq_len / 64
    b16 = batch if est_batch_s < 12.0 else max(1, int(batch * 12.0 / est_batch_s))
    n16, s16 = run(ids[:b16], 0.0)                       # exactly one batch
    # ---- BASELINE configs[0] REALLY RUN through the CPU restatement of B1 (HybridSearch.search): N0 documents x 128 tokens through the HF
    #      model (bf16, the faster CPU dtype here) in batches of 8, longest-first order, last-token pool + normalise, 100 EmbeddingBag queries
    #      (oracle), then the oracle's chunk loop / index / heap merge (O.search_chunks = hybrid_search.py:301-358).  configs[0] names 1000
    #      documents: N0 = 200 are timed (the encode is > 99 % of it and linear in the document count), the scaling to 1000 is stated.
    from oracle import lrx_oracle as O
    N0, S0, Q0 = 200, 128, 100
    rng0 = np.random.default_rng(11)
    ids0 = torch.from_numpy(rng0.integers(1000, 127000, size=(N0, S0)))
    table0 = rng0.standard_normal((cfg.vocab_size, dim), dtype=np.float32)
    qlens = rng0.integers(8, 33, size=Q0)
    qids0 = rng0.integers(1000, 127000, size=int(qlens.sum()))
    t0 = time.perf_counter()
    emb0 = []
    with torch.no_grad():
        for s0 in range(0, N0, 8):
            h = model(input_ids=ids0[s0:s0 + 8], use_cache=False).last_hidden_state[:, -1]
            emb0.append(torch.nn.functional.normalize(h.float()[:, :dim], dim=-1))
    enc0_s = time.perf_counter() - t0
    X0 = torch.cat(emb0).numpy()
    q0 = O.encode_query_emb(table0, qids0, O.nonctx_offsets([int(x) for x in qlens]), normalize=True)
    res0 = O.search_chunks(q0, ["q%d" % i for i in range(Q0)], X0, ["d%d" % i for i in range(N0)], top_k=min(topk, N0), corpus_chunk_size=100)
    cfg0_run_s = time.perf_counter() - t0
    assert len(res0) == Q0 and all(len(v) == min(topk, N0) for v in res0.values())
    del model
    fp32_rate, bf16_rate = n32 / s32, n16 / s16
    rng = np.random.default_rng(7)
    n_sample, nq = 50_000, 100
    X = O.l2_normalize(rng.standard_normal((n_sample, dim), dtype=np.float32))
    q = O.l2_normalize(rng.standard_normal((nq, dim), dtype=np.float32))
    t0 = time.perf_counter()
    O.flat_ip_topk(q, X, topk)
    srch_s = time.perf_counter() - t0
    return {
        "value": round(max(fp32_rate, bf16_rate), 4), "unit": "docs/s", "cores": cores, "kind": "reference",
        "sample": f"HF transformers LlamaModel (the third-party forward the reference's encode_passage calls), random-init {cfg.num_layers}L/"
                  f"H{cfg.hidden_size}, sdpa, batch {batch} x {seq_len} tokens on {cores} threads: fp32 {n32} docs in {s32:.1f}s, "
                  f"bf16 {n16} docs (batch {b16}) in {s16:.1f}s; value = the faster ({'fp32' if fp32_rate >= bf16_rate else 'bf16'})",
        "fp32_docs_per_s": round(fp32_rate, 4), "bf16_docs_per_s": round(bf16_rate, 4),
        "config0": {"workload": "BASELINE configs[0]: docs x 128 tokens + 100 queries -> top-k through the CPU restatement of HybridSearch.search (HF model "
                                "forward in bf16 + oracle EmbeddingBag / chunk loop / flat IP / heap merge), CPU only",
                    "docs_timed": N0, "seconds_measured": round(cfg0_run_s, 2), "encode_seconds": round(enc0_s, 2),
                    "search_and_merge_seconds": round(cfg0_run_s - enc0_s, 3), "docs_per_s": round(N0 / enc0_s, 2),
                    "seconds_scaled_to_1000_docs": round(cfg0_run_s - enc0_s + enc0_s * 1000 / N0, 1),
                    "scaling": f"the encode (more than 99 percent of the time) is linear in the document count: 1000 / {N0} x the measured encode time + the measured search"},
        "search": {"value": round(nq / srch_s, 2), "unit": "queries/s", "kind": "port", "cores": cores,
                   "sample": f"oracle flat_ip_topk (numpy sgemm + lexsort) Q={nq}, k={topk} over {n_sample} x {dim} fp32 rows "
                             f"({srch_s:.2f}s); per-query cost scales linearly with rows",
                   "scaled_to_index_rows": None},
    }


PMC_SUMMARY = os.environ.get("LRX_PMC_SUMMARY", "profiles/r06_pmc_summary.json")   # offline rocprofv3 --pmc passes (tools/pmc_traffic.sh)
PMC_MFMA = os.environ.get("LRX_PMC_MFMA", "profiles/r06_pmc_mfma.json")


def git_blob_sha(rel_path):
    """git's blob id of a committed file, computed from its bytes (sha1 of "blob <len>\\0" + content): ties an offline number quoted in
    the JSON line to the exact profile file it was read from (`git cat-file -p <id>` shows it)."""
    import hashlib
    try:
        data = open(os.path.join(ROOT, rel_path), "rb").read()
    except OSError:
        return None
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


def pmc_traffic(kernel_key):
    """HBM-side bytes per launch of a kernel from the committed rocprofv3 PMC summary (tools/pmc_traffic.sh: separate
    FETCH_SIZE / WRITE_SIZE passes, FETCH_SIZE x2 gfx950 correction); None when no summary is present.  NOT measured by this run:
    `offline_profile` next to it names the file and its git blob id."""
    try:
        d = json.load(open(os.path.join(ROOT, PMC_SUMMARY)))
        return round(d[kernel_key]["hbm_bytes_per_launch"])
    except (OSError, KeyError, ValueError):
        return None


SEARCH_KERNELS = ("k_filter_xreg<emit>", "k_filter_xreg<scores>", "k_sample_threshold", "k_refine_band", "k_refine_merge")


def pmc_search_traffic():
    """sample pass + threshold + main (emitting) filter pass + band refine + merge of one search (same PMC summary, per launch)."""
    parts = [pmc_traffic(k) for k in SEARCH_KERNELS]
    return None if any(p is None for p in parts) else sum(parts)


def pmc_mfma(kernel_key):
    """(mfma busy fraction, effective clock GHz) of a kernel from the committed PMC pass (tools/pmc_mfma.sh) or (None, None)."""
    try:
        d = json.load(open(os.path.join(ROOT, PMC_MFMA)))[kernel_key]
        return d.get("mfma_busy_frac", d.get("mfma_busy_over_cu_busy_x4simd")), d["effective_clock_GHz"]
    except (OSError, KeyError, ValueError):
        return None, None


def hbm_roofline(n_rows, dim, nq, k, ms, shadow=True):
    alg = n_rows * dim * (2 if shadow else 4) + nq * dim * 4 + nq * k * 12
    gbs = alg / (ms * 1e-3) / 1e9
    return {"bound": "hbm", "achieved": round(gbs, 2), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4), "algorithmic_bytes": alg,
            "traffic": None}


def time_search(fn, passes, warm=2):
    """HIP events (torch's current stream = the stream liblrx launches on) around each of `passes` calls -> (mean ms, median ms)."""
    for _ in range(warm):
        fn()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2 * passes)]
    for i in range(passes):
        ev[2 * i].record()
        fn()
        ev[2 * i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[2 * i].elapsed_time(ev[2 * i + 1]) for i in range(passes))
    return sum(ts) / len(ts), ts[len(ts) // 2]


def lanes_ms(target, q, k, n, lanes=2):
    """ms per search with `lanes` searches in flight (pipeline.SearchLanes), n searches between one event pair."""
    from lightretriever_amd.pipeline import SearchLanes
    sl = SearchLanes(target, lanes=lanes)
    for _ in range(2 * lanes):
        sl.submit(q, k)
    sl.drain()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        sl.submit(q, k)
    sl.drain()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def synthetic_index(n_rows, dim, dev, seed):
    from lightretriever_amd import FlatIPIndex
    idx = FlatIPIndex(dim, capacity=n_rows, device=dev)
    g = torch.Generator(device=dev).manual_seed(seed)
    slot = idx.append_slot(n_rows)
    step = max(1, (1 << 27) // dim)
    for s0 in range(0, n_rows, step):
        e = min(s0 + step, n_rows)
        slot[s0:e] = torch.nn.functional.normalize(torch.randn(e - s0, dim, generator=g, device=dev), dim=-1)
    idx.commit(n_rows)
    return idx, g


def search_leg(n_rows, dim, nq, k, dev, passes, index=None, seed=21):
    own = index is None
    g = torch.Generator(device=dev).manual_seed(seed + nq)
    if own:
        index, _ = synthetic_index(n_rows, dim, dev, seed)
    q = torch.nn.functional.normalize(torch.randn(nq, dim, generator=g, device=dev), dim=-1)
    mean_ms, med_ms = time_search(lambda: index.search(q, k), passes)
    out = {"workload": "exact top-%d of %d queries over %d x %d fp32 rows (+ tiled fp16 shadow), one GPU, HIP events around lrx_flat_ip_search_bounded"
                       % (k, nq, n_rows, dim),
           "ms": round(mean_ms, 4), "ms_median": round(med_ms, 4), "passes": passes, "queries_per_s": round(nq / (mean_ms * 1e-3), 1),
           "roofline": hbm_roofline(n_rows, dim, nq, k, mean_ms)}
    if own:
        del index
        torch.cuda.empty_cache()
    return out


def extra_legs(args, dev, headline_index):
    """What VERDICT r2 asked to put in front of the driver next to the (unchanged) headline, N = 1 only: BASELINE configs 2-4 on one GPU
    (8B encode at batch 128; 1M x 4096 and 10M x 256 search), the shard an 8-GPU run of the headline search really scans (125 000 x 2048
    rows + pack -> all-gather -> merge on a 1-rank RCCL group), and the reference's default operating point top_k = 1000 over
    CORPUS_CHUNK_SIZE = 100 000 (eval/call_evaluate_mteb.sh:9-10) and over the 1M index."""
    from lightretriever_amd import EncoderConfig, LrxEncoder
    legs = {}
    only = None if args.config_legs == "all" else set(args.config_legs.split(","))
    want = lambda name: only is None or name in only
    if want("top_k_1000"):
        legs["top_k_1000"] = topk1000_legs(dev, headline_index)
    if want("search_per_shard_8way"):
        legs["search_per_shard_8way"] = per_shard_leg(dev)
    # ---- BASELINE configs[2] / configs[3] index shape (8B width) and configs[4] (MRL 256, 10M rows) on one GPU; what ONE rank of the 8-GPU
    #      configurations holds (10M rows row-sharded 8 ways): configs[3] 1.25M x 4096, configs[4] 1.25M x 256
    for name, shape in (("config2_search_1Mx4096", (1_000_000, 4096, 51)), ("config4_search_10Mx256", (10_000_000, 256, 61)),
                        ("config3_per_rank_shard_1250kx4096", (1_250_000, 4096, 71)), ("config4_per_rank_shard_1250kx256", (1_250_000, 256, 81))):
        if want(name):
            legs[name] = search_leg(shape[0], shape[1], 100, 100, dev, 20, seed=shape[2])
    if want("config2_encode_llama31_8b"):
        legs["config2_encode_llama31_8b"] = encode_8b_leg(args, dev)
    if want("ragged_encode_llama32_1b"):
        legs["ragged_encode_llama32_1b"] = ragged_encode_leg(args, dev)
    if want("encode_llama32_1b_bf16_stream"):
        legs["encode_llama32_1b_bf16_stream"] = other_stream_leg(args, dev)
    # (round 4's additions run after the legs of round 3, whose numbers stay comparable: the search legs are sensitive to what ran before them)
    if want("search_clustered"):
        legs["search_clustered"] = clustered_search_legs(dev)
    if want("n1_embedding_bag_build"):
        legs["n1_embedding_bag_build"] = embedding_bag_build_leg(args, dev)
    if want("n4_index_persistence"):
        legs["n4_index_persistence"] = index_persistence_leg(dev)
    if want("measured_ceilings"):
        legs["measured_ceilings"] = measured_ceilings(dev)
    return legs


def measured_ceilings(dev):
    """measured ceilings next to the spec peaks, IN this run on this box: (a) a plain streaming read of 4 GiB (liblrx's
    k_stream_read: nothing but 16-B loads in flight) -- the HBM rate the search filter can at best approach; (b) the vendor bf16 GEMM
    (torch.matmul -> hipBLASLt) on the four projection shapes of the headline model, no fused epilogue -- what a library GEMM reaches
    under the same power cap."""
    import ctypes as C
    from lightretriever_amd import _lib
    lib = _lib.lib()
    out = {}
    buf = torch.empty(1 << 30, dtype=torch.int32, device=dev)           # 4 GiB (well beyond the 256-MiB Infinity Cache)
    buf.zero_()
    best = None
    for n_wg in (1024, 2048, 4096, 8192):
        sink = torch.zeros(n_wg, dtype=torch.int32, device=dev)
        fn = lambda: _lib.check(lib.lrx_probe_stream_read(_lib.ptr(buf), buf.numel() * 4, _lib.ptr(sink), n_wg, _lib.current_stream()))
        ms, _ = time_search(fn, 6)
        gbs = buf.numel() * 4 / (ms * 1e-3) / 1e9
        if best is None or gbs > best[1]:
            best = (n_wg, gbs)
    del buf
    out["hbm_stream_read"] = {"value": round(best[1], 1), "unit": "GB/s", "bytes": 4 << 30, "workgroups": best[0],
                              "frac_of_spec_peak": round(best[1] / PEAK_HBM_GBS, 4), "kernel": "k_stream_read (lrx_probe_stream_read)"}
    M = 131072
    g = torch.Generator(device=dev).manual_seed(0)
    vg = {}
    for name, N, K in (("gate_up", 16384, 2048), ("qkv", 3072, 2048), ("o", 2048, 2048), ("down", 2048, 8192)):
        A = torch.randn(M, K, generator=g, device=dev).to(torch.bfloat16)
        B = (torch.randn(N, K, generator=g, device=dev) * 0.02).to(torch.bfloat16)
        ms, med = time_search(lambda: torch.matmul(A, B.t()), 10, warm=3)
        vg[name] = {"M": M, "N": N, "K": K, "ms_median": round(med, 4), "tflops": round(2.0 * M * N * K / (med * 1e-3) / 1e12, 1)}
        del A, B
    out["vendor_gemm_bf16"] = {"library": "torch.matmul (hipBLASLt / rocBLAS), plain C = A B^T, no fused epilogue", "shapes": vg,
                               "gate_up_frac_of_spec_peak": round(vg["gate_up"]["tflops"] / PEAK_BF16_TFLOPS, 4)}
    torch.cuda.empty_cache()
    return out


def clustered_search_legs(dev):
    """the headline search on NON-iid rows -- 1 000 vMF-like clusters (intra-cluster cosine ~0.9), 1 % exact duplicates,
    queries near cluster centres -- rows iid over the clusters and rows stored cluster by cluster; next to the time: the rows per query that
    passed the filter threshold and reached the refine step, and the queries that took the exact fallback.  Exactness of the same
    workload: tests/test_gpu_search_clustered.py."""
    from lightretriever_amd import FlatIPIndex, _lib
    from lightretriever_amd.synth import clustered_corpus, cluster_queries
    lib = _lib.lib()
    out = {}
    N, D, Q, k, passes = 1_000_000, 2048, 100, 100, 20
    for order in ("shuffled", "by_cluster"):
        idx = FlatIPIndex(D, capacity=N, device=dev)
        info = clustered_corpus(idx.append_slot(N), n_clusters=1000, intra_cos=0.9, dup_frac=0.01, seed=5, order=order)
        idx.commit(N)
        q = cluster_queries(info["centres"], Q, query_cos=0.9, seed=6)
        idx.search(q, k)
        hits = idx.last_list_counts().float()
        torch.cuda.synchronize()
        lib.lrx_search_fallback_count(1)
        mean_ms, med_ms = time_search(lambda: idx.search(q, k), passes, warm=1)
        n_fb = int(lib.lrx_search_fallback_count(1))
        out[order] = {"workload": "exact top-%d of %d queries near cluster centres over %d x %d rows in 1000 clusters (intra-cluster cosine 0.9, "
                                  "1 %% exact duplicates), rows %s" % (k, Q, N, D, "iid over the clusters" if order == "shuffled" else "stored cluster by cluster"),
                      "ms": round(mean_ms, 4), "ms_median": round(med_ms, 4), "passes": passes, "roofline": hbm_roofline(N, D, Q, k, mean_ms),
                      "filter_hits_per_query": {"mean": round(hits.mean().item(), 1), "max": int(hits.max().item()), "list_capacity": 65536},
                      "fallback_queries_per_pass": round(n_fb / (passes + 1), 2)}
        del idx, info
        torch.cuda.empty_cache()
    return out


def embedding_bag_build_leg(args, dev):
    """Row N1 (finetune/nonctx_emb_utils.py:239-313): seconds to build the [vocab, H] query table of the headline model -- the shared-prefix
    build (prefix encoded once, lrx_encode_prefixed) over the whole vocabulary, and the reference's literal loop (every `[bos] prompt tok
    [eos]` sequence in full, lrx_encode_packed) timed on 20 000 rows and scaled."""
    from lightretriever_amd import EncoderConfig, LrxEncoder
    cfg = EncoderConfig.llama32_1b(args.seq_len)
    enc = LrxEncoder.random_init(cfg, seed=0, device=dev)
    V, H, P, bs = cfg.vocab_size, cfg.hidden_size, 21, 5000           # [bos] + a 20-token instruction
    pre = torch.randint(5, 1000, (P,), dtype=torch.int32, device=dev)
    L = P + 2

    def fast():
        table = torch.empty(V, H, dtype=torch.float32, device=dev)
        step = bs * max(1, L // 2)
        for s0 in range(0, V, step):
            e = min(s0 + step, V)
            suf = torch.empty(e - s0, 2, dtype=torch.int32, device=dev)
            suf[:, 0] = torch.arange(s0, e, dtype=torch.int32, device=dev)
            suf[:, 1] = 2
            enc.encode_prefixed(pre, suf, out=table[s0:e])
        return table

    def literal(rows):
        table = torch.empty(rows, H, dtype=torch.float32, device=dev)
        base = torch.empty(bs, L, dtype=torch.int32, device=dev)
        base[:, :P] = pre
        base[:, -1] = 2
        for s0 in range(0, rows, bs):
            e = min(s0 + bs, rows)
            base[:e - s0, -2] = torch.arange(s0, e, dtype=torch.int32, device=dev)
            cu = (torch.arange(e - s0 + 1, device=dev, dtype=torch.int64) * L).to(torch.int32)
            enc.encode_packed(base[:e - s0].reshape(-1), cu, L, out=table[s0:e], normalize=False)
        return table

    fast()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tf = fast()
    torch.cuda.synchronize()
    t_fast = time.perf_counter() - t0
    rows = 20000
    literal(bs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tl = literal(rows)
    torch.cuda.synchronize()
    t_lit = (time.perf_counter() - t0) * V / rows
    cos = torch.nn.functional.cosine_similarity(tf[:rows], tl, dim=-1).min().item()
    del enc, tf, tl
    torch.cuda.empty_cache()
    return {"workload": "EmbeddingBag table [%d, %d] of lightretriever-llama3.2-1b dims, %d-token shared prefix + [tok, eos]" % (V, H, P),
            "shared_prefix_build_s": round(t_fast, 3), "literal_full_sequence_build_s_scaled_from_%d_rows" % rows: round(t_lit, 3),
            "speedup": round(t_lit / t_fast, 2), "min_cosine_between_the_two_tables": round(cos, 6)}


def index_persistence_leg(dev):
    """Row N4 (retriever/faiss_search.py:99-123): save / load rate of a shard in the Faiss flat-index file layout (200 000 x 2048 fp32 =
    1.6 GB through /tmp), and that the reloaded shard returns the same bits."""
    import tempfile
    from lightretriever_amd import FlatIPIndex
    N, D = 200_000, 2048
    idx, g = synthetic_index(N, D, dev, 91)
    q = torch.nn.functional.normalize(torch.randn(100, D, generator=g, device=dev), dim=-1)
    D0, I0 = idx.search(q, 100)
    gb = N * D * 4 / 1e9
    with tempfile.TemporaryDirectory(dir=os.environ.get("TMPDIR", "/tmp")) as td:
        f = os.path.join(td, "shard.flat.faiss")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        idx.save(f)
        t_save = time.perf_counter() - t0
        t0 = time.perf_counter()
        idx2 = FlatIPIndex.load(f, device=dev)
        torch.cuda.synchronize()
        t_load = time.perf_counter() - t0
    D1, I1 = idx2.search(q, 100)
    same = bool(torch.equal(D0, D1) and torch.equal(I0, I1))
    del idx, idx2
    torch.cuda.empty_cache()
    return {"workload": "FlatIPIndex.save / load of %d x %d fp32 rows (%.2f GB, Faiss IndexFlatIP file layout) through a temporary directory; the "
                        "load includes rebuilding the fp16 shadow and the bounds on the GPU" % (N, D, gb),
            "save_gb_per_s": round(gb / t_save, 2), "load_gb_per_s": round(gb / t_load, 2), "save_s": round(t_save, 2), "load_s": round(t_load, 2),
            "hits_identical_after_reload": same}


def topk1000_legs(dev, headline_index):
    # ---- top_k = 1000 (the reference's default), Q = 100 and 1000, over the headline index and over one 100k-row chunk
    k1000 = {}
    if headline_index is not None and headline_index.d == 2048 and headline_index.ntotal == 1_000_000:
        for Qx, passes in ((100, 20), (1000, 5)):
            k1000["1Mx2048_Q%d" % Qx] = search_leg(1_000_000, 2048, Qx, 1000, dev, passes, index=headline_index)
        k100 = search_leg(1_000_000, 2048, 100, 100, dev, 20, index=headline_index)
        k1000["1Mx2048_Q100_k100_same_harness_ms"] = k100["ms"]
        k1000["k1000_over_k100_time_ratio_Q100"] = round(k1000["1Mx2048_Q100"]["ms"] / k100["ms"], 3)
    idx100k, _ = synthetic_index(100_000, 2048, dev, 31)
    for Qx, passes in ((100, 20), (1000, 5)):
        k1000["100kx2048_Q%d" % Qx] = search_leg(100_000, 2048, Qx, 1000, dev, passes, index=idx100k)
    del idx100k
    return k1000


def per_shard_leg(dev):
    # ---- the per-rank shard of the 8-GPU headline search: 125 000 x 2048, Q = 100, k = 100, + the exchange on a 1-rank RCCL group
    from lightretriever_amd.sharded import ShardedFlatIPIndex, _collective
    sh, g = synthetic_index(125_000, 2048, dev, 41)
    q = torch.nn.functional.normalize(torch.randn(100, 2048, generator=g, device=dev), dim=-1)
    local_ms, local_med = time_search(lambda: sh.search(q, 100), 40)
    pipe_l = lanes_ms(sh, q, 100, 80)
    rccl = "not initialised"
    try:
        if not dist.is_initialized():
            import socket
            with socket.socket() as so:
                so.bind(("127.0.0.1", 0))
                port = so.getsockname()[1]
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=dev)
        os.environ["LRX_FORCE_COLLECTIVE"] = "1"                                   # one-rank rehearsal: the exchange runs its collective
        shd = ShardedFlatIPIndex(sh)
        assert _collective(None, False)
        shd.search(q, 100)                                                         # communicator set-up outside the timing
        rccl = "1-rank RCCL all_gather_into_tensor"
        full_ms, full_med = time_search(lambda: shd.search(q, 100), 40)
        pipe_x = lanes_ms(shd, q, 100, 80)
    except Exception as e:  # noqa: BLE001  (the local time stands on its own; say why the exchange is missing)
        rccl = "unavailable: %r" % (e,)
        full_ms = full_med = pipe_x = None
    finally:
        os.environ.pop("LRX_FORCE_COLLECTIVE", None)
    # small query batches on the same shard (a serving loop that does not wait for 100 queries): the fused filter launch (sample + selection + main pass
    # in one persistent kernel, picked by the rule of lrx_search.hip:plan_chunk up to 16 queries) against the three-launch chain forced by flag
    from lightretriever_amd import _lib as _l
    small = {}
    for Qs in (1, 16, 32):
        qs_ = torch.nn.functional.normalize(torch.randn(Qs, 2048, generator=g, device=dev), dim=-1)
        sh.search_flags = _l.SEARCH_FILTER_AUTO
        f_ms, _ = time_search(lambda: sh.search(qs_, 100), 30)
        sh.search_flags = _l.SEARCH_FILTER_AUTO | _l.SEARCH_FUSED_NEVER
        c_ms, _ = time_search(lambda: sh.search(qs_, 100), 30)
        small["Q%d" % Qs] = {"fused_ms": round(f_ms, 4), "three_launch_chain_ms": round(c_ms, 4), "roofline_frac_fused": hbm_roofline(125_000, 2048, Qs, 100, f_ms)["frac"]}
    sh.search_flags = _l.SEARCH_FILTER_AUTO
    out = {
        "small_batches": small,
        "workload": "what ONE rank of an 8-GPU run of the headline search does: exact top-100 of 100 queries over its 125 000 x 2048 shard (the wire "
                    "words written by the search's own last kernel) -> all-gather of [Q,k] words -> lrx_merge_topk_packed", "local_search_ms": round(local_ms, 4),
        "local_search_ms_median": round(local_med, 4), "with_exchange_ms": None if full_ms is None else round(full_ms, 4),
        "with_exchange_ms_median": None if full_med is None else round(full_med, 4), "exchange": rccl,
        "two_in_flight": {"local_search_ms": round(pipe_l, 4), "with_exchange_ms": None if pipe_x is None else round(pipe_x, 4),
                          "note": "ms per search with two searches in flight on two HIP streams (pipeline.SearchLanes), 80 back to back",
                          "roofline_frac_local": hbm_roofline(125_000, 2048, 100, 100, pipe_l)["frac"]},
        "roofline": hbm_roofline(125_000, 2048, 100, 100, local_ms)}
    del sh
    torch.cuda.empty_cache()
    return out


def encode_8b_leg(args, dev):
    # ---- BASELINE configs[2]: Llama-3.1-8B dims, 128 documents x seq_len per step
    from lightretriever_amd import EncoderConfig, LrxEncoder
    cfg8 = EncoderConfig.llama31_8b(args.seq_len)
    if os.environ.get("LRX_BENCH_OPERANDS") in ("bf16", "fp16", "fp16_qkv"):  # dev switch (A/B of the fp32 stream's GEMM operands)
        cfg8.operand_dtype = os.environ["LRX_BENCH_OPERANDS"]
    enc8 = LrxEncoder.random_init(cfg8, seed=0, device=dev)
    ops8 = enc8.operand_mode
    B8, S = 128, args.seq_len
    g8 = torch.Generator(device=dev).manual_seed(77)
    ids8 = torch.randint(1000, 127000, (3, B8 * S), generator=g8, device=dev, dtype=torch.int64).to(torch.int32)
    cu8 = (torch.arange(B8 + 1, device=dev, dtype=torch.int64) * S).to(torch.int32)
    out8 = torch.empty(B8, cfg8.hidden_size, device=dev)
    enc8.encode_packed(ids8[0], cu8, S, out=out8)
    enc8.lib.lrx_set_profiling(1 << 3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in (1, 2):
        enc8.encode_packed(ids8[i], cu8, S, out=out8)
    torch.cuda.synchronize()
    t8 = time.perf_counter() - t0
    gu8 = enc8.get_profile()["gemm_swiglu"]
    enc8.set_profiling(False)
    tf8 = gu8["flops"] / (gu8["ms"] * 1e-3) / 1e12 if gu8["ms"] > 0 else 0.0
    out = {
        "workload": "lightretriever-llama3.1-8b dims, fp32 residual stream + %s GEMM operands, %d docs/step x seq_len %d, 2 timed steps after 1 warm-up "
                    "(BASELINE configs[2] encoder)" % (ops8, B8, S),
        "operands": ops8, "docs_per_s": round(2 * B8 / t8, 2), "ms_per_step": round(1e3 * t8 / 2, 2),
        "end_to_end_tflops": round(2 * B8 / t8 * cfg8.flops_per_doc(S) / 1e12, 1),
        "roofline": {"bound": "mfma", "kernel": "k_gemm_bf16_nt<EPI_SWIGLU> (M=%d N=%d K=%d)" % (B8 * S, 2 * cfg8.intermediate_size, cfg8.hidden_size),
                     "achieved": round(tf8, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(tf8 / PEAK_BF16_TFLOPS, 4),
                     "avg_launch_ms": round(gu8["ms"] / max(gu8["launches"], 1), 4), "launches": gu8["launches"], "traffic": None}}
    del enc8, out8, ids8
    torch.cuda.empty_cache()
    return out


def other_stream_leg(args, dev):
    """The headline model on the OTHER arithmetic: bf16 residual stream + norm weights folded into the projections (HF-bf16-like; the mode
    rounds 1-4 benchmarked).  It misses the 1e-3 bar on trained-like weights and is no backbone's default any more;
    this leg states what the default (fp32 stream + exact weights) costs in docs/s against it, on this box."""
    import dataclasses
    from lightretriever_amd import EncoderConfig, LrxEncoder
    B, S = 256, args.seq_len
    out = {}
    for name, precise, operands in (("bf16_stream_folded_norm", False, None), ("precise_fp32_stream_bf16_operands", True, "bf16"),
                                    ("precise_fp32_stream", True, "fp16_qkv"), ("precise_fp32_stream_fp16_operands", True, "fp16")):
        cfg = dataclasses.replace(EncoderConfig.llama32_1b(S), precise_stream=precise, operand_dtype=operands)
        enc = LrxEncoder.random_init(cfg, seed=0, device=dev)
        g = torch.Generator(device=dev).manual_seed(79)
        ids = torch.randint(1000, 127000, (4, B * S), generator=g, device=dev, dtype=torch.int64).to(torch.int32)
        cu = (torch.arange(B + 1, device=dev, dtype=torch.int64) * S).to(torch.int32)
        o = torch.empty(B, cfg.hidden_size, device=dev)
        enc.encode_packed(ids[0], cu, S, out=o)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in (1, 2, 3):
            enc.encode_packed(ids[i], cu, S, out=o)
        torch.cuda.synchronize()
        out[name] = {"docs_per_s": round(3 * B / (time.perf_counter() - t0), 1)}
        del enc, o, ids
        torch.cuda.empty_cache()
    out["workload"] = ("lightretriever-llama3.2-1b dims, 256 docs/step x seq_len %d, 3 timed steps after 1 warm-up: the bf16 stream; the fp32 stream with bf16 GEMM "
                       "operands, with the QKV projection's operands in fp16 (the default) and with every operand in fp16 (on request), on this box" % S)
    out["precise_over_bf16"] = round(out["precise_fp32_stream"]["docs_per_s"] / out["bf16_stream_folded_norm"]["docs_per_s"], 4)
    out["fp16_qkv_over_bf16_operands"] = round(out["precise_fp32_stream"]["docs_per_s"] / out["precise_fp32_stream_bf16_operands"]["docs_per_s"], 4)
    out["fp16_operands_over_bf16_operands"] = round(out["precise_fp32_stream_fp16_operands"]["docs_per_s"] / out["precise_fp32_stream_bf16_operands"]["docs_per_s"], 4)
    return out


def ragged_encode_leg(args, dev):
    """SURVEY 8d's ragged variant of the headline encoder THROUGH THE PRODUCT'S BATCHING: 8 192 documents with lengths
    clip(lognormal(5.3, 0.6), 16, S) sorted longest first (the reference sorts its corpus that way, retriever/hybrid_search.py:273-276), handed
    over as the collator hands them over -- packed batches of 256 documents -- to modeling._token_budget_batches (consecutive batches merged up
    to 131 072 tokens / 2 048 documents, the defaults of LrxExactSearchModel.encode) and LrxHybridModel.encode_passage, rows written in place.
    Token ids resident in HBM, cu_seqlens on the host like a collator's.  Next to it, on the same encoder in the same process: the
    fixed-length figure (256 x S per step), so that `tokens_per_s_over_fixed_length` is a same-box ratio."""
    import numpy as np
    from lightretriever_amd import EncoderConfig, LrxEncoder
    from lightretriever_amd.modeling import LrxHybridModel, _token_budget_batches
    S = args.seq_len
    cfg1 = EncoderConfig.llama32_1b(S)
    enc1 = LrxEncoder.random_init(cfg1, seed=0, device=dev)
    hm = LrxHybridModel(enc1, normalize=True)
    B1, n_docs, n_warm = 256, 8192, 1024
    max_tokens, max_docs = 131072, 2048
    rng = np.random.default_rng(4321)
    lens = np.sort(np.clip(rng.lognormal(5.3, 0.6, size=n_docs), 16, S).astype(np.int64))[::-1]
    g1 = torch.Generator(device=dev).manual_seed(78)

    def collated(lo, hi):
        for s0 in range(lo, hi, B1):
            l = lens[s0:min(s0 + B1, hi)]
            yield s0, s0 + len(l), {"input_ids": torch.randint(1000, 127000, (int(l.sum()),), generator=g1, device=dev, dtype=torch.int64).to(torch.int32),
                                    "cu_seqlens": torch.tensor(np.concatenate([[0], np.cumsum(l)]), dtype=torch.int32), "max_seqlen": int(l.max())}

    out1 = torch.empty(n_docs, cfg1.hidden_size, device=dev)
    warm = list(collated(0, n_warm))                                  # the longest documents: the workspace reaches its final size here
    timed = list(collated(0, n_docs))
    for s0, e0, b in _token_budget_batches(iter(warm), max_tokens, max_docs):
        hm.encode_passage(b, out=out1[s0:e0])
    torch.cuda.synchronize()
    n_calls, call_tokens = 0, []
    t0 = time.perf_counter()
    for s0, e0, b in _token_budget_batches(iter(timed), max_tokens, max_docs):
        hm.encode_passage(b, out=out1[s0:e0])
        n_calls += 1
        call_tokens.append(int(b["cu_seqlens"][-1]))
    torch.cuda.synchronize()
    t1 = time.perf_counter() - t0
    # fixed-length reference on the same encoder: 3 steps of 256 x S after one warm-up
    idsf = torch.randint(1000, 127000, (4, B1 * S), generator=g1, device=dev, dtype=torch.int64).to(torch.int32)
    cuf = (torch.arange(B1 + 1, device=dev, dtype=torch.int64) * S).to(torch.int32)
    enc1.encode_packed(idsf[0], cuf, S, out=out1[:B1])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in (1, 2, 3):
        enc1.encode_packed(idsf[i], cuf, S, out=out1[:B1])
    torch.cuda.synchronize()
    tf = time.perf_counter() - t0
    fixed_tps = 3 * B1 * S / tf
    out = {
        "workload": "lightretriever-llama3.2-1b dims bf16, %d documents, lengths clip(lognormal(5.3,0.6),16,%d) sorted longest first, collated in batches "
                    "of %d and merged by the product's token-budget batching (<= %d tokens, <= %d documents per encode call): %d encode calls "
                    "(SURVEY 8d ragged variant through LrxHybridModel.encode_passage; not the headline)" % (n_docs, S, B1, max_tokens, max_docs, n_calls),
        "docs_per_s": round(n_docs / t1, 1), "tokens_per_s": round(float(lens.sum()) / t1, 0), "mean_tokens_per_doc": round(float(lens.mean()), 1),
        "longest_doc": int(lens.max()), "shortest_doc": int(lens.min()), "encode_calls": n_calls,
        "tokens_per_call": {"min": min(call_tokens), "max": max(call_tokens)},
        "end_to_end_tflops": round(float(sum(cfg1.flops_per_doc(int(x)) for x in lens)) / t1 / 1e12, 1),
        "fixed_length_tokens_per_s_same_process": round(fixed_tps, 0), "tokens_per_s_over_fixed_length": round(float(lens.sum()) / t1 / fixed_tps, 4)}
    del enc1, hm, out1, warm, timed
    torch.cuda.empty_cache()
    return out


def coll_device(dev):
    """Where the bench's own tiny collectives (timing reductions, shard sizes) live: on the GPU over RCCL; on the host when the process group is
    gloo (LRX_BENCH_BACKEND=gloo: the rehearsal of N > 1 ranks on ONE GPU, tests/test_gpu_00_multi_gpu.py -- RCCL refuses two ranks per device)."""
    return torch.device("cpu") if (dist.is_initialized() and dist.get_backend() == "gloo") else dev


def reduce_max(x, dev, distributed):
    t = torch.tensor([x], device=coll_device(dev), dtype=torch.float64)
    if distributed:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_counts(n, dev, distributed):
    """[n of rank 0, n of rank 1, ...] as RCCL (or gloo) reports it"""
    if not distributed:
        return [int(n)]
    cd = coll_device(dev)
    t = torch.empty(dist.get_world_size(), dtype=torch.int64, device=cd)
    dist.all_gather_into_tensor(t, torch.tensor([int(n)], dtype=torch.int64, device=cd))
    return t.tolist()


def sharded_search_leg(name, index_rows, dim, nq, k, dev, rank, world, distributed, passes, seed):
    """One row-sharded search configuration OVER THE COMMUNICATOR (BASELINE configs[3] / configs[4]; reference: Faiss IndexShards over all GPUs,
    retriever/faiss_index.py:60-70): this rank holds shard_split(index_rows, rank, world) synthetic rows; a pass = local exact top-k (the
    search's last kernel writes the wire words) -> ONE all_gather_into_tensor of [Q,k] 64-bit words -> lrx_merge_topk_packed on every rank.
    Timed between barrier + synchronize, MAX over ranks; HIP events on the launch stream split a pass into local / exchange / merge.
    Also with two searches in flight (pipeline.SearchLanes over the communicator)."""
    from lightretriever_amd import FlatIPIndex
    from lightretriever_amd.sharded import ShardedFlatIPIndex
    rows, base = shard_split(index_rows, rank, world)
    need = rows * dim * 6 + (2 << 30)
    free = torch.cuda.mem_get_info(dev)[0] + torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev)
    fits = torch.tensor([1 if need < 0.85 * free else 0], device=coll_device(dev), dtype=torch.int64)
    if distributed:
        dist.all_reduce(fits, op=dist.ReduceOp.MIN)          # every rank takes the same branch (the leg is a sequence of collectives)
    if int(fits.item()) == 0:
        return {"skipped": "shard of %d x %d rows (%.1f GB with its fp16 shadow) does not fit next to what this rank holds (%.1f GB free)"
                           % (rows, dim, need / 1e9, free / 1e9)}
    idx = FlatIPIndex(dim, capacity=rows, device=dev, id_base=base)
    gi = torch.Generator(device=dev).manual_seed(seed + rank)
    slot = idx.append_slot(rows)
    step = max(1, (1 << 27) // dim)
    for s0 in range(0, rows, step):
        e = min(s0 + step, rows)
        slot[s0:e] = torch.nn.functional.normalize(torch.randn(e - s0, dim, generator=gi, device=dev), dim=-1)
    idx.commit(rows)
    sh = ShardedFlatIPIndex(idx)
    gq = torch.Generator(device=dev).manual_seed(seed + 1000)      # the same queries on every rank (replicated query side)
    q = torch.nn.functional.normalize(torch.randn(nq, dim, generator=gq, device=dev), dim=-1)
    for _ in range(2):
        Dm, Im = sh.search(q, k)
    marks = [{n_: torch.cuda.Event(enable_timing=True) for n_ in ("start", "local", "gathered", "merged")} for _ in range(passes)]
    barrier_sync(distributed)
    t0 = time.perf_counter()
    for m in marks:
        m["start"].record()
        Dl, Il, Wl = sh.local_search(q, k)
        m["local"].record()
        Dm, Im = sh.finish(Dl, Il, Wl, on_stage=lambda st, m=m: m[st].record())
        if Wl is None:                                        # (no exchange: plain one-process run)
            m["gathered"].record()
            m["merged"].record()
    barrier_sync(distributed)
    wall = reduce_max(time.perf_counter() - t0, dev, distributed)
    mean = lambda a, b: sum(m[a].elapsed_time(m[b]) for m in marks) / passes
    local_ms, exch_ms, merge_ms = mean("start", "local"), mean("local", "gathered"), mean("gathered", "merged")
    local_max = reduce_max(local_ms, dev, distributed)
    pipe = None
    try:
        from lightretriever_amd.pipeline import SearchLanes
        lanes = SearchLanes(sh, lanes=2)
        for _ in range(4):
            lanes.submit(q, k)
        lanes.drain()
        n_lp = max(passes, 100)
        barrier_sync(distributed)
        t0 = time.perf_counter()
        pend = [lanes.submit(q, k) for _ in range(n_lp)]
        lanes.drain()
        barrier_sync(distributed)
        pipe_s = reduce_max(time.perf_counter() - t0, dev, distributed)
        Dp_, Ip_ = pend[-1].result()
        pipe = {"lanes": 2, "passes": n_lp, "queries_per_s": round(nq * n_lp / pipe_s, 1), "ms_per_pass": round(1e3 * pipe_s / n_lp, 4),
                "identical_to_one_at_a_time": bool(torch.equal(Dp_, Dm) and torch.equal(Ip_, Im))}
    except Exception as e:  # noqa: BLE001
        pipe = {"failed": "%r" % (e,)}
    sizes = gather_counts(idx.ntotal, dev, distributed)
    out = {"workload": "%s: exact top-%d of %d queries over %d x %d fp32 rows (+ tiled fp16 shadow) row-sharded over %d rank(s); a pass = local "
                       "search -> all-gather of [Q,k] wire words -> on-device merge on every rank" % (name, k, nq, index_rows, dim, world),
           "queries_per_s": round(nq * passes / wall, 1), "ms_per_pass": round(1e3 * wall / passes, 4), "passes": passes,
           "local_search_ms": round(local_ms, 4), "local_search_ms_max_over_ranks": round(local_max, 4), "exchange_ms": round(exch_ms, 4),
           "merge_ms": round(merge_ms, 4), "two_in_flight": pipe, "rccl_ranks": dist.get_world_size() if distributed else 1,
           "shard_rows_per_rank": sizes, "index_rows": index_rows, "dim": dim, "scaling": "strong (fixed index row-sharded over ranks)",
           "exchange": ("all_gather_into_tensor over %s, %d bytes per rank" % (dist.get_backend(), nq * k * 8)) if distributed else "none (one process)",
           "roofline": hbm_roofline(rows, dim, nq, k, local_max)}
    out["roofline"]["note"] = "this rank's shard bytes / the slowest rank's local search time"
    del sh, idx, slot
    torch.cuda.empty_cache()
    return out


def sharded_encode_8b_leg(args, dev, rank, world, distributed):
    """BASELINE configs[2]/[3] encoder under weak scaling: every rank encodes its own 128 x seq_len batch of Llama-3.1-8B dims (replicated
    weights, no collective on the encode path); docs/s = all ranks' documents / the slowest rank's time."""
    from lightretriever_amd import EncoderConfig, LrxEncoder
    cfg8 = EncoderConfig.llama31_8b(args.seq_len)
    enc8 = LrxEncoder.random_init(cfg8, seed=0, device=dev)
    B8, S, n_t = 128, args.seq_len, max(1, min(args.steps, 3))
    g8 = torch.Generator(device=dev).manual_seed(77 + rank)
    ids8 = torch.randint(1000, 127000, (1 + n_t, B8 * S), generator=g8, device=dev, dtype=torch.int64).to(torch.int32)
    cu8 = (torch.arange(B8 + 1, device=dev, dtype=torch.int64) * S).to(torch.int32)
    out8 = torch.empty(B8, cfg8.hidden_size, device=dev)
    enc8.encode_packed(ids8[0], cu8, S, out=out8)
    enc8.lib.lrx_set_profiling(1 << 3)
    barrier_sync(distributed)
    t0 = time.perf_counter()
    for i in range(1, 1 + n_t):
        enc8.encode_packed(ids8[i], cu8, S, out=out8)
    barrier_sync(distributed)
    t8 = reduce_max(time.perf_counter() - t0, dev, distributed)
    gu8 = enc8.get_profile()["gemm_swiglu"]
    enc8.set_profiling(False)
    tf8 = gu8["flops"] / (gu8["ms"] * 1e-3) / 1e12 if gu8["ms"] > 0 else 0.0
    ops8 = enc8.operand_mode
    out = {"workload": "lightretriever-llama3.1-8b dims, fp32 residual stream + %s GEMM operands, %d docs/step/GPU x seq_len %d on %d rank(s), %d timed steps "
                       "after 1 warm-up, barrier + synchronize around them, MAX over ranks" % (ops8, B8, S, world, n_t),
           "operands": ops8, "docs_per_s": round(world * n_t * B8 / t8, 2), "ms_per_step": round(1e3 * t8 / n_t, 2), "n_gpus": world, "scaling": "weak",
           "end_to_end_tflops_per_gpu": round(n_t * B8 / t8 * cfg8.flops_per_doc(S) / 1e12, 1),
           "roofline": {"bound": "mfma", "kernel": "k_gemm_bf16_nt<EPI_SWIGLU> (M=%d N=%d K=%d), rank 0" % (B8 * S, 2 * cfg8.intermediate_size, cfg8.hidden_size),
                        "achieved": round(tf8, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(tf8 / PEAK_BF16_TFLOPS, 4),
                        "avg_launch_ms": round(gu8["ms"] / max(gu8["launches"], 1), 4), "launches": gu8["launches"], "traffic": None}}
    del enc8, out8, ids8
    torch.cuda.empty_cache()
    return out


def sharded_config_legs(args, dev, rank, world, distributed):
    """The BASELINE configurations that only exist over a communicator (configs[3]: 10M x 4096 rows row-sharded, configs[4]: 10M x 256 MRL rows,
    and the 8B encoder under weak scaling), run by EVERY rank; rank 0 keeps the dict.  With one forced RCCL rank (LRX_BENCH_FORCE_DIST=1) the
    same code runs on one GPU (tests pass a smaller --sharded-rows)."""
    legs = {}
    n_pass = max(20, 2 * args.steps)
    legs["config3_10Mx4096"] = sharded_search_leg("BASELINE configs[3] index (lightretriever-llama3.1-8b width)", args.sharded_rows, 4096,
                                                  args.queries, args.topk, dev, rank, world, distributed, n_pass, 301)
    legs["config4_10Mx256_mrl"] = sharded_search_leg("BASELINE configs[4] index (MRL dim 256)", args.sharded_rows, 256, args.queries, args.topk,
                                                     dev, rank, world, distributed, n_pass, 401)
    legs["config3_encode_llama31_8b"] = sharded_encode_8b_leg(args, dev, rank, world, distributed)
    return legs


def shard_split(index_rows, rank, world):
    """(rows, first global row) of rank `rank` of a row-sharded index of `index_rows` rows: contiguous ranges, index_rows // world rows
    each, the remainder one row each to the first ranks -- every row has exactly one owner for ANY world size."""
    base, rem = divmod(index_rows, world)
    return base + (1 if rank < rem else 0), rank * base + min(rank, rem)


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N ranks with torch.distributed.run as a CHILD process (before anything in
    this process touches the GPU) and pass its exit code on.  Under a launcher (WORLD_SIZE set) the world size must equal --gpus."""
    import subprocess
    n_dev = torch.cuda.device_count()                     # (does not initialise the GPU on this image)
    if n_dev < args.gpus:
        sys.stderr.write("bench.py: --gpus %d but only %d GPU(s) visible\n" % (args.gpus, n_dev))
        sys.exit(2)
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    sys.exit(subprocess.call(cmd, env=env))


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        launch_ranks(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.stderr.write("bench.py: launched with WORLD_SIZE=%d but --gpus %d\n" % (world, args.gpus))
        sys.exit(2)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # LRX_BENCH_FORCE_DIST=1: create the RCCL process group (and run every collective of the N>1 path) even with one rank --
    # the single-GPU rehearsal of the `torch.distributed.run` launch the driver uses for N = 2, 4, 8
    distributed = world > 1 or os.environ.get("LRX_BENCH_FORCE_DIST") == "1"
    # LRX_BENCH_BACKEND=gloo + LRX_BENCH_ONE_GPU=1: the rehearsal of an N-rank run on a box with ONE GPU (every rank on device 0, collectives over
    # gloo with the wire words staged through the host: sharded.py) -- every branch of the N > 1 path runs; the numbers mean nothing
    backend = os.environ.get("LRX_BENCH_BACKEND", "nccl")
    if os.environ.get("LRX_BENCH_ONE_GPU") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if distributed:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "gloo":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        if world == 1:
            os.environ["LRX_FORCE_COLLECTIVE"] = "1"      # the one-rank rehearsal runs the exchange's all-gather + merge too (sharded.py)
    dev = torch.device("cuda", local_rank)

    from lightretriever_amd import EncoderConfig, FlatIPIndex, LrxEncoder
    from lightretriever_amd.sharded import ShardedFlatIPIndex

    cfg = {"llama3.2-1b": EncoderConfig.llama32_1b, "llama3.2-3b": EncoderConfig.llama32_3b, "llama3.1-8b": EncoderConfig.llama31_8b,
           "qwen2.5-1.5b": EncoderConfig.qwen25_1_5b, "qwen2.5-3b": EncoderConfig.qwen25_3b, "qwen2.5-7b": EncoderConfig.qwen25_7b}[args.model](args.seq_len)
    if os.environ.get("LRX_FOLD_NORM") is not None:          # dev A/B switch; the default is the library's (folded)
        cfg.fold_norm = os.environ["LRX_FOLD_NORM"] != "0"
    all_legs = ("encode", "search", "sparse", "sharded", "configs", "cpu")
    legs = set(all_legs) if args.legs == "all" else set(args.legs.split(","))
    if legs - set(all_legs):
        sys.stderr.write("bench.py: unknown --legs %s\n" % sorted(legs - set(all_legs)))
        sys.exit(2)
    partial = legs != set(all_legs)
    if args.no_search:
        legs.discard("search")
    if args.no_sparse:
        legs.discard("sparse")
    if args.no_configs:
        legs.discard("configs")
    if args.no_cpu_baseline:
        legs.discard("cpu")
    need_enc = bool(legs & {"encode", "sparse"})
    B, S, H = args.batch_docs, args.seq_len, cfg.hidden_size
    # the arithmetic mode the timed steps run: the library default (fp32 residual stream + exact weights since round 5: the mode that holds
    # 1e-3 cosine against the HF fp32 model on trained-like weights, tests/test_gpu_trained_like.py); LRX_BENCH_BF16_STREAM=1 is a dev switch
    if os.environ.get("LRX_BENCH_BF16_STREAM") == "1":
        cfg.precise_stream = False
    if os.environ.get("LRX_BENCH_OPERANDS") in ("bf16", "fp16", "fp16_qkv") and cfg.use_precise_stream():     # dev switch (A/B of the fp32 stream's GEMM operands)
        cfg.operand_dtype = os.environ["LRX_BENCH_OPERANDS"]
    stream_mode = ("fp32-stream(precise, %s operands)" % {"fp16": "fp16", "fp16_qkv": "bf16 + fp16-QKV", "bf16": "bf16"}[cfg.operand_mode()]) if cfg.use_precise_stream() else "bf16-stream(folded-norm)"
    enc = LrxEncoder.random_init(cfg, seed=0, device=dev) if need_enc else None
    D = args.mrl_dim or H                       # embedding / index width (MRL slice of the pooled state when < H)
    from lightretriever_amd import _lib
    lrx = _lib.lib()

    # ---- synthetic inputs, resident in HBM: ids uniform in [1000,127000), bos first / eos last
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    n_batches = args.warmup + args.steps
    ids_all = torch.randint(1000, 127000, (n_batches, B, S), generator=g, device=dev, dtype=torch.int64).to(torch.int32)
    ids_all[:, :, 0] = 128000
    ids_all[:, :, -1] = 128001
    cu = (torch.arange(B + 1, device=dev, dtype=torch.int64) * S).to(torch.int32)
    batches = None
    if args.ragged:
        import numpy as np
        rng = np.random.default_rng(1234 + rank)
        lens = np.clip(rng.lognormal(5.3, 0.6, size=n_batches * B), 16, S).astype(np.int64)
        lens = np.sort(lens)[::-1]                                  # longest first, like the reference's corpus sort
        batches = []
        for i in range(n_batches):
            l = lens[i * B:(i + 1) * B]
            cu_i = torch.tensor(np.concatenate([[0], np.cumsum(l)]), dtype=torch.int32, device=dev)
            ids_i = torch.randint(1000, 127000, (int(l.sum()),), generator=g, device=dev, dtype=torch.int64).to(torch.int32)
            batches.append((ids_i, cu_i, int(l.max()), int(l.sum()), float(sum(cfg.flops_per_doc(int(x)) for x in l)) / B))

    # ---- index shard: rows/world rows of L2-normalised N(0,1) fp32 (seed 7); encoded batches overwrite its first rows
    # (row r of the index belongs to the rank whose [id_base, id_base + shard_rows) holds it; a remainder of index_rows % world rows goes
    # one each to the first ranks: every row of the index is searched whatever the world size)
    shard_rows, id_base = shard_split(args.index_rows, rank, world)
    index = FlatIPIndex(D, capacity=max(shard_rows, n_batches * B), device=dev, id_base=id_base)
    gi = torch.Generator(device=dev).manual_seed(7 + rank)
    slot = index.append_slot(shard_rows)
    for s in range(0, shard_rows, 65536):
        e = min(s + 65536, shard_rows)
        slot[s:e] = torch.nn.functional.normalize(torch.randn(e - s, D, generator=gi, device=dev), dim=-1)
    index.commit(shard_rows)
    sharded = ShardedFlatIPIndex(index)
    # query-side synthetic inputs (8-32 random token ids per query -> EmbeddingBag(mean) over a synthetic [V, D] table), generated NOW: no
    # torch kernel runs between the first encode launch and the last search launch
    qsets = {}
    if "search" in legs:
        gq = torch.Generator(device=dev).manual_seed(99)      # same queries on every rank (replicated query side)
        table = torch.randn(cfg.vocab_size, D, generator=gq, device=dev)
        for Qx in (args.queries, 1, 1000):
            lens_x = torch.randint(8, 33, (Qx,), generator=gq, device=dev)
            offs_x = (torch.cumsum(lens_x, 0) - lens_x).to(torch.int64)
            ids_x = torch.randint(1000, 127000, (int(lens_x.sum().item()),), generator=gq, device=dev)
            qsets.setdefault(Qx, (ids_x, offs_x))
    # what RCCL itself reports: one all-gather of every rank's shard size (also the first collective: communicator set-up stays out
    # of the timed regions)
    rccl_ranks, shard_rows_all = (1, [shard_rows])
    if distributed:
        rccl_ranks, shard_rows_all = dist.get_world_size(), gather_counts(index.ntotal, dev, distributed)

    def encode_step(i):
        out = index._x[i * B:(i + 1) * B]           # in place into the shard (no host round trip)
        if batches is None:
            enc.encode_packed(ids_all[i].reshape(-1), cu, S, out=out, out_dim=D)
        else:
            enc.encode_packed(batches[i][0], batches[i][1], batches[i][2], out=out, out_dim=D)

    # ---- encode leg: W warmup steps, then EXACTLY K timed steps between barrier + synchronize.  Inside the timed region only the
    #      dominant kernel (gate-up GEMM, class 2) is bracketed by HIP events on the launch stream -- read after the region, no
    #      host sync inside it; the per-class table comes from two extra fully profiled steps afterwards.
    # k_trace_marker<0> ... <1> bracket the HEADLINE legs in a rocprofv3 kernel trace (tools/check_trace_clean.py counts every kernel between
    # them that is not liblrx's)
    _lib.check(lrx.lrx_trace_marker(0, _lib.current_stream()))
    enc_s, docs_per_s, gu_timed, prof, n_prof = None, None, None, {}, 0
    if "encode" in legs:
        for i in range(args.warmup):
            encode_step(i)
        enc.lib.lrx_set_profiling(1 << 3)
        barrier_sync(distributed)
        t0 = time.perf_counter()
        for i in range(args.steps):
            encode_step(args.warmup + i)
        barrier_sync(distributed)
        enc_s = time.perf_counter() - t0
        gu_timed = enc.get_profile()["gemm_swiglu"]
        enc_s = reduce_max(enc_s, dev, distributed)
        docs_per_s = world * B * args.steps / enc_s
        enc.set_profiling(True)
        n_prof = min(2, args.steps)
        for i in range(n_prof):
            encode_step(args.warmup + i)
        prof = enc.get_profile()
        enc.set_profiling(False)

    # ---- search leg (queries: 8-32 random token ids -> EmbeddingBag(mean) over a synthetic [V,H] table -> normalise)
    search = None
    if "search" in legs:
        from lightretriever_amd import ops
        q_ids, offs = qsets[args.queries]
        # a search pass is ~1 ms against ~180 ms for an encode step: K passes would be dominated by the fixed cost of the two barriers
        # around them (20 passes: +6 % per pass), so the search leg times 5 K passes and says so (`passes`)
        n_pass = 5 * args.steps
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2 * n_pass)]
        q = ops.embedding_bag_mean(table, q_ids, offs, normalize=True)
        for _ in range(max(1, args.warmup)):
            sharded.search(q, args.topk)
        barrier_sync(distributed)
        t0 = time.perf_counter()
        for i in range(n_pass):
            q = ops.embedding_bag_mean(table, q_ids, offs, normalize=True)
            ev[2 * i].record()
            Dk, Ik, Wk = sharded.local_search(q, args.topk)        # (with more than one rank the search's last kernel also writes the wire words)
            ev[2 * i + 1].record()
            Dk, Ik = sharded.finish(Dk, Ik, Wk)
        _lib.check(lrx.lrx_trace_marker(1, _lib.current_stream()))
        barrier_sync(distributed)
        srch_s = time.perf_counter() - t0
        srch_s = reduce_max(srch_s, dev, distributed)
        local_ms = sum(ev[2 * i].elapsed_time(ev[2 * i + 1]) for i in range(n_pass)) / n_pass
        # the filter pass streams the fp16 shadow of the shard (2 B/element); the exact rescoring of the few hundred band rows per
        # query comes on top (measured: `traffic`) -- the fp32 rows themselves are never streamed
        shadow = index._xb is not None and index.two_pass
        alg_bytes = shard_rows * D * (2 if shadow else 4) + args.queries * D * 4 + args.queries * args.topk * 12
        # ---- the same passes with TWO searches in flight (pipeline.SearchLanes: two HIP streams, own workspaces; the query producer runs on
        #      the lane too): the short latency-bound kernels that frame one search's streaming passes overlap the other's passes.  Same work
        #      per pass, same bits; between the same barrier + synchronize brackets: the throughput a serving loop with the next batch at hand
        #      gets.  Reported next to `value`, not as it.
        pipe = None
        try:
            # (more than one rank: every rank submits the same searches in the same order, so the lanes' all-gathers reach the communicator in the
            # same order everywhere; tests/test_gpu_00_multi_gpu.py runs exactly this over R = 2, 4, 8 RCCL ranks where the GPUs exist)
            from lightretriever_amd.pipeline import SearchLanes
            lanes = SearchLanes(sharded, lanes=2)
            mk = lambda: ops.embedding_bag_mean(table, q_ids, offs, normalize=True)
            for _ in range(4):
                lanes.submit(mk, args.topk)
            lanes.drain()
            # (at least 100 passes: the fixed cost of the two barriers and of filling / draining the lanes is ~1 ms, +7 % on 20 passes of 0.7 ms)
            n_lp = max(n_pass, 100)
            barrier_sync(distributed)
            t0 = time.perf_counter()
            pend = [lanes.submit(mk, args.topk) for _ in range(n_lp)]
            lanes.drain()
            barrier_sync(distributed)
            pipe_s = time.perf_counter() - t0
            Dp_, Ip_ = pend[-1].result()
            pipe_s = reduce_max(pipe_s, dev, distributed)
            pipe = {"lanes": 2, "passes": n_lp, "queries_per_s": round(args.queries * n_lp / pipe_s, 2), "ms_per_pass": round(1e3 * pipe_s / n_lp, 4),
                    "identical_to_one_at_a_time": bool(torch.equal(Dp_, Dk) and torch.equal(Ip_, Ik))}
        except Exception as e:  # noqa: BLE001  (the one-at-a-time figure then stands as the value)
            pipe = {"failed": "%r" % (e,)}
        # (`value` stays the one-search-at-a-time figure of rounds 1-3; at 1M rows the chain is 90 % streaming passes and a second search in
        # flight buys 0-4 %, on the 125 k-row per-rank shard 14 % -- `configs.search_per_shard_8way.two_in_flight`)
        best_s = srch_s
        search = {
            "metric": "queries/sec @ top-%d over %d-doc fp32 index" % (args.topk, args.index_rows),
            "value": round(args.queries * n_pass / best_s, 2),
            "unit": "queries/s", "ms_per_pass": round(1e3 * best_s / n_pass, 4), "passes": n_pass, "queries": args.queries, "index_rows": args.index_rows,
            "two_in_flight": pipe,
            "dim": D, "shard_rows": shard_rows, "shard_rows_per_rank": shard_rows_all, "rccl_ranks": rccl_ranks,
            "scaling": "strong (fixed index row-sharded over ranks)",
            "roofline": {"bound": "hbm", "achieved": round(alg_bytes / (local_ms * 1e-3) / 1e9, 2), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                         "frac": round(alg_bytes / (local_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),

                         "traffic": pmc_search_traffic() if (args.index_rows == 1_000_000 and world == 1 and args.queries == 100 and D == 2048) else None,
                         "algorithmic_bytes": alg_bytes, "traffic_source": {"file": PMC_SUMMARY, "git_blob": git_blob_sha(PMC_SUMMARY), "note": "offline rocprofv3 --pmc passes, not measured by this run"},
                         "kernel": "k_filter_xreg_emit (+ sample, threshold, refine: DESIGN.md 5.4)", "corpus_bytes_per_element_streamed": 2 if shadow else 4, "ms": round(local_ms, 4),
                         "fp32_equiv_tflops": round(2.0 * args.queries * D * shard_rows / (local_ms * 1e-3) / 1e12, 2),
                         "fp32_mfma_peak_for_reference": PEAK_F32_MFMA_TFLOPS},
        }

        # the same local search captured in a HIP graph and replayed (the library neither allocates nor synchronises, so a serving loop can
        # replay it; tests/test_gpu_search_emit.py::test_search_captured_in_a_hip_graph_replays_bit_identically): launch gaps of the chain gone
        try:
            qg = q.clone()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                index.search(qg, args.topk)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                Dgr, Igr = index.search(qg, args.topk)
            def back_to_back(fn, n):                 # one event pair around n calls queued without a gap: what a serving loop sees
                for _ in range(3):
                    fn()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(n):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                return e0.elapsed_time(e1) / n
            g_ms = back_to_back(graph.replay, 2 * n_pass)
            e_ms = back_to_back(lambda: index.search(qg, args.topk), 2 * n_pass)
            De_, Ie_ = index.search(qg, args.topk)
            search["graph_replay"] = {"ms_back_to_back": round(g_ms, 4), "eager_ms_back_to_back": round(e_ms, 4), "calls": 2 * n_pass,
                                      "bit_identical_to_eager": bool(torch.equal(Dgr, De_) and torch.equal(Igr, Ie_)),
                                      "roofline_frac": round(alg_bytes / (g_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)}
            del graph
        except Exception as e:  # noqa: BLE001
            search["graph_replay"] = {"failed": "%r" % (e,)}

        # the other query counts of SURVEY 8d (Q = 1 and Q = 1000), same flow (EmbeddingBag -> local search -> exchange + merge)
        other = {}
        for Qx in (1, 1000):
            ids_x, offs_x = qsets[Qx]
            sharded.search(ops.embedding_bag_mean(table, ids_x, offs_x, normalize=True), args.topk)
            barrier_sync(distributed)
            t0 = time.perf_counter()
            n_px = n_pass if Qx == 1 else args.steps
            for _ in range(n_px):
                sharded.search(ops.embedding_bag_mean(table, ids_x, offs_x, normalize=True), args.topk)
            barrier_sync(distributed)
            sx = reduce_max(time.perf_counter() - t0, dev, distributed)
            ms_x = 1e3 * sx / n_px
            # which bound applies: one query streams the shard's fp16 shadow once (HBM); 1000 queries are one pass of the f16 MFMA
            # GEMM over it (wide chunks: the shadow is read once per <= 1024 queries) -- both fractions, the applicable one named
            hb = hbm_roofline(shard_rows, D, Qx, args.topk, ms_x)
            tf = 2.0 * Qx * D * shard_rows / (ms_x * 1e-3) / 1e12
            other[str(Qx)] = {"queries_per_s": round(Qx * n_px / sx, 1), "ms_per_pass": round(ms_x, 4), "passes": n_px,
                              "roofline": {"bound": "hbm" if Qx <= 128 else "mfma", "frac": hb["frac"] if Qx <= 128 else round(tf / PEAK_BF16_TFLOPS, 4),
                                           "hbm_frac": hb["frac"], "hbm_achieved_gbs": hb["achieved"], "mfma_f16_tflops": round(tf, 1),
                                           "mfma_frac": round(tf / PEAK_BF16_TFLOPS, 4), "mfma_peak_tflops": PEAK_BF16_TFLOPS,
                                           "note": "whole pass incl. EmbeddingBag + exchange; f16 dense MFMA peak = the bf16 figure"}}
        search["other_query_counts"] = other

    # ---- dense + sparse document vectors (row N2): same batches through lrx_encode_packed_sparse; not part of `value`
    if "search" not in legs:
        _lib.check(lrx.lrx_trace_marker(1, _lib.current_stream()))
    sparse = None
    if "sparse" in legs and batches is None:
        n_sp = min(2, args.steps)
        enc.encode_packed_sparse(ids_all[0].reshape(-1), cu, S)
        enc.lib.lrx_set_profiling(1 << 8)                    # events around the max-aggregation GEMM (class 7) only
        barrier_sync(distributed)
        t0 = time.perf_counter()
        for i in range(n_sp):
            enc.encode_packed_sparse(ids_all[args.warmup + i].reshape(-1), cu, S)
        barrier_sync(distributed)
        sp_s = time.perf_counter() - t0
        mx_ms = enc.get_profile()["gemm_maxagg"]["ms"]
        enc.set_profiling(False)
        sp_s = reduce_max(sp_s, dev, distributed)
        mx_fl = 2.0 * B * S * cfg.vocab_size * H
        sparse = {"metric": "docs/sec with dense + sparse (LM-head max aggregation, relu, log1p) vectors", "value": round(world * B * n_sp / sp_s, 2),
                  "unit": "docs/s", "steps": n_sp, "ms_per_step": round(1e3 * sp_s / n_sp, 3),
                  "roofline": {"bound": "mfma", "kernel": "k_gemm_bf16_nt<EPI_MAXAGG> (M=%d N=%d K=%d, segmented column max in the epilogue)" % (B * S, cfg.vocab_size, H),
                               "achieved": round(mx_fl / (mx_ms / n_sp * 1e-3) / 1e12, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                               "frac": round(mx_fl / (mx_ms / n_sp * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4), "avg_launch_ms": round(mx_ms / n_sp, 3),
                               "traffic": pmc_traffic("k_gemm_bf16_nt<4>") if (args.model == "llama3.2-1b" and B == 256 and S == 512) else None}}

    # ---- BASELINE configs[3] / configs[4] + the 8B encoder over the communicator (every rank takes part; N > 1, or one forced RCCL rank)
    sharded_legs = None
    if distributed and "sharded" in legs and batches is None:
        if world > 1:                              # (N = 1 keeps the headline index for the single-GPU `configs` legs below)
            del sharded, index, slot
            enc = None
            torch.cuda.empty_cache()
        try:
            sharded_legs = sharded_config_legs(args, dev, rank, world, distributed)
        except Exception as e:  # noqa: BLE001  (never take the headline line down; a rank that fails here fails the collectives of all)
            sharded_legs = {"failed": "%r" % (e,)}

    if rank != 0:
        if distributed:
            dist.destroy_process_group()
        return

    if "encode" not in legs:
        line = {"partial_run": sorted(legs), "metric": "partial run (profiling aid): not the headline line", "value": None, "unit": "docs/s", "n_gpus": world,
                "search": search, "sparse": sparse}
        if world == 1 and "configs" in legs and batches is None:
            line["configs"] = extra_legs(args, dev, index if (D == 2048 and args.index_rows == 1_000_000) else None)
        if sharded_legs is not None:
            line.setdefault("configs", {}).update(sharded_legs)
        print(json.dumps(line), flush=True)
        if dist.is_initialized():
            dist.destroy_process_group()
        return

    # ---- roofline of the dominant kernel: the gate-up SwiGLU GEMM (55% of the model FLOPs)
    gu = gu_timed                      # gate-up launches of the timed region
    timed = range(args.warmup, args.warmup + args.steps)
    # FLOPs per document of the timed batches: F(S) for the fixed-length headline, the mean of F(len) over the ragged batches
    fl_doc = cfg.flops_per_doc(S) if batches is None else sum(batches[i][4] for i in timed) / args.steps
    m_rows = "%d" % (B * S) if batches is None else "%d..%d" % (min(batches[i][3] for i in timed), max(batches[i][3] for i in timed))
    gemm_all_ms = sum(prof[k_]["ms"] for k_ in ("gemm_store", "gemm_resid", "gemm_swiglu"))
    gemm_all_fl = sum(prof[k_]["flops"] for k_ in ("gemm_store", "gemm_resid", "gemm_swiglu"))
    achieved = gu["flops"] / (gu["ms"] * 1e-3) / 1e12 if gu["ms"] > 0 else 0.0
    roofline = {
        "bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / PEAK_BF16_TFLOPS, 4),
        "traffic": pmc_traffic("k_gemm_bf16_nt<2>") if (args.model == "llama3.2-1b" and B == 256 and S == 512 and batches is None) else None,
        "kernel": "k_gemm_bf16_nt<EPI_SWIGLU> M=%s N=%d K=%d" % (m_rows, 2 * cfg.intermediate_size, H),
        "avg_launch_ms": round(gu["ms"] / max(gu["launches"], 1), 4), "launches": gu["launches"],
        "all_gemms_tflops": round(gemm_all_fl / (gemm_all_ms * 1e-3) / 1e12, 2) if gemm_all_ms > 0 else None,
        "per_class_ms_per_step": {k_: round(v["ms"] / n_prof, 3) for k_, v in prof.items()},
        # NOT measured by this run (PMC counters cannot be read from inside the timed process): numbers of committed rocprofv3 --pmc passes,
        # each with the git blob id of the file it was read from; `traffic` above comes from pmc_summary
        "offline_profile": {"pmc_summary": {"file": PMC_SUMMARY, "git_blob": git_blob_sha(PMC_SUMMARY)},
                            "pmc_mfma": {"file": PMC_MFMA, "git_blob": git_blob_sha(PMC_MFMA), "mfma_busy_frac": pmc_mfma("k_gemm_bf16_nt<2>")[0],
                                         "effective_clock_ghz": pmc_mfma("k_gemm_bf16_nt<2>")[1]},
                            "gemm_power": {"file": "profiles/r04_gemm_power.txt", "git_blob": git_blob_sha("profiles/r04_gemm_power.txt"),
                                           "note": "tools/power_probe.sh on the round-4 gate-up kernel: rocm-smi package power / shader clock while one launch loops",
                                           "random_operands": {"tflops": 1432, "shader_clock_ghz": 1.85, "package_power_w": "1373-1377 of 1400 (cap)"},
                                           "constant_or_zero_operands": {"tflops": "1925-1936", "shader_clock_ghz": 2.40, "package_power_w": "1086-1242"},
                                           "vendor_gemm_same_shape_random_operands": {"tflops": 1449, "shader_clock_ghz": 1.85, "package_power_w": 1390}}}
                           if args.model == "llama3.2-1b" else None,
        "model_flops_per_doc": fl_doc,
        "end_to_end_tflops": round(docs_per_s / world * fl_doc / 1e12, 2),
    }
    # (end_to_end_tflops counts the algorithmic F(S) per document; the final layer's O-projection / MLP run on the B pooled rows only, so the
    # FLOPs actually executed are lower: `executed_tflops` = the sum over the profiled kernel classes / the profiled steps' time share)
    ex_fl = sum(v["flops"] for v in prof.values()) / max(n_prof, 1)
    roofline["executed_tflops"] = round(ex_fl / (enc_s / args.steps) / 1e12, 2)
    line = {
        "metric": "docs embedded/sec (%s dims, seq_len=%d, bf16) [+ queries/sec@top-%d over %d-doc index in `search`]" % (args.model, S, args.topk, args.index_rows),
        "value": round(docs_per_s, 2), "unit": "docs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * enc_s / args.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "fp16" if cfg.use_f16_operands() else "bf16", "data": "synthetic (random-init weights, uniform random token ids, N(0,1) normalised index rows)",
        "rccl_ranks": rccl_ranks,
        "config": {"workload": ("configs[1] %s bf16 %s, %d docs x %d tok/step/GPU, top-%d over %dx%d fp32 index"
                                % (args.model, stream_mode, B, S, args.topk, args.index_rows, D)) if batches is None else
                               ("RAGGED variant (not the headline): lightretriever-%s bf16, %d docs/step, lengths clip(lognormal(5.3,0.6),16,%d) sorted "
                                "longest first, mean %.0f tokens/doc" % (args.model, B, S, sum(b_[3] for b_ in batches[args.warmup:]) / (args.steps * B))),
                   "global_batch": world * B, "seq_len": S, "parallelism": "dp%d" % world, "stream_mode": stream_mode},
        "roofline": roofline,
        "search": search,
        "sparse": sparse,
    }
    # the queries/sec half of BASELINE.json's metric as plain scalars -- at the top level, inside `config` / `roofline` (the driver's parsed
    # record keeps the scalars of those objects) and once more in `headline`, the LAST key of the line (the driver's stdout tail ends with it)
    if search is not None:
        sk = {"search_qps": search["value"], "search_ms_per_pass": search["ms_per_pass"], "search_local_ms": search["roofline"]["ms"],
              "search_roofline_frac": search["roofline"]["frac"], "search_achieved_gbs": search["roofline"]["achieved"],
              "search_alg_bytes": search["roofline"]["algorithmic_bytes"], "search_rccl_ranks": rccl_ranks, "search_shard_rows": shard_rows}
        line.update(sk)
        line["config"].update({"search_qps": sk["search_qps"], "search_ms_per_pass": sk["search_ms_per_pass"], "search_queries": args.queries,
                               "search_topk": args.topk, "search_index_rows": args.index_rows, "search_dim": D})
        roofline.update({"search_bound": "hbm", "search_frac": sk["search_roofline_frac"], "search_achieved_gbs": sk["search_achieved_gbs"],
                         "search_peak_gbs": PEAK_HBM_GBS, "search_alg_bytes": sk["search_alg_bytes"], "search_local_ms": sk["search_local_ms"]})
    if partial:
        line["partial_run"] = sorted(legs)
    if world == 1 and "configs" in legs and batches is None:
        try:
            line["configs"] = extra_legs(args, dev, index if (D == 2048 and args.index_rows == 1_000_000) else None)
            mc = line["configs"].get("measured_ceilings")
            if mc:                                         # the ceilings measured in this run, next to the spec peaks of `roofline`
                roofline["measured_ceilings"] = {"vendor_gemm_same_shape_tflops": mc["vendor_gemm_bf16"]["shapes"]["gate_up"]["tflops"],
                                                 "frac_of_vendor_gemm": round(achieved / mc["vendor_gemm_bf16"]["shapes"]["gate_up"]["tflops"], 4),
                                                 "hbm_stream_read_gbs": mc["hbm_stream_read"]["value"]}
                if search is not None:
                    search["roofline"]["measured_ceiling_gbs"] = mc["hbm_stream_read"]["value"]
                    search["roofline"]["frac_of_measured_ceiling"] = round(search["roofline"]["achieved"] / mc["hbm_stream_read"]["value"], 4)
        except Exception as e:  # noqa: BLE001  (additional legs: never take the headline line down with them)
            line["configs"] = {"failed": "%r" % (e,)}
    if sharded_legs is not None:
        line.setdefault("configs", {}).update(sharded_legs)
    if world == 1 and "cpu" in legs:
        try:
            cb = cpu_baseline(cfg, S, args.topk, D)
            if cb.get("search"):
                cb["search"]["scaled_to_index_rows"] = round(cb["search"]["value"] * 50_000 / args.index_rows, 3)
            line["cpu_baseline"] = cb
        except Exception as e:  # noqa: BLE001  (the baseline is a reported number, never the product path)
            line["cpu_baseline"] = {"value": None, "unit": "docs/s", "cores": usable_cores(), "kind": "reference", "sample": "failed: %r" % (e,)}
    line["headline"] = {"docs_per_s": line["value"], "ms_per_step": line["ms_per_step"], "stream_mode": stream_mode, "n_gpus": world,
                        "encode_roofline_frac": roofline["frac"], "encode_achieved_tflops": roofline["achieved"],
                        "search_qps": line.get("search_qps"), "search_ms_per_pass": line.get("search_ms_per_pass"),
                        "search_local_ms": line.get("search_local_ms"), "search_roofline_frac": line.get("search_roofline_frac"),
                        "search_alg_bytes": line.get("search_alg_bytes"), "rccl_ranks": rccl_ranks,
                        "cpu_baseline_docs_per_s": (line.get("cpu_baseline") or {}).get("value"),
                        "cpu_baseline_search_qps_scaled": ((line.get("cpu_baseline") or {}).get("search") or {}).get("scaled_to_index_rows")}
    print(json.dumps(line), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
