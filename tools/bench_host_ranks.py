"""Host side of an R-rank node: R concurrent processes, each doing what ONE rank's host thread does for encode_corpus --
format_text + tokenizer + packed collation (modeling.EncodeCollator) over its own documents, no GPU involved -- under the box's CPU quota
(the GPU boxes grant 16 cores for 8 GPUs: two per rank).  The question: does R x the per-rank host rate cover R x the device rate
(~1 385 docs/s per GPU at 512 tokens for lightretriever-llama3.2-1b)?  The reference feeds its GPUs from <= 16 DataLoader worker processes
(inference/exact_search_torchrpc.py:176-203).

  python tools/bench_host_ranks.py [--ranks 8] [--docs 4096] [--batch 256] [--words 400] [--tok-threads 0]
"""
import argparse
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def usable_cores():
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def rank_worker(rank, args, start_evt, ret):
    if args.tok_threads:
        os.environ["RAYON_NUM_THREADS"] = str(args.tok_threads)
    os.environ.setdefault("TOKENIZERS_PARALLELISM", "true")
    import numpy as np
    from transformers import PreTrainedTokenizerFast
    from lightretriever_amd.modeling import EncodeCollator
    tok = PreTrainedTokenizerFast.from_pretrained(os.path.join(ROOT, "tests", "golden", "tok"))
    rng = np.random.default_rng(rank)
    letters = np.array(list("abcdefghijklmnopqrstuvwxyz"))
    words = ["".join(rng.choice(letters, size=rng.integers(2, 10))) for _ in range(5000)]
    docs = [{"title": " ".join(rng.choice(words, size=6)), "text": " ".join(rng.choice(words, size=max(1, int(rng.integers(args.words // 2, args.words + 1)))))}
            for _ in range(args.docs)]
    coll = EncodeCollator(tok, encode_is_query=False, p_max_len=512)
    coll(docs[:args.batch])                                   # warm-up (tokenizer thread pool, caches)
    start_evt.wait()
    t0 = time.perf_counter()
    n_tok = 0
    for s in range(0, args.docs, args.batch):
        n_tok += int(coll(docs[s:s + args.batch])["cu_seqlens"][-1])
    dt = time.perf_counter() - t0
    ret[rank] = (args.docs / dt, n_tok / args.docs)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--docs", type=int, default=4096)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--words", type=int, default=400)
    ap.add_argument("--tok-threads", type=int, default=0, help="RAYON_NUM_THREADS per process (0 = the tokenizers default: all visible CPUs)")
    args = ap.parse_args()
    ctx = mp.get_context("spawn")
    for ranks in sorted({1, args.ranks}):
        mgr = ctx.Manager()
        ret, evt = mgr.dict(), ctx.Event()
        procs = [ctx.Process(target=rank_worker, args=(r, args, evt, ret)) for r in range(ranks)]
        [p.start() for p in procs]
        time.sleep(8 + ranks)                                 # (imports + document synthesis + warm-up in every child)
        evt.set()
        [p.join() for p in procs]
        rates = [ret[r][0] for r in range(ranks)]
        print("%d concurrent rank process(es) on %d usable cores (tokenizer threads per process: %s): per rank %.0f .. %.0f docs/s (mean %.0f), "
              "all ranks %.0f docs/s; %.0f tokens/doc" % (ranks, usable_cores(), args.tok_threads or "default", min(rates), max(rates),
                                                          sum(rates) / ranks, sum(rates), ret[0][1]), flush=True)


if __name__ == "__main__":
    main()
