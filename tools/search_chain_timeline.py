#!/usr/bin/env python3
"""One search chain as the GPU saw it: run under rocprofv3 --kernel-trace by tools/exp/chain_timeline.sh, then called with the trace to
print, for the LAST of a run of back-to-back searches, every launch with its start (from the chain's first launch), duration and the gap
to the previous kernel's end -- plus per-kernel averages over all chains.

  python tools/search_chain_timeline.py run                (the workload; env N, D, Q, K, EXCHANGE=1 for the 1-rank RCCL exchange)
  python tools/search_chain_timeline.py show <kernel_trace.csv>"""
import collections, csv, os, sys

def run():
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    from lightretriever_amd import FlatIPIndex
    from lightretriever_amd.sharded import ShardedFlatIPIndex
    N, D, Q, K = (int(os.environ.get(k, v)) for k, v in (("N", 125000), ("D", 2048), ("Q", 100), ("K", 100)))
    g = torch.Generator(device="cuda").manual_seed(7)
    idx = FlatIPIndex(D, capacity=N)
    slot = idx.append_slot(N)
    for s in range(0, N, 65536):
        e = min(s + 65536, N)
        slot[s:e] = torch.nn.functional.normalize(torch.randn(e - s, D, generator=g, device="cuda"), dim=-1)
    idx.commit(N)
    q = torch.nn.functional.normalize(torch.randn(Q, D, generator=g, device="cuda"), dim=-1)
    fn = lambda: idx.search(q, K)
    if os.environ.get("EXCHANGE") == "1":
        import torch.distributed as dist
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29700 + os.getpid() % 200), HSA_ENABLE_IPC_MODE_LEGACY="0", LRX_FORCE_COLLECTIVE="1")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        sh = ShardedFlatIPIndex(idx)
        fn = lambda: sh.search(q, K)
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(40):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print("N=%d D=%d Q=%d K=%d%s: %.4f ms per search (40 back to back)" % (N, D, Q, K, " + exchange" if os.environ.get("EXCHANGE") == "1" else "", e0.elapsed_time(e1) / 40))

def show(path):
    rows = [r for r in csv.DictReader(open(path))]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    first = lambda n: "k_pack_queries_xb" in n
    starts = [i for i, r in enumerate(rows) if first(r["Kernel_Name"])]
    if len(starts) < 3:
        print("no search chains found"); return
    a, b = starts[-2], starts[-1]                      # the last complete chain
    chain = rows[a:b]
    t0 = int(chain[0]["Start_Timestamp"])
    prev_end = None
    print("   %-64s %9s %9s %7s" % ("kernel", "start us", "dur us", "gap us"))
    for r in chain:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print("   %-64s %9.1f %9.1f %7s" % (r["Kernel_Name"][:64], (s - t0) / 1e3, (e - s) / 1e3, "" if prev_end is None else "%.1f" % ((s - prev_end) / 1e3)))
        prev_end = e
    print("   chain: first launch -> end of last kernel %.1f us; next chain starts %.1f us after that" % ((prev_end - t0) / 1e3, (int(rows[b]["Start_Timestamp"]) - prev_end) / 1e3))
    agg = collections.defaultdict(list)
    for r in rows[starts[5]:]:
        agg[r["Kernel_Name"][:64]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print("   averages over %d chains:" % (len(starts) - 5))
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        print("     %-64s x%-4d avg %7.1f us" % (k, len(v), sum(v) / len(v)))

if __name__ == "__main__":
    run() if sys.argv[1] == "run" else show(sys.argv[2])
