"""Multi-GPU (RCCL) execution of the sharded search path -- the file sorts FIRST among the `-m gpu` files so that on a multi-GPU box the
children are started before anything in this pytest process has touched a GPU (`torch.cuda.device_count()` does not initialise it).

  * on ANY GPU box: the worker script (tests/dist_workers/rccl_cases.py) under `torch.distributed.run --nproc-per-node 1` with the exchange
    forced -- the same code, one RCCL rank: proves the script and the launch shape;
  * gated on `torch.cuda.device_count() >= 2`: the same script with R = 2 and 4 ranks as available -- ShardedFlatIPIndex.search over RCCL ==
    one index bit for bit (k = 100 and 1000, a row count no R divides, contiguous and interleaved shards, a duplicate pair across shards),
    pipeline.SearchLanes over the communicator, HybridSearch.search over RCCL ranks == single process -- and `bench.py --gpus R`, whose
    line must report `rccl_ranks == R` and the per-rank shard sizes.

Replaces retriever/faiss_index.py:60-70 (Faiss IndexShards over all GPUs) and the RPC fan-out of inference/exact_search_torchrpc.py:243-328.
A failing rank exits non-zero (torch.distributed.run reports it); no process that has initialised a GPU is ever re-exec'ed."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "dist_workers", "rccl_cases.py")

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "LRX_FORCE_COLLECTIVE")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def _run(cmd, timeout, extra_env=None):
    """child in its own process group; on a timeout the whole group (launcher + ranks) is killed by its group id -- never by pattern"""
    import signal
    p = subprocess.Popen(cmd, env=dict(_env(), **(extra_env or {})), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
    try:
        out, err = p.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        os.killpg(p.pid, signal.SIGKILL)
        out, err = p.communicate()
        pytest.fail("timed out after %d s: %s\n%s\n%s" % (timeout, " ".join(cmd[-6:]), out[-2000:], err[-4000:]))
    return p.returncode, out, err


def _run_ranks(n_ranks, cases, timeout=420):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), WORKER, cases]
    rc, out, err = _run(cmd, timeout)
    assert rc == 0, "rank failure (rc %d)\n%s\n%s" % (rc, out[-3000:], err[-6000:])
    assert "RCCL_ALL_OK ranks=%d" % n_ranks in out, out[-3000:]
    return [l for l in out.splitlines() if l.startswith("RCCL_CASE_OK")]


def _ranks_available():
    n = torch.cuda.device_count()                    # (no GPU initialisation on this image)
    return [r for r in (2, 4) if r <= n]             # (at most 4 ranks: the GPU boxes allow few processes on the cards at once; 8 is the driver's bench)


def test_worker_script_with_one_rccl_rank_and_the_exchange_forced():
    notes = _run_ranks(1, "sharded,lanes,hybrid")
    assert len(notes) == 3, notes
    print("\n".join(notes))


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs >= 2 GPUs (RCCL ranks on distinct devices)")
@pytest.mark.parametrize("n_ranks", [2, 4])
def test_sharded_search_lanes_and_hybrid_search_over_rccl_ranks(n_ranks):
    if n_ranks not in _ranks_available():
        pytest.skip("only %d GPU(s) visible" % torch.cuda.device_count())
    notes = _run_ranks(n_ranks, "sharded,lanes,hybrid")
    assert len(notes) == 3, notes
    print("\n".join(notes))


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs >= 2 GPUs (RCCL ranks on distinct devices)")
def test_bench_line_reports_the_rccl_ranks_and_the_shard_sizes():
    n_ranks = max(_ranks_available())
    rows = 1_000_003                                 # no R divides it: the remainder rows go one each to the first ranks
    rc, out, err = _run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n_ranks), "--steps", "2", "--warmup", "1", "--legs", "encode,search,sharded",
                         "--index-rows", str(rows), "--sharded-rows", "2000003"], 900)
    assert rc == 0, err[-6000:]
    line = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == n_ranks and line["rccl_ranks"] == n_ranks and line["config"]["parallelism"] == "dp%d" % n_ranks
    s = line["search"]
    assert s["rccl_ranks"] == n_ranks and len(s["shard_rows_per_rank"]) == n_ranks and sum(s["shard_rows_per_rank"]) == rows
    assert max(s["shard_rows_per_rank"]) - min(s["shard_rows_per_rank"]) <= 1 and s["value"] > 0
    assert s["two_in_flight"].get("identical_to_one_at_a_time") is True, s["two_in_flight"]     # the lanes ran over the communicator
    assert line["search_qps"] == s["value"] and line["headline"]["rccl_ranks"] == n_ranks
    # round 6: BASELINE configs[3] / configs[4] over the communicator + the 8B encoder under weak scaling
    for key in ("config3_10Mx4096", "config4_10Mx256_mrl"):
        c = line["configs"][key]
        assert c["rccl_ranks"] == n_ranks and sum(c["shard_rows_per_rank"]) == 2_000_003 and max(c["shard_rows_per_rank"]) - min(c["shard_rows_per_rank"]) <= 1
        assert c["queries_per_s"] > 0 and c["exchange_ms"] > 0 and c["two_in_flight"].get("identical_to_one_at_a_time") is True, c
    assert line["configs"]["config3_encode_llama31_8b"]["n_gpus"] == n_ranks


def test_bench_with_two_ranks_on_one_gpu_over_gloo_runs_every_branch_of_the_n_gt_1_path():
    """Round 6: the driver's N > 1 launch shape (`torch.distributed.run --nproc-per-node N bench.py --gpus N`) with TWO ranks on any box --
    both on device 0, the process group over gloo (RCCL refuses two ranks per device; LRX_BENCH_BACKEND / LRX_BENCH_ONE_GPU are rehearsal
    switches, the numbers mean nothing).  What it proves that the one-rank RCCL rehearsal cannot: the `world > 1` branches of bench.py (shards
    by shard_split, rank != 0 leaving after the collectives, the headline index dropped before the sharded legs, BASELINE configs[3] / [4]
    and the 8B encoder over a two-rank communicator, lanes over it) run to a well-formed line.  Replaces retriever/faiss_index.py:60-70."""
    rows, srows = 200_003, 300_001
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--legs", "encode,search,sharded", "--index-rows", str(rows),
           "--sharded-rows", str(srows)]
    rc, out, err = _run(cmd, 1100, {"LRX_BENCH_BACKEND": "gloo", "LRX_BENCH_ONE_GPU": "1"})
    assert rc == 0, err[-6000:]
    line = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["config"]["parallelism"] == "dp2" and line["config"]["global_batch"] == 512
    s = line["search"]
    assert s["shard_rows_per_rank"] == [100_002, 100_001] and s["rccl_ranks"] == 2 and s["value"] > 0
    assert s["two_in_flight"].get("identical_to_one_at_a_time") is True, s["two_in_flight"]
    assert set(s["other_query_counts"]) == {"1", "1000"} and s["other_query_counts"]["1000"]["roofline"]["bound"] == "mfma"
    for key, dim in (("config3_10Mx4096", 4096), ("config4_10Mx256_mrl", 256)):
        c = line["configs"][key]
        assert c["rccl_ranks"] == 2 and c["shard_rows_per_rank"] == [150_001, 150_000] and c["dim"] == dim and c["queries_per_s"] > 0
        assert c["exchange"].startswith("all_gather_into_tensor over gloo") and c["exchange_ms"] > 0 and c["merge_ms"] > 0
        assert c["two_in_flight"].get("identical_to_one_at_a_time") is True, c["two_in_flight"]
    e8 = line["configs"]["config3_encode_llama31_8b"]
    assert e8["n_gpus"] == 2 and e8["docs_per_s"] > 0
    assert list(line)[-1] == "headline" and line["headline"]["rccl_ranks"] == 2
