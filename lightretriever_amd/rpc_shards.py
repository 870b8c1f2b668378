"""Row-sharded search driven from ONE rank over torch RPC -- the launch shape of the reference's own eval driver.

eval/eval_utils.py:launch_eval starts one process per GPU with `rpc.init_rpc(name=f"worker{rank}")`, builds
`PytorchRPCExactSearchModel(args)` on every rank and then runs the evaluation on rank 0 only; the other ranks sit in
`rpc.shutdown()` serving calls.  The reference uses that to fan batches out and pull every embedding back to rank 0
(inference/exact_search_torchrpc.py:122-351).  Here the same launch keeps the corpus where it is encoded: rank 0 sends each
worker the TEXTS of its batches (batch j of a sorted chunk -> worker j % R), the worker encodes them into a shard in its own
HBM, later receives the query embeddings (a few hundred KB) and answers with its local top-k (Q x k pairs); rank 0 merges.
No embedding row ever crosses the RPC transport.  With an initialised `torch.distributed` group the SPMD path of
`retriever._chunked_dense_search` (RCCL all-gather) is used instead; with neither, everything runs on the calling rank."""
from __future__ import annotations

import torch

_WORKER: dict = {}      # per-process registry (the reference's MODEL_REGISTRY, exact_search_torchrpc.py:95-96)


def register_worker(model) -> None:
    """Called by PytorchRPCExactSearchModel.__init__ on every rank: the model remote calls will use."""
    _WORKER["model"] = model
    _WORKER.pop("searcher", None)


def rpc_workers() -> list:
    """Names of the RPC workers ordered by id ([] when torch RPC is not initialised in this process)."""
    try:
        from torch.distributed import rpc
        if not rpc.api._is_current_rpc_agent_set():
            return []
        return [w.name for w in sorted(rpc.api._get_current_rpc_agent().get_worker_infos(), key=lambda w: w.id)]
    except Exception:   # torch built without RPC
        return []


# ---- executed on the worker (an RPC server thread: the current device must be set explicitly) ------------------------------
def _searcher(batch_size: int):
    from .retriever import FlatIPFaissSearch
    s = _WORKER.get("searcher")
    if s is None or s.batch_size != batch_size:
        s = _WORKER["searcher"] = FlatIPFaissSearch(_WORKER["model"], batch_size=batch_size)
        s.show_progress_bar = False
    return s


def _w_index(docs: list, rows: list, dim: int, batch_size: int, capacity: int, first: bool, last: bool) -> int:
    """One piece of this worker's share of a corpus chunk: encoded straight into the shard (allocated for `capacity` rows by the first
    piece); the last piece publishes the shard to the worker's searcher with the rows' global positions as ids."""
    if "model" not in _WORKER:
        raise RuntimeError("lrx rpc worker: no model registered in this process (construct PytorchRPCExactSearchModel on every rank)")
    from .index import FlatIPIndex
    from .retriever import FaissIndex
    model = _WORKER["model"]
    dev = model.model.device
    with torch.cuda.device(dev):
        s = _searcher(batch_size)
        if first:
            s._clear()
            _WORKER["pending"] = (FlatIPIndex(dim, capacity=capacity, device=dev), [])
        idx, ids = _WORKER["pending"]
        if docs:
            slot = idx.append_slot(len(docs))
            emb = model.encode_corpus(docs, batch_size=batch_size, show_progress_bar=False, convert_to_tensor=True, out=slot, rpc_fanout=False)
            emb = emb["dense_reps"] if isinstance(emb, dict) else emb
            if emb.data_ptr() != slot.data_ptr():
                slot.copy_(emb.to(slot.device))
            idx.commit(len(docs))
            ids.extend(rows)
        if last:
            s.dim_size = dim
            s.faiss_index = FaissIndex(idx, ids if ids else None)
            _WORKER.pop("pending", None)
        torch.cuda.synchronize()
    return len(docs)


def _w_encode(items: list, batch_size: int) -> dict:
    """A contiguous span of a direct encode call, encoded on this worker; tensors travel back as CPU tensors (the caller asked for them)."""
    model = _WORKER["model"]
    with torch.cuda.device(model.model.device):
        res = model.encode(items, batch_size=batch_size, show_progress_bar=False, convert_to_tensor=True, rpc_fanout=False)
        return {k: (v.cpu() if isinstance(v, torch.Tensor) else v) for k, v in res.items()}


def _w_search(q_cpu: torch.Tensor, top_k: int, batch_size: int):
    dev = _WORKER["model"].model.device
    with torch.cuda.device(dev):
        D, I = _searcher(batch_size)._retrieve_device(q_cpu.to(dev), top_k)
        return D.cpu(), I.cpu()


def _w_clear(batch_size: int) -> None:
    s = _WORKER.get("searcher")
    if s is not None:
        s._clear()


# ---- executed on the driving rank ---------------------------------------------------------------------------------------------
PIECE_DOCS = 32768   # documents per RPC message and worker (tens of MB of text; a 10 M-document chunk must not travel as one pickle)


def index_chunk(workers: list, docs: list, first_row: int, dim: int, batch_size: int) -> None:
    """Batch j of `docs` (one sorted corpus chunk; global rows first_row ..) -> worker j % R, encoded there into a fresh shard.
    Every worker's share goes out in pieces of PIECE_DOCS documents, all workers working on their k-th piece at the same time."""
    from torch.distributed import rpc
    from .sharded import local_to_global_rows
    shares = [local_to_global_rows(len(docs), batch_size, r, len(workers)).tolist() for r in range(len(workers))]
    n_pieces = max(1, max(-(-len(sh) // PIECE_DOCS) for sh in shares))
    for k in range(n_pieces):
        futs = []   # timeout=0: a piece takes a while to encode (the RPC default would give up after 60 s)
        for name, sh in zip(workers, shares):
            part = sh[k * PIECE_DOCS:(k + 1) * PIECE_DOCS]
            futs.append(rpc.rpc_async(name, _w_index, args=([docs[i] for i in part], [first_row + i for i in part], dim, batch_size, len(sh),
                                                             k == 0, k == n_pieces - 1), timeout=0))
        for f in futs:
            f.wait()


def search_shards(workers: list, q: torch.Tensor, top_k: int, batch_size: int):
    """-> ([R,Q,k] scores, [R,Q,k] global rows) on q's device: every worker's local top-k over the shard it holds."""
    from torch.distributed import rpc
    q_cpu = q.detach().cpu()
    futs = [rpc.rpc_async(name, _w_search, args=(q_cpu, top_k, batch_size), timeout=0) for name in workers]
    parts = [f.wait() for f in futs]
    return torch.stack([p[0] for p in parts]).to(q.device), torch.stack([p[1] for p in parts]).to(q.device)


def clear_shards(workers: list, batch_size: int) -> None:
    from torch.distributed import rpc
    for f in [rpc.rpc_async(name, _w_clear, args=(batch_size,), timeout=0) for name in workers]:
        f.wait()


def encode_fanout(workers: list, items: list, batch_size: int, convert_to_tensor: bool, device) -> dict:
    """Direct `encode` / `encode_corpus` call on the driving rank: span r of the (already prompt-formatted) items -> worker r, rows
    assembled in input order (row i of the output is input i)."""
    from torch.distributed import rpc
    n, R = len(items), len(workers)
    per = -(-n // R)
    per = -(-per // batch_size) * batch_size               # whole batches per worker
    spans = [(s, min(s + per, n)) for s in range(0, n, per)]
    futs = [rpc.rpc_async(workers[i], _w_encode, args=(items[s:e], batch_size), timeout=0) for i, (s, e) in enumerate(spans)]
    parts = [f.wait() for f in futs]
    out = {}
    for k in parts[0]:
        if isinstance(parts[0][k], torch.Tensor):
            t = torch.cat([p[k] for p in parts], 0)
            out[k] = t.to(device) if convert_to_tensor else t.numpy()
        else:
            out[k] = [x for p in parts for x in p[k]]
    return out
