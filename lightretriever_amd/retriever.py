"""Host-side mirror of the reference's searcher interfaces, dense branch only.

    FaissIndex          <- retriever/faiss_index.py:20-73      (build / search / reset over the HBM-resident FlatIPIndex)
    FlatIPFaissSearch   <- retriever/faiss_search.py:46-293, :477-510   (BEIR-style dense searcher)
    HybridSearch        <- retriever/hybrid_search.py:25-403   (dense `den` / `emb` branches; sparse + fusion out of scope)

Design differences, results preserved: corpus embeddings are encoded straight into the index shard (no CPU round trip, no
`index.add` copy); the per-chunk (score, pid) heaps of hybrid_search.py:182-205 are a running [Q, top_k] list kept on the
GPU and merged with lrx_merge_topk, so Python touches O(Q*k) values once at the end instead of once per corpus chunk.
`ignore_identical_ids` keeps the reference's order of operations (per-chunk top_k first, then the qid == pid hit is dropped).
Equal scores: inside one chunk (one index) the lower row wins (this build's IndexFlatIP contract); ACROSS
chunks the reference's heap compares (score, pid) tuples, so at equal scores the LARGER pid survives -- the running list is
therefore merged on keys that order the documents by descending pid.  Pinned by tests/golden/search_ref.json (outputs of the
reference's own HybridSearch.search / FlatIPFaissSearch.search).
"""
from __future__ import annotations

import logging
import os
import time
from typing import Optional

import numpy as np
import torch

from .index import FlatIPIndex, merge_topk

logger = logging.getLogger(__name__)
FLT_MAX = float(np.finfo(np.float32).max)


class FaissIndex:
    """`FaissIndex(index, passage_ids)` of the reference: search() maps row numbers through `_passage_ids` and logs QPS."""

    def __init__(self, index: FlatIPIndex, passage_ids: Optional[list] = None):
        self.index = index
        self._passage_ids = None if passage_ids is None else torch.as_tensor(np.asarray(passage_ids, dtype=np.int64), device=index.device)

    def search(self, query_embeddings, k: int, **kwargs):
        n = query_embeddings.shape[0]
        t0 = time.time()
        scores, ids = self.index.search(query_embeddings, k)
        if self._passage_ids is not None:
            ids = torch.where(ids >= 0, self._passage_ids[ids.clamp(min=0)], ids)
        torch.cuda.synchronize(self.index.device)
        dt = max(time.time() - t0, 1e-9)
        logger.info("Num of queries: %d\tSearch time (s): %.3f\tQPS: %.3f", n, dt, n / dt)
        return scores, ids

    @classmethod
    def build(cls, passage_ids: list, passage_embeddings, index: Optional[FlatIPIndex] = None, buffer_size: int = 50000):
        if index is None:
            index = FlatIPIndex(passage_embeddings.shape[1], capacity=len(passage_ids))
        for s in range(0, len(passage_ids), buffer_size):
            index.add(passage_embeddings[s:s + buffer_size])
        return cls(index, passage_ids)

    def save(self, fname: str):
        self.index.save(fname)     # faiss.write_index layout (index_io.py)

    def to_gpu(self):
        return self.index   # already HBM-resident; multi-GPU = one process per GPU (sharded.ShardedFlatIPIndex)

    def reset(self):
        self.index.reset()


def _ids_and_list(queries):
    if isinstance(queries, dict):
        return list(queries.keys()), [queries[q] for q in queries]
    try:
        import datasets
        if isinstance(queries, datasets.Dataset):
            name = None
            for c in ["id", "_id", "query_id"]:
                if c in queries.column_names:
                    name = c
            if name is None:
                raise KeyError(f"No id column in queries dataset {queries.column_names}")
            return list(queries[name]), queries
    except ImportError:  # pragma: no cover
        pass
    raise NotImplementedError(f"Unrecognized type {type(queries)}")


def _sorted_corpus(corpus):
    """Longest text first (faiss_search.py:214-216 / hybrid_search.py:273-276); returns (corpus_ids, list of docs)."""
    if isinstance(corpus, dict):
        ids = sorted(corpus, key=lambda k: len(corpus[k].get("text", "")) if isinstance(corpus[k], dict) else len(corpus[k]), reverse=True)
        return ids, [corpus[c] for c in ids]
    try:
        import datasets
        if isinstance(corpus, datasets.Dataset):
            name = None
            for c in ["id", "_id", "docid", "doc_id"]:
                if c in corpus.column_names:
                    name = c
            if name is None:
                raise KeyError(f"No id column in corpus dataset {corpus.column_names}")
            rows = corpus.to_list()                      # one sequential read (row-by-row / permuted access costs ~30-90 us per document)
            rows.sort(key=lambda r: len(r["text"]), reverse=True)
            return [r[name] for r in rows], rows
    except ImportError:  # pragma: no cover
        pass
    raise NotImplementedError(f"Unrecognized type {type(corpus)}")


class DenseRetrievalFaissSearch:
    def __init__(self, model, batch_size: int = 128, corpus_chunk_size: Optional[int] = None, use_single_gpu: bool = False,
                 use_multiple_gpu: bool = False, **kwargs):
        self.model = model       # provides encode_corpus() and encode_queries()
        self.batch_size = batch_size
        self.corpus_chunk_size = batch_size * 800 if corpus_chunk_size is None else corpus_chunk_size
        self.show_progress_bar = kwargs.get("show_progress_bar", True)
        self.convert_to_tensor = kwargs.get("convert_to_tensor", True)
        self.faiss_index: Optional[FaissIndex] = None
        self.use_single_gpu, self.use_multiple_gpu = use_single_gpu, use_multiple_gpu
        self.dim_size = None
        self.mapping, self.rev_mapping = {}, {}
        self.mteb_model_meta = None

    @classmethod
    def name(cls):
        return "faiss_search"

    def encode(self, sentences, batch_size, show_progress_bar=True, convert_to_tensor=True, **kwargs):
        return self.model.encode(sentences=sentences, batch_size=batch_size, show_progress_bar=show_progress_bar, convert_to_tensor=convert_to_tensor, **kwargs)

    def encode_queries(self, queries, batch_size, show_progress_bar=True, convert_to_tensor=True, **kwargs):
        return self.model.encode_queries(queries=queries, batch_size=batch_size, show_progress_bar=show_progress_bar, convert_to_tensor=convert_to_tensor, **kwargs)

    def encode_corpus(self, corpus, batch_size, show_progress_bar=True, convert_to_tensor=True, **kwargs):
        return self.model.encode_corpus(corpus=corpus, batch_size=batch_size, show_progress_bar=show_progress_bar, convert_to_tensor=convert_to_tensor, **kwargs)

    def _create_mapping_ids(self, corpus_ids):
        if not all(isinstance(d, int) for d in corpus_ids):
            for i, d in enumerate(corpus_ids):
                self.mapping[d] = i
                self.rev_mapping[i] = d

    def _clear(self):
        if self.faiss_index is not None:
            self.faiss_index.reset()
        self.faiss_index = None
        self.dim_size = None
        self.mapping, self.rev_mapping = {}, {}

    def index(self, corpus_emb, corpus_ids):
        raise NotImplementedError("Base class function. Please implement this depands on index type.")

    # -- persistence (faiss_search.py:99-123): {prefix}.{ext}.tsv id map + {prefix}.{ext}.faiss; one pair per rank ------------
    mapping_tsv_keys = ["beir-docid", "faiss-docid"]

    @staticmethod
    def _rank_world():
        import torch.distributed as dist
        return (dist.get_rank(), dist.get_world_size()) if dist.is_available() and dist.is_initialized() else (0, 1)

    def _load(self, input_dir: str, prefix: str, ext: str):
        from .index_io import load_tsv_to_dict, shard_prefix
        prefix = shard_prefix(prefix, *self._rank_world())
        self.mapping = load_tsv_to_dict(os.path.join(input_dir, "{}.{}.tsv".format(prefix, ext)), header=True)
        self.rev_mapping = {v: k for k, v in self.mapping.items()}
        return os.path.join(input_dir, "{}.{}.faiss".format(prefix, ext)), sorted(self.rev_mapping)

    def save(self, output_dir: str, prefix: str, ext: str):
        from .index_io import save_dict_to_tsv, shard_prefix
        prefix = shard_prefix(prefix, *self._rank_world())
        os.makedirs(output_dir, exist_ok=True)
        save_dict_to_tsv(self.mapping, os.path.join(output_dir, "{}.{}.tsv".format(prefix, ext)), keys=self.mapping_tsv_keys)
        path = os.path.join(output_dir, "{}.{}.faiss".format(prefix, ext))
        self.faiss_index.save(path)
        logger.info("Index size: {:.2f}MB".format(os.path.getsize(path) * 0.000001))

    # -- device-level retrieval used by the chunk loop ---------------------------------------------------------------
    def _retrieve_device(self, query_emb, top_k: int):
        return self.faiss_index.search(_as_device(query_emb, self.faiss_index.index.device), top_k)

    def retrieve_with_emb(self, query_emb, query_ids: list, top_k: int, **kwargs) -> dict:
        """-> {qid: {pid: score}} (faiss_search.py:143-173).  Rows beyond the index size (id -1) are dropped."""
        scores, ids = self._retrieve_device(query_emb, top_k)
        return _to_result_dict(scores, ids, query_ids, self.rev_mapping)

    def search(self, corpus, queries, top_k: int = 1000, score_function: str = None, return_sorted: bool = False,
               ignore_identical_ids: bool = False, **kwargs) -> dict:
        query_ids, queries_list = _ids_and_list(queries)
        q = self.model.encode_queries(queries_list, batch_size=self.batch_size, show_progress_bar=self.show_progress_bar,
                                      convert_to_tensor=self.convert_to_tensor)
        if isinstance(q, dict):
            q = q["dense_reps"] if "dense_reps" in q else q["emb_reps"]
        return _chunked_dense_search(self, q, query_ids, corpus, top_k, ignore_identical_ids)


class FlatIPFaissSearch(DenseRetrievalFaissSearch):
    def index(self, corpus_emb, corpus_ids):
        """Index already-encoded embeddings (Tensor on any device / ndarray) -- faiss_search.py:490-504."""
        self._create_mapping_ids(corpus_ids)
        self.dim_size = corpus_emb.shape[1]
        rows = [self.mapping.get(c, c) for c in corpus_ids]
        self.faiss_index = FaissIndex.build(rows, corpus_emb)

    def _index_in_place(self, docs: list, corpus_ids: list, dim: int):
        """Encode a corpus chunk straight into a fresh shard (embeddings never leave HBM)."""
        self._create_mapping_ids(corpus_ids)
        self.dim_size = dim
        idx = FlatIPIndex(dim, capacity=len(docs))
        if not docs:                                  # a rank without a batch in this chunk: empty shard, searches return padding
            self.faiss_index = FaissIndex(idx, None)
            return None
        slot = idx.append_slot(len(docs))
        enc = emb = self.model.encode_corpus(docs, batch_size=self.batch_size, show_progress_bar=self.show_progress_bar,
                                             convert_to_tensor=True, out=slot)
        if isinstance(emb, dict):
            emb = emb["dense_reps"]
        if emb.data_ptr() != slot.data_ptr():       # a model that does not support `out=`: one device copy
            slot.copy_(emb.to(slot.device))
        idx.commit(len(docs))
        self.faiss_index = FaissIndex(idx, [self.mapping.get(c, c) for c in corpus_ids])
        return enc if isinstance(enc, dict) else {"dense_reps": enc}   # what encode_corpus returned (sparse_reps ride along)

    def load(self, input_dir: str, prefix: str = "my-index", ext: str = "flat"):
        """faiss_search.py:478-488: id map + index file -> HBM shard of this rank."""
        path, passage_ids = self._load(input_dir, prefix, ext)
        idx = FlatIPIndex.load(path)
        if passage_ids and len(passage_ids) != idx.ntotal:
            raise ValueError(f"{path}: {idx.ntotal} rows but {len(passage_ids)} ids in the map")
        self.dim_size = idx.d
        self.faiss_index = FaissIndex(idx, passage_ids or None)

    def save(self, output_dir: str, prefix: str = "my-index", ext: str = "flat"):
        super().save(output_dir, prefix, ext)

    def get_index_name(self):
        return "flat_faiss_index"


class HybridSearch:
    """Dense half of the reference's HybridSearch: routes `dense_reps` -> results["den"], `emb_reps` -> results["emb"]
    (hybrid_search.py:121-180); `search()` returns the last enabled type unless return_all_results."""

    def __init__(self, model, batch_size: int = 128, corpus_chunk_size: Optional[int] = None, use_multiple_gpu: bool = False,
                 score_fuse_method: str = "linear", fuse_weights=(0.7, 0.3), return_all_results: bool = False, sparse_search=None, **kwargs):
        """sparse_search: an engine with the reference's AnseriniSearch interface (`index(corpus_emb, corpus_ids)`,
        `retrieve_with_emb(query_emb, query_ids, top_k)`, `_clear()`); the Lucene engine itself is outside this package.  When
        one is given, `tok` / `emb_tok` (query token counts x sparse document vectors, and their fusion with the dense hits,
        hybrid_search.py:160-180) are produced like the reference does; the fusion runs on the GPU (score_fuse_utils)."""
        self.model = model
        self.score_fuse_method = score_fuse_method
        self.fuse_weights = list(fuse_weights)
        self.sparse_search = sparse_search
        self.batch_size = batch_size
        self.corpus_chunk_size = batch_size * 800 if corpus_chunk_size is None else corpus_chunk_size
        self.show_progress_bar = kwargs.get("show_progress_bar", True)
        self.convert_to_tensor = kwargs.get("convert_to_tensor", True)
        self.dense_search = FlatIPFaissSearch(model, batch_size=batch_size, corpus_chunk_size=corpus_chunk_size, use_multiple_gpu=use_multiple_gpu)
        self.return_all_results = return_all_results
        self.mteb_model_meta = None

    @classmethod
    def name(cls):
        return "hybrid_search"

    def encode(self, sentences, batch_size, **kw):
        return self.model.encode(sentences=sentences, batch_size=batch_size, **kw)

    def encode_queries(self, queries, batch_size, **kw):
        return self.model.encode_queries(queries=queries, batch_size=batch_size, **kw)

    def encode_corpus(self, corpus, batch_size, **kw):
        return self.model.encode_corpus(corpus=corpus, batch_size=batch_size, **kw)

    def _clear(self, dense: bool = True, sparse: bool = True):
        if dense:
            self.dense_search._clear()
        if sparse and self.sparse_search is not None:
            self.sparse_search._clear()

    def index(self, corpus_emb: dict, corpus_ids: list):
        assert isinstance(corpus_emb, dict) and corpus_emb.get("dense_reps") is not None
        self.dense_search.index(corpus_emb["dense_reps"], corpus_ids)
        if self.sparse_search is not None and corpus_emb.get("sparse_reps") is not None:
            self.sparse_search.index(corpus_emb["sparse_reps"], corpus_ids)

    def _fuse_results(self, dense_results=None, sparse_results=None, weights=(0.7, 0.3)):
        """hybrid_search.py:207-232: one of the lists alone, or their RRF / min-max linear fusion (on the GPU)."""
        from .score_fuse_utils import fuse_scores_linear, fuse_scores_rrf
        if dense_results is None and sparse_results is None:
            raise ValueError("All scores are None. Please check model settings.")
        if dense_results is None:
            return sparse_results
        if sparse_results is None:
            return dense_results
        if self.score_fuse_method == "rrf":
            return fuse_scores_rrf([dense_results, sparse_results])
        if self.score_fuse_method == "linear":
            return fuse_scores_linear([dense_results, sparse_results], weights=weights)
        raise NotImplementedError(f"score_fuse_method {self.score_fuse_method} is not supported.")

    def retrieve_with_emb(self, query_emb: dict, query_ids: list, top_k: int, dense: bool = True, sparse: bool = True, **kwargs):
        assert isinstance(query_emb, dict) and (query_emb.get("dense_reps") is not None or query_emb.get("emb_reps") is not None)
        results = {}
        if dense:
            if query_emb.get("dense_reps") is not None:
                results["den"] = self.dense_search.retrieve_with_emb(query_emb["dense_reps"], query_ids, top_k=top_k)
            if query_emb.get("emb_reps") is not None:
                results["emb"] = self.dense_search.retrieve_with_emb(query_emb["emb_reps"], query_ids, top_k=top_k)
        if sparse and self.sparse_search is not None and query_emb.get("token_id_reps") is not None:
            results["tok"] = self.sparse_search.retrieve_with_emb(query_emb["token_id_reps"], query_ids, top_k=top_k)
        if sparse and self.sparse_search is not None and query_emb.get("sparse_reps") is not None:       # LM-head sparse queries (hybrid_search.py:166-172)
            results["spr"] = self.sparse_search.retrieve_with_emb(query_emb["sparse_reps"], query_ids, top_k=top_k)
            if "den" in results:
                results["den_spr"] = self._fuse_results(results["den"], results["spr"], weights=self.fuse_weights)
        if "emb" in results and "tok" in results:
            results["emb_tok"] = self._fuse_results(results["emb"], results["tok"], weights=self.fuse_weights)
        return results

    def search(self, corpus, queries, top_k: int = 1000, score_function: str = None, return_sorted: bool = False,
               ignore_identical_ids: bool = False, **kwargs):
        query_ids, queries_list = _ids_and_list(queries)
        qe = self.model.encode_queries(queries_list, batch_size=self.batch_size, show_progress_bar=self.show_progress_bar,
                                       convert_to_tensor=self.convert_to_tensor)
        assert isinstance(qe, dict) and any(k in qe for k in ("dense_reps", "emb_reps", "sparse_reps", "token_id_reps"))
        results, default = {}, None
        # one corpus pass serves every enabled query representation (they share the document `dense_reps`)
        kinds = [(k, name) for k, name in (("dense_reps", "den"), ("emb_reps", "emb")) if qe.get(k) is not None]
        # sparse half (hybrid_search.py:330-375): every chunk's document vectors go into the engine as they are encoded, the engine is
        # searched once at the end with the queries' token-id counts, `emb_tok` is the fusion of the two final hit lists
        use_tok = self.sparse_search is not None and qe.get("token_id_reps") is not None
        use_spr = self.sparse_search is not None and qe.get("sparse_reps") is not None      # LM-head sparse query vectors (pseudo text), hybrid_search.py:364-369
        on_chunk = None
        if use_tok or use_spr:
            def on_chunk(chunk_ids, enc):
                assert enc.get("sparse_reps") is not None, "sparse engine given but encode_corpus returned no sparse_reps"
                self.sparse_search.index(enc["sparse_reps"], list(chunk_ids))
        multi = _chunked_dense_search(self.dense_search, [qe[k] for k, _ in kinds], query_ids, corpus, top_k, ignore_identical_ids, on_chunk=on_chunk)
        dense_res = dict(zip([name for _, name in kinds], multi))
        tok_res = self.sparse_search.retrieve_with_emb(query_emb=qe["token_id_reps"], query_ids=query_ids, top_k=top_k) if use_tok else None
        spr_res = self.sparse_search.retrieve_with_emb(query_emb=qe["sparse_reps"], query_ids=query_ids, top_k=top_k) if use_spr else None
        # result names and the default (the LAST one set) in the reference's order (hybrid_search.py:380-403): den, spr, emb, tok, den_spr, emb_tok
        if "den" in dense_res:
            results["den"] = default = dense_res["den"]
        if use_spr:
            results["spr"] = default = spr_res
        if "emb" in dense_res:
            results["emb"] = default = dense_res["emb"]
        if use_tok:
            results["tok"] = default = tok_res
        if "den" in dense_res and use_spr:
            results["den_spr"] = default = self._fuse_results(dense_res["den"], spr_res, weights=self.fuse_weights)
        if "emb" in dense_res and use_tok:
            results["emb_tok"] = default = self._fuse_results(dense_res["emb"], tok_res, weights=self.fuse_weights)
        self._clear()
        return results if self.return_all_results else default


# ------------------------------------------------------------------------------------------------------------------
def _as_device(x, device):
    if not isinstance(x, torch.Tensor):
        x = torch.from_numpy(np.ascontiguousarray(x))
    return x.to(device=device, dtype=torch.float32).contiguous()


def _to_result_dict(scores: torch.Tensor, ids: torch.Tensor, query_ids: list, rev_mapping) -> dict:
    """[Q, k] device arrays -> {qid: {pid: score}} (retriever/faiss_search.py:165-171, hybrid_search.py:347-355).  rev_mapping: row -> pid as a
    dict or a sequence (empty / None: the row number as a string).  One D2H copy, then per query an object-array gather of the pids and
    dict(zip(...)) over Python lists -- no per-hit Python arithmetic (1000 queries x top-1000: 0.21 s instead of 0.79 s; the nested dict is
    the reference's return type)."""
    S, I = scores.cpu().numpy(), ids.cpu().numpy()
    names = None
    if rev_mapping is not None and len(rev_mapping):
        if isinstance(rev_mapping, dict):
            names = np.empty(max(rev_mapping) + 1, dtype=object)
            names[list(rev_mapping.keys())] = list(rev_mapping.values())
        else:
            names = np.empty(len(rev_mapping), dtype=object)
            names[:] = list(rev_mapping)
    out = {}
    for qi, qid in enumerate(query_ids):
        r, sc = I[qi], S[qi]
        keep = r >= 0
        if not keep.all():
            r, sc = r[keep], sc[keep]
        keys = names[r].tolist() if names is not None else [str(x) for x in r.tolist()]
        out[qid] = dict(zip(keys, sc.tolist()))
    return out


def _chunked_dense_search(searcher: FlatIPFaissSearch, query_embs, query_ids: list, corpus, top_k: int, ignore_identical_ids: bool,
                          on_chunk=None):
    """Chunk loop of faiss_search.py:228-291 / hybrid_search.py:301-358: encode chunk -> index -> retrieve -> merge.
    `query_embs` may be one tensor or a list of tensors (several query representations scored against the same docs).
    `on_chunk(corpus_ids_of_chunk, encode_corpus_result)`: called after each chunk is encoded (the sparse half of
    HybridSearch.index, hybrid_search.py:330-331); single-process launch only."""
    single = not isinstance(query_embs, (list, tuple))
    qlist = [query_embs] if single else list(query_embs)
    corpus_ids, docs = _sorted_corpus(corpus)
    n = len(docs)
    device = searcher.model.model.device if hasattr(searcher.model, "model") and hasattr(searcher.model.model, "device") else torch.device("cuda", torch.cuda.current_device())
    qlist = [_as_device(q, device) for q in qlist]
    dim = qlist[0].shape[1]
    Q = len(query_ids)
    row_of = {c: i for i, c in enumerate(corpus_ids)}
    ident = torch.tensor([row_of.get(q, -2) for q in query_ids], dtype=torch.int64, device=device) if ignore_identical_ids else None
    run_D = [torch.full((Q, top_k), -FLT_MAX, dtype=torch.float32, device=device) for _ in qlist]
    run_I = [torch.full((Q, top_k), -1, dtype=torch.int64, device=device) for _ in qlist]
    # cross-chunk ties (hybrid_search.py:182-205: heapq on (score, pid) tuples keeps the larger pid): the running list carries
    # key = position of the document in DESCENDING pid order, so the merge's (score desc, key asc) is the heap's order
    key_of_row = row_of_key = None
    if n > searcher.corpus_chunk_size:
        try:
            order = sorted(range(n), key=corpus_ids.__getitem__, reverse=True)
        except TypeError:                             # ids of mixed types do not compare (the reference's heap would raise on a tie)
            order = None
        if order is not None:
            row_of_key = torch.tensor(order, dtype=torch.int64, device=device)
            key_of_row = torch.empty_like(row_of_key)
            key_of_row[row_of_key] = torch.arange(n, dtype=torch.int64, device=device)
    rank, world = DenseRetrievalFaissSearch._rank_world()
    if world > 1:
        from .sharded import exchange_merge, local_to_global_rows
        if on_chunk is not None:
            raise NotImplementedError("a sparse engine needs the chunk's sparse vectors on the calling rank: single-process launch only")
    # the reference's own launch (eval/eval_utils.py: torch RPC, only rank 0 drives): shards live on the RPC workers
    rpc_names = []
    if world == 1:
        from . import rpc_shards
        rpc_names = rpc_shards.rpc_workers()
        if len(rpc_names) <= 1 or "model" not in rpc_shards._WORKER or on_chunk is not None:
            rpc_names = []                            # (with a sparse engine the calling rank encodes everything itself)
    for s in range(0, n, searcher.corpus_chunk_size):
        e = min(s + searcher.corpus_chunk_size, n)
        logger.info("Encoding Batch %d/%d...", s // searcher.corpus_chunk_size + 1, -(-n // searcher.corpus_chunk_size))
        if world > 1:
            # one process per GPU: batch j of this chunk's sorted documents belongs to rank j % world; every rank
            # encodes its batches into its own HBM shard whose rows carry their global sorted position
            rows = (local_to_global_rows(e - s, searcher.batch_size, rank, world) + s).tolist()
            searcher._index_in_place([docs[i] for i in rows], rows, dim)
        elif rpc_names:
            rpc_shards.index_chunk(rpc_names, docs[s:e], s, dim, searcher.batch_size)   # texts out, nothing back
        else:
            enc = searcher._index_in_place(docs[s:e], list(range(s, e)), dim)   # rows carry their global sorted position
            if on_chunk is not None and enc is not None:
                on_chunk(corpus_ids[s:e], enc)
        for j, q in enumerate(qlist):
            if rpc_names:                             # local top-k of every worker's shard (Q x k pairs each), merged here
                D, I = merge_topk(*rpc_shards.search_shards(rpc_names, q, top_k, searcher.batch_size))
            else:
                D, I = searcher._retrieve_device(q, top_k)
            if world > 1:                             # one all-gather of the packed per-shard lists, merge on every rank
                D, I = exchange_merge(D, I)
            if ident is not None:                     # drop the qid == pid hit AFTER the per-chunk top_k, like the reference
                hit = I == ident[:, None]
                D = torch.where(hit, torch.full_like(D, -FLT_MAX), D)
                I = torch.where(hit, torch.full_like(I, -1), I)
            if key_of_row is not None:
                I = torch.where(I >= 0, key_of_row[I.clamp(min=0)], I)
            run_D[j], run_I[j] = merge_topk(torch.stack([run_D[j], D]), torch.stack([run_I[j], I]))
        if rpc_names:
            rpc_shards.clear_shards(rpc_names, searcher.batch_size)
        searcher._clear()
    if row_of_key is not None:
        run_I = [torch.where(i >= 0, row_of_key[i.clamp(min=0)], i) for i in run_I]
    outs = [_to_result_dict(d, i, query_ids, corpus_ids) for d, i in zip(run_D, run_I)]
    return outs[0] if single else outs
