"""`InferenceArguments` + `PytorchRPCExactSearchModel` under the reference's names (inference/arguments.py:19-157,
inference/exact_search_torchrpc.py:49-101): the vector-type flag sets of eval/README.md:13-52 -- symmetric dense
(`--hybrid_use_dense_vector`), asymmetric dense (`--hybrid_use_emb_vector [--noncontextual_query_embedding]`), asymmetric sparse
(`--hybrid_use_token_id_vector`) and their combinations -- with the defaults of the reference's argument classes.

No RPC: one process per GPU under torchrun; every rank constructs this object, encodes its own share and keeps the
embeddings in its HBM shard (lightretriever_amd.sharded).  The constructor signature, attribute names and the
encode_queries / encode_corpus / encode surface are the reference's."""
from __future__ import annotations

import os
from dataclasses import dataclass, field
from typing import Optional

import torch

from .loader import default_special_tokens, encoder_from_pretrained, load_model_args, load_tokenizer
from .modeling import LrxExactSearchModel, LrxHybridModel


@dataclass
class InferenceArguments:
    model_name_or_path: Optional[str] = None
    # inference/arguments.py:27 + exact_search_torchrpc.py:84 `_MODEL_CLS[model_type]`: "EncoderModel" (the default) = the symmetric dense
    # encoder (queries and documents through the LM, bare tensors, hybrid_* flags not read); "HybridModel" = the vector types the hybrid_*
    # flags select, dict results.  Both are the same kernels here.
    model_type: str = "EncoderModel"
    inference_arch: str = "PytorchRPCExactSearchModel"
    batch_size: int = 64
    append_prompt_sep: bool = False
    q_max_len: int = 128
    p_max_len: int = 512
    # (inference/arguments.py:68-73.  The HIP encoder has ONE arithmetic: bf16 MFMA operands, fp32 accumulation, fp32 residual stream, fp16
    # q|k|v|P; `--bf16` and `--fp16` are accepted and recorded in `dtype`, neither selects a different kernel: the result is within 1e-3
    # cosine of the fp32 model either way -- closer to it than the reference's own bf16 run)
    bf16: bool = False
    fp16: bool = False
    seed: int = 42
    attn_implementation: str = "flash_attention_2"   # accepted for CLI compatibility; the HIP path has one implementation
    cumulative_seq: bool = True                         # packed varlen is the only layout here
    liger_kernel: bool = False
    # model args that change the dense-path numerics (finetune/arguments.py:75-335)
    pooling_strategy: Optional[str] = None             # finetune/arguments.py:85-90: None is served as the released models' 'lasttoken'; cls / mean / second_to_last / third_to_last too
    score_function: str = "cos_sim"
    dense_shrink_dim: Optional[int] = None
    lowercase: bool = False
    edit_tokenizer_normalizers: bool = True
    edit_tokenizer_post_processor: bool = True
    add_bos_num: int = -1
    add_eos_num: int = -1
    add_pad_token: bool = True
    pad_token: str = "<|pad|>"
    add_sep_token: bool = False
    sep_token: str = "<|sep|>"
    hybrid_use_dense_vector: bool = False              # finetune/arguments.py:175-195: all four vector types and the EmbeddingBag
    hybrid_use_emb_vector: bool = False                # switch are off by default; eval/README.md:13-52 picks them
    noncontextual_query_embedding: bool = False
    noncontextual_prompt_prefix: Optional[str] = None
    eval_batch_size_embedding_bag: int = 5000
    # sparse document vectors (finetune/arguments.py:203-290): passages through the LM head, queries as token-id counts
    hybrid_use_sparse_vector: bool = False
    hybrid_use_token_id_vector: bool = False
    token_id_vector_type: str = "sum"
    sparse_use_max_aggregation: bool = True
    sparse_use_relu: bool = False
    sparse_use_log_saturation: bool = False
    sparse_min_tokens_to_keep: int = 8
    sparse_top_p_psg: float = 1.0
    sparse_top_k_psg: int = 0
    sparse_top_p_qry: float = 1.0                       # the *_qry ratios of get_sparse_emb: applied to LM-head sparse query vectors (--hybrid_use_sparse_vector)
    sparse_top_k_qry: int = 0
    normalize: Optional[bool] = None                    # None -> derived from score_function
    # accepted for CLI compatibility with the reference's argument classes; the options below marked (*) must keep their default
    hybrid_model_architecture: str = "gpt"              # (*)
    untie_encoder: bool = False                         # (*)
    enable_bidirectional_attention: bool = False        # (*)
    use_sparse_linear_projector: bool = False           # (*)
    use_sparse_down_projector: bool = False             # (*)
    use_icu_word_pretokenizer: bool = False             # (*)
    sparse_remove_stopwords: bool = False               # (*)
    sparse_pool_from_unique_token_ids: bool = False     # (*)
    sparse_pool_from_original_input_ids_qry: bool = False   # served: only the sequence's own tokens keep their aggregated logit (modeling_hybrid.py:175-180)
    sparse_pool_from_original_input_ids_psg: bool = False
    sparse_pooling_strategy: Optional[str] = None
    pad_to_multiple_of: Optional[int] = None
    max_length: int = 1024                              # reranker only
    pad_to_max_length: bool = False                     # packed varlen never pads
    padding: Optional[str] = None
    anserini_lang: Optional[str] = None                 # sparse engine (out of scope, handed through to `sparse_search`)
    anserini_vector_type: str = "JsonVectorCollection"
    anserini_pretokenized: bool = True
    anserini_impact_search: bool = True
    anserini_bm25_k1: float = 0.9
    anserini_bm25_b: float = 0.4
    # rank wiring (env, inference/arguments.py:140-150)
    local_rank: int = -1
    rank: int = -1
    world_size: int = 0
    master_addr: str = "127.0.0.1"
    master_port: int = 12345
    debug: bool = False

    def __post_init__(self):
        for name, env in (("local_rank", "LOCAL_RANK"), ("rank", "RANK"), ("world_size", "WORLD_SIZE")):
            if os.getenv(env):
                setattr(self, name, int(os.environ[env]))
        if os.getenv("MASTER_ADDR"):
            self.master_addr = os.environ["MASTER_ADDR"]
        if os.getenv("MASTER_PORT"):
            self.master_port = os.environ["MASTER_PORT"]
        self.dtype = torch.bfloat16 if self.bf16 else (torch.float16 if self.fp16 else None)
        if self.normalize is None:
            self.normalize = self.score_function == "cos_sim"   # finetune/arguments.py:312-317
        self.pad_token, self.sep_token = default_special_tokens(self.model_name_or_path, self.pad_token, self.sep_token)
        if self.pooling_strategy not in (None, "lasttoken", "cls", "mean", "second_to_last", "third_to_last", "avg_first_last", "avg_top2"):
            raise NotImplementedError(f"--pooling_strategy {self.pooling_strategy}: finetune/dense_pooling.py:12-82 knows lasttoken (the released models), "
                                      "cls, mean, second_to_last, third_to_last, avg_first_last and avg_top2 ('none' returns no vector)")
        if self.fp16 and self.bf16:
            raise ValueError("--bf16 and --fp16 are mutually exclusive (inference/arguments.py:68-73)")
        # options whose non-default value selects a part of the reference this path does not implement: fail loudly, never silently
        for name, default in (("hybrid_model_architecture", "gpt"), ("untie_encoder", False), ("enable_bidirectional_attention", False),
                              ("use_sparse_linear_projector", False), ("use_sparse_down_projector", False), ("use_icu_word_pretokenizer", False),
                              ("sparse_remove_stopwords", False), ("sparse_pool_from_unique_token_ids", False),
                              ("sparse_use_max_aggregation", True)):
            if getattr(self, name) != default:
                raise NotImplementedError(f"--{name}={getattr(self, name)!r}: not implemented by the MI355X path (one tied gpt-style encoder, "
                                          f"LM-head max aggregation for the sparse document vector)")
        if self.model_type not in ("EncoderModel", "HybridModel"):
            raise NotImplementedError(f"--model_type {self.model_type}: the MI355X path serves EncoderModel and HybridModel")
        if self.model_type == "HybridModel" and not (self.hybrid_use_dense_vector or self.hybrid_use_emb_vector or self.hybrid_use_sparse_vector
                                                     or self.hybrid_use_token_id_vector):
            raise ValueError("no vector type selected: pass at least one of --hybrid_use_dense_vector / --hybrid_use_emb_vector "
                             "[--noncontextual_query_embedding] / --hybrid_use_token_id_vector (eval/README.md:13-30)")
        hybrid = self.model_type == "HybridModel"
        self.encode_sparse = hybrid and (self.hybrid_use_sparse_vector or self.hybrid_use_token_id_vector)   # modeling_hybrid.py:241-245


def arguments_from_checkpoint(model_name_or_path: str, **overrides) -> "InferenceArguments":
    """InferenceArguments resumed from the checkpoint's model_args.yaml (EncoderModel._load_model_args,
    finetune/modeling_encoder.py:635-656): every field this path knows takes the value the retriever was trained with, extra keys
    are ignored (allow_extra_keys=True in the reference), explicit keyword overrides win."""
    import dataclasses
    saved = load_model_args(model_name_or_path)
    known = {f.name for f in dataclasses.fields(InferenceArguments)}
    kw = {k: v for k, v in saved.items() if k in known and v is not None}
    # (`--hybrid_use_sparse_vector` -- LM-head sparse QUERY vectors next to the sparse document vectors, modeling_hybrid.py:404-438 -- is served
    # as saved since round 6, like `--hybrid_use_dense_vector` and `--hybrid_use_token_id_vector`.)
    kw.update(overrides)
    kw["model_name_or_path"] = model_name_or_path
    kw.setdefault("model_type", "HybridModel")     # (model_args.yaml holds the training ModelArguments: no model_type; this mirrors HybridModel.load)
    return InferenceArguments(**kw)


def host_threads_per_rank() -> int:
    """CPU threads this rank's tokenizer should use: the cores this process may run on (affinity mask capped by the cgroup quota) divided by
    the ranks of the node.  The Rust tokenizer parallelises a batch over a rayon pool of ALL visible CPUs by default; with one process per
    GPU that is 8 pools of 256 threads on a 16-core quota (tools/bench_host_ranks.py).  The reference sizes its DataLoader worker pool the
    same way (inference/exact_search_torchrpc.py:176-203)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    ranks = int(os.environ.get("LOCAL_WORLD_SIZE") or os.environ.get("WORLD_SIZE") or 1)
    return max(1, n // max(1, ranks))


class PytorchRPCExactSearchModel(LrxExactSearchModel):
    """Drop-in for eval/eval_utils.py:179 `PytorchRPCExactSearchModel(args)`."""

    @classmethod
    def load(cls, model_name_or_path: str, args: Optional[InferenceArguments] = None, **overrides) -> "PytorchRPCExactSearchModel":
        """HybridModel.load(path, model_args=None) (finetune/modeling_hybrid.py:909-957): without arguments the flags come from the
        checkpoint's own model_args.yaml."""
        return cls(args if args is not None else arguments_from_checkpoint(model_name_or_path, **overrides))

    def __init__(self, args: InferenceArguments):
        self.args = args
        dev = torch.device("cuda", args.local_rank if args.local_rank >= 0 else torch.cuda.current_device())
        torch.cuda.set_device(dev)
        # the tokenizer's thread pool is sized on its first parallel call: this rank's share of the host cores (an explicit RAYON_NUM_THREADS wins)
        os.environ.setdefault("RAYON_NUM_THREADS", str(host_threads_per_rank()))
        tok = load_tokenizer(args.model_name_or_path, lowercase=args.lowercase and args.edit_tokenizer_normalizers,
                             add_bos_num=args.add_bos_num if args.edit_tokenizer_post_processor else -1,
                             add_eos_num=args.add_eos_num if args.edit_tokenizer_post_processor else -1,
                             add_pad_token=args.add_pad_token, pad_token=args.pad_token, add_sep_token=args.add_sep_token,
                             sep_token=args.sep_token)
        enc = encoder_from_pretrained(args.model_name_or_path, max_positions=max(args.p_max_len, args.q_max_len, 64), device=dev, tokenizer=tok,
                                      pad_to_multiple_of=getattr(args, "pad_to_multiple_of", None))
        hybrid = args.model_type == "HybridModel"
        hm = LrxHybridModel(enc, normalize=args.normalize, dense_shrink_dim=args.dense_shrink_dim, pad_token_id=tok.pad_token_id,
                            encode_sparse=args.encode_sparse, sep_token_id=getattr(tok, "sep_token_id", None), add_sep_token=args.add_sep_token,
                            sparse_use_relu=args.sparse_use_relu, sparse_use_log_saturation=args.sparse_use_log_saturation,
                            sparse_top_k_psg=args.sparse_top_k_psg, sparse_top_p_psg=args.sparse_top_p_psg,
                            sparse_min_tokens_to_keep=args.sparse_min_tokens_to_keep,
                            hybrid_use_dense_vector=args.hybrid_use_dense_vector if hybrid else True,      # EncoderModel: the symmetric dense vector
                            hybrid_use_emb_vector=args.hybrid_use_emb_vector if hybrid else False,
                            noncontextual_query_embedding=args.noncontextual_query_embedding if hybrid else False,
                            pooling_strategy=args.pooling_strategy, hybrid_use_sparse_vector=args.hybrid_use_sparse_vector if hybrid else False,
                            hybrid_use_token_id_vector=args.hybrid_use_token_id_vector if hybrid else False,
                            sparse_top_k_qry=args.sparse_top_k_qry, sparse_top_p_qry=args.sparse_top_p_qry,
                            sparse_pool_from_original_input_ids_qry=args.sparse_pool_from_original_input_ids_qry if hybrid else False,
                            sparse_pool_from_original_input_ids_psg=args.sparse_pool_from_original_input_ids_psg if hybrid else False)
        super().__init__(model=hm, tokenizer=tok, q_max_len=args.q_max_len, p_max_len=args.p_max_len,
                         append_prompt_sep=args.append_prompt_sep, eval_batch_size_embedding_bag=args.eval_batch_size_embedding_bag,
                         token_id_vector_type=args.token_id_vector_type, noncontextual_prompt_prefix=args.noncontextual_prompt_prefix,
                         single_tensor_output=not hybrid)
        self.encoding_kwargs["anserini_vector_type"] = args.anserini_vector_type      # exact_search_torchrpc.py:100-101
        from . import rpc_shards
        rpc_shards.register_worker(self)              # the model remote calls of a driving rank will use (MODEL_REGISTRY of the reference)

    def empty_cache(self, rank=None):  # exact_search_torchrpc.py:330-335: here it only returns this process's cached allocator blocks
        torch.cuda.empty_cache()

    def stop_multi_process_pool(self):  # API parity with the reference (exact_search_torchrpc.py:103-120); nothing to stop
        return None
