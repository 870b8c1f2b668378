"""`InferenceArguments` + `PytorchRPCExactSearchModel` under the reference's names (inference/arguments.py:19-157,
inference/exact_search_torchrpc.py:49-101), dense asymmetric configuration only.

No RPC: one process per GPU under torchrun; every rank constructs this object, encodes its own share and keeps the
embeddings in its HBM shard (lightretriever_amd.sharded).  The constructor signature, attribute names and the
encode_queries / encode_corpus / encode surface are the reference's."""
from __future__ import annotations

import os
from dataclasses import dataclass, field
from typing import Optional

import torch

from .loader import default_special_tokens, encoder_from_pretrained, load_tokenizer
from .modeling import LrxExactSearchModel, LrxHybridModel


@dataclass
class InferenceArguments:
    model_name_or_path: Optional[str] = None
    model_type: str = "HybridModel"
    inference_arch: str = "PytorchRPCExactSearchModel"
    batch_size: int = 64
    append_prompt_sep: bool = False
    q_max_len: int = 128
    p_max_len: int = 512
    bf16: bool = True
    fp16: bool = False
    seed: int = 42
    attn_implementation: str = "flash_attention_2"   # accepted for CLI compatibility; the HIP path has one implementation
    cumulative_seq: bool = True                         # packed varlen is the only layout here
    liger_kernel: bool = False
    # model args that change the dense-path numerics (finetune/arguments.py:75-335)
    pooling_strategy: str = "lasttoken"
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
    hybrid_use_dense_vector: bool = False
    hybrid_use_emb_vector: bool = True
    noncontextual_query_embedding: bool = True
    noncontextual_prompt_prefix: Optional[str] = None
    eval_batch_size_embedding_bag: int = 5000
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
        self.normalize = self.score_function == "cos_sim"       # finetune/arguments.py:312-317
        self.pad_token, self.sep_token = default_special_tokens(self.model_name_or_path, self.pad_token, self.sep_token)
        if self.pooling_strategy != "lasttoken":
            raise NotImplementedError("the MI355X path implements the shipped 'lasttoken' pooling only")
        if self.fp16:
            raise NotImplementedError("bf16 is the compute type of the HIP encoder")


class PytorchRPCExactSearchModel(LrxExactSearchModel):
    """Drop-in for eval/eval_utils.py:179 `PytorchRPCExactSearchModel(args)`."""

    def __init__(self, args: InferenceArguments):
        self.args = args
        dev = torch.device("cuda", args.local_rank if args.local_rank >= 0 else torch.cuda.current_device())
        torch.cuda.set_device(dev)
        tok = load_tokenizer(args.model_name_or_path, lowercase=args.lowercase and args.edit_tokenizer_normalizers,
                             add_bos_num=args.add_bos_num if args.edit_tokenizer_post_processor else -1,
                             add_eos_num=args.add_eos_num if args.edit_tokenizer_post_processor else -1,
                             add_pad_token=args.add_pad_token, pad_token=args.pad_token, add_sep_token=args.add_sep_token,
                             sep_token=args.sep_token)
        enc = encoder_from_pretrained(args.model_name_or_path, max_positions=max(args.p_max_len, args.q_max_len, 64), device=dev)
        hm = LrxHybridModel(enc, normalize=args.normalize, dense_shrink_dim=args.dense_shrink_dim, pad_token_id=tok.pad_token_id)
        super().__init__(model=hm, tokenizer=tok, q_max_len=args.q_max_len, p_max_len=args.p_max_len,
                         append_prompt_sep=args.append_prompt_sep, eval_batch_size_embedding_bag=args.eval_batch_size_embedding_bag)

    def stop_multi_process_pool(self):  # API parity with the reference (exact_search_torchrpc.py:103-120); nothing to stop
        return None
