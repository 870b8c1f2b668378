"""torch.ops.lrx.*: the PyTorch custom-op binding of liblrx (csrc/lrx_torch.cpp, TORCH_LIBRARY), SURVEY.md 8b last row.

    from lightretriever_amd import torch_ops          # loads liblrx_torch.so once; raises if it has not been built
    torch.ops.lrx.encode_packed(ids, cu_seqlens, max_seqlen, enc.handle, out, mrl_dim, normalize)
    D, I = torch.ops.lrx.flat_ip_topk(q, X, k)

Same kernels, same results as the ctypes binding (`_lib`): both call the C ABI of include/lrx.h; this one takes tensors, uses the
current HIP stream, allocates workspaces from PyTorch's caching allocator and raises RuntimeError (TORCH_CHECK) on liblrx errors.
There is no fallback: without liblrx_torch.so importing this module fails."""
import os

import torch

from . import _lib

HERE = os.path.dirname(os.path.abspath(__file__))
TORCH_LIB_PATH = os.path.join(HERE, "liblrx_torch.so")
OPS = ("encode_packed", "rmsnorm", "rope_qkv_gemm", "attn_varlen", "swiglu_gemm", "embedding_bag_mean", "flat_ip_topk", "flat_ip_topk_bounded",
       "flat_ip_topk_bounded_wire", "merge_topk_packed", "shard_commit_rows", "merge_topk")

_loaded = False


def load():
    global _loaded
    if not _loaded:
        if not os.path.exists(TORCH_LIB_PATH):
            raise _lib.LrxError(f"{TORCH_LIB_PATH} not found: build it with `python -m lightretriever_amd.build`; there is no fallback")
        _lib.lib()                                  # liblrx.so first (same file the registration layer links against)
        torch.ops.load_library(TORCH_LIB_PATH)
        _loaded = True
    return torch.ops.lrx


load()
