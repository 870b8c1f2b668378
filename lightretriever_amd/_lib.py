"""ctypes binding of liblrx.so (include/lrx.h).  There is NO fallback: if the HIP library is missing or fails to load
every entry point raises -- the product path never runs on a CPU/PyTorch substitute."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LRX_LIB_DEV_VARIANT") or os.path.join(HERE, "liblrx.so")   # env override: tools/ diagnostics only
LRX_PROF_CLASSES = 8
ABI_VERSION = 8   # LRX_ABI_VERSION of include/lrx.h
# lrx_flat_ip_search_bounded flags (LRX_SEARCH_FILTER_*): A/B runs and tests; the hits do not depend on them
SEARCH_FILTER_AUTO, SEARCH_FILTER_MATRIX, SEARCH_FILTER_SCORE_FREE, SEARCH_FILTER_SCORE_FREE_NO_GEMM = 0, 1, 2, 3
SEARCH_FUSED_ALWAYS, SEARCH_FUSED_NEVER = 4, 8
SEARCH_REFINE_ROWS_ALWAYS, SEARCH_REFINE_ROWS_NEVER = 16, 32   # exact rescoring grouped by row / per query (include/lrx.h)       # OR-ed on: the fused filter launch wherever eligible / never (default: a measured rule)
PROF_CLASS_NAMES = ["gemm_store", "gemm_resid", "gemm_swiglu", "attention", "rmsnorm", "rope", "other", "gemm_maxagg"]


class LrxError(RuntimeError):
    pass


class EncoderConfigC(C.Structure):
    _fields_ = [("vocab_size", C.c_int32), ("hidden_size", C.c_int32), ("num_layers", C.c_int32), ("num_q_heads", C.c_int32),
                ("num_kv_heads", C.c_int32), ("head_dim", C.c_int32), ("intermediate_size", C.c_int32), ("rms_eps", C.c_float),
                ("qkv_bias", C.c_int32), ("max_positions", C.c_int32), ("norm_folded", C.c_int32), ("precise_stream", C.c_int32)]


class LayerWeightsC(C.Structure):
    _fields_ = [("wqkv", C.c_void_p), ("bqkv", C.c_void_p), ("wo", C.c_void_p), ("wgu", C.c_void_p), ("wdown", C.c_void_p),
                ("ln1", C.c_void_p), ("ln2", C.c_void_p)]


class EncoderWeightsC(C.Structure):
    _fields_ = [("embed", C.c_void_p), ("final_norm", C.c_void_p), ("rope_cos", C.c_void_p), ("rope_sin", C.c_void_p),
                ("layers", C.POINTER(LayerWeightsC))]


# LRX_POOL_* of include/lrx.h: `--pooling_strategy` (finetune/dense_pooling.py:12-82)
POOLING = {"lasttoken": 0, "cls": 1, "mean": 2, "second_to_last": 3, "third_to_last": 4, "avg_first_last": 5, "avg_top2": 6}

_P, _I32, _I64, _F, _SZ = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_size_t

# name -> (restype, argtypes): exactly the entry points include/lrx.h declares
SIGNATURES = {
    "lrx_abi_version": (_I32, []),
    "lrx_last_error": (C.c_char_p, []),
    "lrx_encode_workspace_bytes": (_SZ, [C.POINTER(EncoderConfigC), _I32, _I32]),
    "lrx_encode_packed": (_I32, [C.POINTER(EncoderConfigC), C.POINTER(EncoderWeightsC), _P, _P, _I32, _I32, _I32, _P, _I64, _I32, _I32, _P, _SZ, _P]),
    "lrx_encode_packed_shard": (_I32, [C.POINTER(EncoderConfigC), C.POINTER(EncoderWeightsC), _P, _P, _I32, _I32, _I32, _P, _I64, _I32, _I32, _P, _I64,
                                       _P, _P, _SZ, _P]),
    "lrx_encode_packed_pooled": (_I32, [C.POINTER(EncoderConfigC), C.POINTER(EncoderWeightsC), _P, _P, _I32, _I32, _I32, _I32, _P, _I64, _I32, _I32, _P,
                                        _I64, _P, _P, _SZ, _P]),
    "lrx_pool_norm_mode": (_I32, [_P, _P, _P, _I32, _I32, _F, _I32, _P, _I64, _I32, _I32, _P, _I64, _P, _I32, _P]),
    "lrx_encode_hidden": (_I32, [C.POINTER(EncoderConfigC), C.POINTER(EncoderWeightsC), _P, _P, _I32, _I32, _I32, _P, _P, _SZ, _P]),
    "lrx_encode_prefixed_workspace_bytes": (_SZ, [C.POINTER(EncoderConfigC), _I32, _I32, _I32]),
    "lrx_encode_prefixed": (_I32, [C.POINTER(EncoderConfigC), C.POINTER(EncoderWeightsC), _P, _I32, _P, _I32, _I32, _P, _I64, _I32, _I32, _P, _SZ, _P]),
    "lrx_attn_prefix_suffix": (_I32, [_P, _P, _I32, _I32, _I32, _I32, _I32, _I32, _P, _P]),
    "lrx_uniform_layout": (_I32, [_P, _P, _I32, _I32, _I32, _P]),
    "lrx_encode_packed_sparse": (_I32, [C.POINTER(EncoderConfigC), C.POINTER(EncoderWeightsC), _P, _P, _P, _P, _P, _I32, _I32, _I32, _P, _I64, _I32, _I32,
                                        _P, _I64, _I32, _I32, _I32, _I32, _I32, _P, _SZ, _P]),
    "lrx_sparse_max_aggregate": (_I32, [_P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _P, _I64, _P, _P]),
    "lrx_sparsify": (_I32, [_P, _I32, _I32, _I64, _I32, _I32, _I32, _I32, _I32, _P]),
    "lrx_sparse_compact": (_I32, [_P, _I32, _I32, _I64, _I32, _I32, _P, _P, _P, _P]),
    "lrx_hit_contributions": (_I32, [_P, _P, _I32, _I32, _I64, _I32, C.c_double, C.c_double, _P, _I64, _P]),
    "lrx_hit_union": (_I32, [_P, _P, _I32, _I32, _I64, _P, _P, _P, _P]),
    "lrx_flat_ip_bounded_workspace_bytes": (_SZ, [_I64, _I32, _I32, _I32, _I32]),
    "lrx_flat_ip_search_bounded": (_I32, [_P, _I64, _I64, _I32, _P, _P, _P, _I32, _I32, _I64, _P, _P, _P, _SZ, _I32, _P]),
    "lrx_flat_ip_search_bounded_wire": (_I32, [_P, _I64, _I64, _I32, _P, _P, _P, _I32, _I32, _I64, _P, _P, _P, _P, _P, _SZ, _I32, _P]),
    "lrx_search_fallback_count": (_I64, [_I32]),
    "lrx_flat_ip_bounded_chunk_queries": (_I32, [_I64, _I32, _I32, _I32, _I32, _I32]),
    "lrx_flat_ip_bounded_list_counts": (_I32, [_P, _I64, _I32, _I32, _I32, _I32, _I32, _P, _P]),
    "lrx_shard_commit_rows": (_I32, [_P, _I64, _I64, _I32, _P, _I64, _P, _P]),
    "lrx_pool_norm_shard": (_I32, [_P, _P, _P, _I32, _I32, _F, _P, _I64, _I32, _I32, _P, _I64, _P, _I32, _P]),
    "lrx_gemm_bf16_nt_resid32": (_I32, [_P, _P, _P, _P, _P, _I32, _I32, _I32, _P, _P]),
    "lrx_embed_stream32": (_I32, [_P, _P, _I32, _I32, _I32, _P, _P, _P, _P, _F, _P]),
    "lrx_rmsnorm_f32": (_I32, [_P, _P, _P, _I32, _I32, _F, _P]),
    "lrx_gemm_bf16_nt_fused": (_I32, [_P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _P, _P, _P]),
    "lrx_gemm_qkv_rope_fused": (_I32, [_P, _P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _P, _P]),
    "lrx_row_rscale": (_I32, [_P, _I32, _I32, C.c_float, _P, _P]),
    "lrx_finalize_rscale": (_I32, [_P, _I32, _I32, _I32, C.c_float, _P, _P]),
    "lrx_set_profiling": (None, [_I32]),
    "lrx_get_profile": (_I32, [C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_int32)]),
    "lrx_embedding_gather": (_I32, [_P, _P, _I32, _I32, _I32, _P, _P]),
    "lrx_trace_marker": (_I32, [_I32, _P]),
    "lrx_probe_stream_read": (_I32, [_P, _SZ, _P, _I32, _P]),
    "lrx_device_error_count": (_I64, [_I32]),
    "lrx_device_saturation_count": (_I64, [_I32]),
    "lrx_rmsnorm": (_I32, [_P, _P, _P, _I32, _I32, _F, _P]),
    "lrx_gemm_bf16_nt": (_I32, [_P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _P]),
    "lrx_gemm_qkv_rope": (_I32, [_P, _P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _P]),
    "lrx_build_positions": (_I32, [_P, _I32, _I32, _P, _P]),
    "lrx_attn_varlen_causal": (_I32, [_P, _P, _I32, _I32, _I32, _I32, _I32, _I32, _P, _I32, _P]),
    "lrx_attn_items_bytes": (_SZ, [_I32, _I32, _I32, _I32, _I32, _I32, _I32]),
    "lrx_attn_build_items": (_I32, [_P, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _P, _SZ, _P]),
    "lrx_attn_varlen_causal_items": (_I32, [_P, _P, _P, _SZ, _I32, _I32, _I32, _I32, _I32, _I32, _P, _I32, _P]),
    "lrx_debug_attn_items_overflow": (_I32, [_P]),
    "lrx_gather_last_rows": (_I32, [_P, _P, _I32, _I32, _P, _P]),
    "lrx_probe_fused_timestamps": (_I32, [_P, _I32]),
    "lrx_scatter_last_rows": (_I32, [_P, _P, _I32, _I32, _P, _I64, _P]),
    "lrx_gemm_qkv_rope_slice": (_I32, [_P, _P, _P, _P, _P, _P, _P, _I32, _I32, _I32, _I32, _I32, _P, _I32, _I32, _P]),
    "lrx_pool_norm": (_I32, [_P, _P, _P, _I32, _I32, _F, _P, _I64, _I32, _I32, _P]),
    "lrx_embedding_bag_mean": (_I32, [_P, _I32, _I32, _P, _I64, _P, _I32, _I64, _P, _I64, _I32, _I32, _P]),
    "lrx_flat_ip_workspace_bytes": (_SZ, [_I64, _I32, _I32, _I32]),
    "lrx_flat_ip_search": (_I32, [_P, _I64, _I64, _I32, _P, _P, _I32, _I32, _I64, _P, _P, _P, _SZ, _P]),
    "lrx_flat_ip_score_ld": (_I64, [_I64]),
    "lrx_flat_ip_scores": (_I32, [_P, _I64, _I64, _I32, _P, _I32, _P, _P]),
    "lrx_merge_topk": (_I32, [_P, _P, _I32, _I32, _I32, _P, _P, _P]),
    "lrx_pack_topk": (_I32, [_P, _P, _P, _I64, _I64, _P, _P]),
    "lrx_merge_topk_packed": (_I32, [_P, _I32, _I32, _I32, _P, _P, _P]),
}

_lib = None


def lib():
    """Load liblrx.so once.  Raises LrxError (never falls back) when the extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise LrxError(f"{LIB_PATH} not found: build it with `python -m lightretriever_amd.build` "
                           "(or __graft_entry__.build()); there is no CPU fallback")
        try:
            l = C.CDLL(LIB_PATH)
        except OSError as e:  # pragma: no cover
            raise LrxError(f"failed to load {LIB_PATH}: {e}") from e
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype, fn.argtypes = res, args
        if l.lrx_abi_version() != ABI_VERSION:
            raise LrxError("liblrx.so ABI version mismatch")
        _lib = l
    return _lib


def check(rc: int):
    if rc != 0:
        raise LrxError(f"liblrx error {rc}: {lib().lrx_last_error().decode()}")


def ptr(t):
    """Device pointer of a torch tensor (or None -> NULL)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def current_stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_gpu():
    import torch
    if not torch.cuda.is_available():
        raise LrxError("lightretriever_amd needs a ROCm GPU (MI355X / gfx950); no CPU path exists")
