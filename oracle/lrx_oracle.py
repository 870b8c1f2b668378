"""CPU oracle for the LightRetriever dense corpus-embedding + flat-IP search path.

TEST INFRASTRUCTURE ONLY.  This file is a plain-numpy restatement of the arithmetic
the reference performs on its dense asymmetric eval path.  It may be imported only by
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` --
and there only as the checker.  The product path (``lightretriever_amd``) never imports
it and has no CPU fallback.

Parity pin status
-----------------
* Encoder forward / pooling / normalise / EmbeddingBag / packing: PINNED by golden
  vectors under ``tests/golden/`` that were produced in the build container by running
  the reference's own Python (``/root/reference/src/lightretriever``) on top of the
  third-party stack it delegates to (HF ``transformers`` 5.15.0, ``torch`` 2.10) --
  see ``tests/golden/gen_goldens.py``.
* Flat inner-product search (Faiss ``IndexFlatIP``): **parity unpinned** against Faiss
  itself -- ``faiss`` (pyproject.toml:6, ``faiss>=1.7.4``) is neither vendored in the
  reference nor installed here, and the reference has no tests.  ``flat_ip_topk`` follows
  the published semantics (exact fp32 inner product, descending top-k, ids = insertion
  order, ``-FLT_MAX`` / ``-1`` padding when k > ntotal = ``CMin<float>::neutral()``); the
  order of equal scores (lower row first) is THIS BUILD'S contract, not Faiss behaviour
  (Faiss leaves it unspecified; its strict heap comparison keeps the earlier row at the
  k-th place, which the rule agrees with).  It is pinned only against ``torch.matmul`` +
  ``torch.topk`` goldens (``tests/golden/search.npz``).
* Everything the reference's OWN code does with the index's answers -- ``FaissIndex.build /
  search`` id mapping (retriever/faiss_index.py:27-58), ``retrieve_with_emb``
  (retriever/faiss_search.py:143-173), the corpus sort + chunk loop + ``_add_to_heap`` /
  ``_parse_heap_results`` merge with its (score, pid) tuple order and ``ignore_identical_ids``
  (retriever/hybrid_search.py:182-205, :234-403; retriever/faiss_search.py:176-291): PINNED by
  ``tests/golden/search_ref.json``, produced by running those reference functions
  (``tests/golden/gen_search_goldens.py``; a numpy stand-in supplies the ``faiss`` module
  they import, which is why Faiss itself stays unpinned).

* Sparse document vectors (row N2: prompt/first/last mask, LM-head max aggregation, relu/log1p, top-k / top-p, quantised
  JSON): PINNED by ``tests/golden/sparse.npz`` / ``sparse_json.json`` (``tests/golden/gen_sparse_goldens.py``, the real
  reference functions).  The JSON converter is pinned against the reference's own torch variant
  (``convert_sparse_reps_to_json_pt``); its Rust crate ``sparse_emb_util`` is not in the reference tree -> **unpinned**.

* Pooling strategies other than 'lasttoken' (cls / mean / second_to_last / third_to_last / avg_first_last / avg_top2): PINNED by ``tests/golden/pooling.npz``
  (``gen_pooling_goldens.py`` runs ``finetune/dense_pooling.pooling`` and ``HybridModel.encode_passage / encode_query`` with each strategy).
* LM-head sparse QUERY vectors (``hybrid_use_sparse_vector``) and their pseudo text: PINNED by ``tests/golden/sparse_query.npz`` /
  ``sparse_query_text.json`` (``gen_sparse_query_goldens.py``: ``HybridModel.encode_query`` and ``convert_sparse_reps_to_pseudo_text_pt``).
* Sparse vectors restricted to the sequence's own tokens (``sparse_pool_from_original_input_ids_qry / _psg``): PINNED by
  ``tests/golden/sparse_pool_ids.npz`` (``gen_sparse_pool_ids_goldens.py``: ``encode_passage`` / ``encode_query`` with the flags on).

* Hit-list fusion (row N3: RRF, min-max linear): PINNED by ``tests/golden/fusion.json`` (``gen_fusion_goldens.py`` runs
  ``retriever/score_fuse_utils.py``), exact float64 equality.

Every function cites the reference ``file:line`` (relative to /root/reference) or the
third-party source it restates.
"""
from __future__ import annotations

import heapq
import math
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

# --------------------------------------------------------------------------------------
# bf16 helpers (round-to-nearest-even on the fp32 bit pattern, NaN preserved)
# --------------------------------------------------------------------------------------


def round_bf16(x: np.ndarray) -> np.ndarray:
    """Round fp32 -> bf16 -> fp32 (RNE), the rounding torch applies to every bf16 op result."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    u = x.view(np.uint32)
    rounded = ((u.astype(np.uint64) + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)
    out = rounded.view(np.float32).copy()
    nan = np.isnan(x)
    if nan.any():
        out[nan] = np.nan
    return out.reshape(x.shape)


def to_bf16_bits(x: np.ndarray) -> np.ndarray:
    """fp32 array -> uint16 bf16 bit patterns (RNE)."""
    return (round_bf16(x).view(np.uint32) >> 16).astype(np.uint16)


def from_bf16_bits(b: np.ndarray) -> np.ndarray:
    return (b.astype(np.uint32) << 16).view(np.float32)


# --------------------------------------------------------------------------------------
# Config
# --------------------------------------------------------------------------------------


@dataclass
class EncoderConfig:
    """Subset of the HF Llama/Qwen2 ``config.json`` that decides the dense-path arithmetic."""

    vocab_size: int
    hidden_size: int
    num_layers: int
    num_q_heads: int
    num_kv_heads: int
    head_dim: int
    intermediate_size: int
    rms_eps: float = 1e-5
    rope_theta: float = 500000.0
    rope_type: str = "default"  # "default" | "llama3"
    rope_factor: float = 32.0
    rope_low_freq_factor: float = 1.0
    rope_high_freq_factor: float = 4.0
    rope_original_max_position: int = 8192
    qkv_bias: bool = False  # Qwen2.5: True
    max_positions: int = 512

    @staticmethod
    def llama32_1b(max_positions: int = 512) -> "EncoderConfig":
        return EncoderConfig(128256, 2048, 16, 32, 8, 64, 8192, 1e-5, 500000.0, "llama3", 32.0, 1.0, 4.0, 8192,
                             False, max_positions)

    @staticmethod
    def llama31_8b(max_positions: int = 512) -> "EncoderConfig":
        return EncoderConfig(128256, 4096, 32, 32, 8, 128, 14336, 1e-5, 500000.0, "llama3", 8.0, 1.0, 4.0, 8192,
                             False, max_positions)


# --------------------------------------------------------------------------------------
# RoPE  (transformers/modeling_rope_utils.py: _compute_default_rope_parameters,
#        _compute_llama3_parameters; models/llama/modeling_llama.py LlamaRotaryEmbedding.forward,
#        rotate_half / apply_rotary_pos_emb)
# --------------------------------------------------------------------------------------


def rope_inv_freq(cfg: EncoderConfig) -> np.ndarray:
    d = cfg.head_dim
    inv_freq = 1.0 / (np.float32(cfg.rope_theta) ** (np.arange(0, d, 2, dtype=np.int64).astype(np.float32) / np.float32(d)))
    inv_freq = inv_freq.astype(np.float32)
    if cfg.rope_type == "default":
        return inv_freq
    if cfg.rope_type != "llama3":
        raise NotImplementedError(cfg.rope_type)
    factor = cfg.rope_factor
    low, high = cfg.rope_low_freq_factor, cfg.rope_high_freq_factor
    old = cfg.rope_original_max_position
    low_wl = old / low
    high_wl = old / high
    wavelen = (2 * math.pi / inv_freq).astype(np.float32)
    inv_llama = np.where(wavelen > low_wl, inv_freq / np.float32(factor), inv_freq).astype(np.float32)
    smooth = ((old / wavelen - low) / (high - low)).astype(np.float32)
    smoothed = ((1 - smooth) * inv_llama / np.float32(factor) + smooth * inv_llama).astype(np.float32)
    is_medium = (~(wavelen < high_wl)) & (~(wavelen > low_wl))
    return np.where(is_medium, smoothed, inv_llama).astype(np.float32)


def rope_table(cfg: EncoderConfig, n_pos: Optional[int] = None) -> tuple[np.ndarray, np.ndarray]:
    """cos/sin [n_pos, head_dim/2] in fp32 (HF duplicates the halves: emb = cat(freqs, freqs))."""
    n_pos = n_pos or cfg.max_positions
    inv = rope_inv_freq(cfg)
    pos = np.arange(n_pos, dtype=np.float32)
    freqs = (pos[:, None] * inv[None, :]).astype(np.float32)
    return np.cos(freqs).astype(np.float32), np.sin(freqs).astype(np.float32)


def apply_rope(x: np.ndarray, cos: np.ndarray, sin: np.ndarray) -> np.ndarray:
    """x [T, heads, d]; cos/sin [T, d/2].  q*cos + rotate_half(q)*sin with half-split pairing."""
    d2 = x.shape[-1] // 2
    x1, x2 = x[..., :d2], x[..., d2:]
    c, s = cos[:, None, :], sin[:, None, :]
    return np.concatenate([x1 * c - x2 * s, x2 * c + x1 * s], axis=-1).astype(np.float32)


# --------------------------------------------------------------------------------------
# Encoder forward (HF LlamaModel.forward, modeling_llama.py:367-418; decoder layer :284-325;
# LlamaRMSNorm :53-67; LlamaMLP :163-176; eager attention :179-281).  Reached from
# finetune/modeling_hybrid.py:248-260 (lm_p_base_unwrap(**forward_kwargs)).
# --------------------------------------------------------------------------------------


def rmsnorm(x: np.ndarray, w: np.ndarray, eps: float, bf16: bool) -> np.ndarray:
    x = x.astype(np.float32)
    var = np.mean(x * x, axis=-1, keepdims=True, dtype=np.float32)
    y = x * (1.0 / np.sqrt(var + np.float32(eps))).astype(np.float32)
    if bf16:
        return round_bf16(w * round_bf16(y))  # weight * x.to(in_dtype), each op rounds
    return (w * y).astype(np.float32)


def _mm(a: np.ndarray, w: np.ndarray, bf16: bool) -> np.ndarray:
    """a [T, K] @ w[N, K]^T (nn.Linear layout), fp32 accumulate."""
    y = a.astype(np.float32) @ w.astype(np.float32).T
    return round_bf16(y) if bf16 else y.astype(np.float32)


def _silu(x: np.ndarray) -> np.ndarray:
    return (x / (1.0 + np.exp(-x))).astype(np.float32)


def weight_names(cfg: EncoderConfig) -> list[str]:
    names = ["embed_tokens.weight", "norm.weight"]
    for i in range(cfg.num_layers):
        p = f"layers.{i}."
        names += [p + "input_layernorm.weight", p + "post_attention_layernorm.weight",
                  p + "self_attn.q_proj.weight", p + "self_attn.k_proj.weight", p + "self_attn.v_proj.weight",
                  p + "self_attn.o_proj.weight", p + "mlp.gate_proj.weight", p + "mlp.up_proj.weight",
                  p + "mlp.down_proj.weight"]
        if cfg.qkv_bias:
            names += [p + "self_attn.q_proj.bias", p + "self_attn.k_proj.bias", p + "self_attn.v_proj.bias"]
    return names


def random_weights(cfg: EncoderConfig, seed: int = 0, std: float = 0.02, bf16: bool = True, norm_std: float = 0.1,
                   bias_std: float = 0.05) -> dict[str, np.ndarray]:
    """N(0, std) matrices, norm weights 1 + N(0, norm_std), biases N(0, bias_std); pre-rounded to bf16-representable
    fp32 when ``bf16``.  numpy Generator(PCG64) stream -> the same values on every machine with this numpy."""
    rng = np.random.default_rng(seed)
    H, d = cfg.hidden_size, cfg.head_dim
    shapes = {"embed_tokens.weight": (cfg.vocab_size, H), "norm.weight": (H,)}
    for i in range(cfg.num_layers):
        p = f"layers.{i}."
        shapes.update({
            p + "input_layernorm.weight": (H,), p + "post_attention_layernorm.weight": (H,),
            p + "self_attn.q_proj.weight": (cfg.num_q_heads * d, H), p + "self_attn.k_proj.weight": (cfg.num_kv_heads * d, H),
            p + "self_attn.v_proj.weight": (cfg.num_kv_heads * d, H), p + "self_attn.o_proj.weight": (H, cfg.num_q_heads * d),
            p + "mlp.gate_proj.weight": (cfg.intermediate_size, H), p + "mlp.up_proj.weight": (cfg.intermediate_size, H),
            p + "mlp.down_proj.weight": (H, cfg.intermediate_size)})
        if cfg.qkv_bias:
            shapes.update({p + "self_attn.q_proj.bias": (cfg.num_q_heads * d,), p + "self_attn.k_proj.bias": (cfg.num_kv_heads * d,),
                           p + "self_attn.v_proj.bias": (cfg.num_kv_heads * d,)})
    out = {}
    for name in weight_names(cfg):
        w = rng.standard_normal(shapes[name], dtype=np.float32)
        if name.endswith("norm.weight"):
            w = 1.0 + w * np.float32(norm_std)
        elif name.endswith(".bias"):
            w = w * np.float32(bias_std)
        else:
            w = w * np.float32(std)
        out[name] = round_bf16(w) if bf16 else w.astype(np.float32)
    return out


def encoder_forward_packed(cfg: EncoderConfig, w: dict[str, np.ndarray], ids: np.ndarray, cu_seqlens: np.ndarray,
                           bf16: bool = False, return_layers: bool = False, final_norm: bool = True):
    """Packed-varlen causal forward.  ids [T] int, cu_seqlens [B+1]; position restarts at 0 per
    sequence (utils/nested_input.py:15-39 builds exactly these position ids).  Returns
    last_hidden_state [T, H] fp32.  ``bf16=True`` rounds to bf16 wherever a bf16 HF model rounds."""
    rnd = round_bf16 if bf16 else (lambda a: np.asarray(a, dtype=np.float32))
    T = int(ids.shape[0])
    H, d, nq, nkv = cfg.hidden_size, cfg.head_dim, cfg.num_q_heads, cfg.num_kv_heads
    grp = nq // nkv
    x = w["embed_tokens.weight"][ids].astype(np.float32)
    pos = np.zeros(T, dtype=np.int64)
    for b in range(len(cu_seqlens) - 1):
        s, e = int(cu_seqlens[b]), int(cu_seqlens[b + 1])
        pos[s:e] = np.arange(e - s)
    cos_t, sin_t = rope_table(cfg, int(pos.max()) + 1 if T else 1)
    cos, sin = rnd(cos_t[pos]), rnd(sin_t[pos])  # HF casts cos/sin to the activation dtype
    scale = np.float32(d ** -0.5)
    layers = []
    for i in range(cfg.num_layers):
        p = f"layers.{i}."
        h = rmsnorm(x, w[p + "input_layernorm.weight"], cfg.rms_eps, bf16)
        q = _mm(h, w[p + "self_attn.q_proj.weight"], False)
        k = _mm(h, w[p + "self_attn.k_proj.weight"], False)
        v = _mm(h, w[p + "self_attn.v_proj.weight"], False)
        if cfg.qkv_bias:
            q = q + w[p + "self_attn.q_proj.bias"]
            k = k + w[p + "self_attn.k_proj.bias"]
            v = v + w[p + "self_attn.v_proj.bias"]
        q, k, v = rnd(q).reshape(T, nq, d), rnd(k).reshape(T, nkv, d), rnd(v).reshape(T, nkv, d)
        q, k = rnd(apply_rope(q, cos, sin)), rnd(apply_rope(k, cos, sin))
        attn = np.zeros((T, nq, d), dtype=np.float32)
        for b in range(len(cu_seqlens) - 1):
            s, e = int(cu_seqlens[b]), int(cu_seqlens[b + 1])
            L = e - s
            if L == 0:
                continue
            causal = np.tril(np.ones((L, L), dtype=bool))
            for hq in range(nq):
                hk = hq // grp
                sc = (q[s:e, hq] @ k[s:e, hk].T) * scale
                sc = np.where(causal, sc, -np.inf).astype(np.float32)
                sc = sc - sc.max(axis=-1, keepdims=True)
                pexp = np.exp(sc).astype(np.float32)
                pr = pexp / pexp.sum(axis=-1, keepdims=True, dtype=np.float32)
                attn[s:e, hq] = rnd(pr) @ v[s:e, hk]
        attn = rnd(attn).reshape(T, nq * d)
        x = rnd(x + _mm(attn, w[p + "self_attn.o_proj.weight"], bf16))
        h = rmsnorm(x, w[p + "post_attention_layernorm.weight"], cfg.rms_eps, bf16)
        g = _mm(h, w[p + "mlp.gate_proj.weight"], bf16)
        u = _mm(h, w[p + "mlp.up_proj.weight"], bf16)
        act = rnd(rnd(_silu(g)) * u)
        x = rnd(x + _mm(act, w[p + "mlp.down_proj.weight"], bf16))
        if return_layers:
            layers.append(x.copy())
    out = rmsnorm(x, w["norm.weight"], cfg.rms_eps, bf16) if final_norm else x
    return (out, layers) if return_layers else out


# --------------------------------------------------------------------------------------
# Packing (utils/nested_input.py:15-39 unpad_to_seqlen_dim) -- integer work, bit-exact
# --------------------------------------------------------------------------------------


def pack_padded(input_ids: np.ndarray, attention_mask: np.ndarray):
    """[B,S] right- (or left-) padded ids+mask -> (ids_nested [T], position_ids [T], indices [T], cu_seqlens [B+1], max_len)."""
    mask = attention_mask.astype(bool)
    seqlens = mask.sum(axis=1).astype(np.int32)
    indices = np.flatnonzero(mask.reshape(-1)).astype(np.int64)
    ids_nested = input_ids.reshape(-1)[indices]
    cu = np.zeros(len(seqlens) + 1, dtype=np.int32)
    cu[1:] = np.cumsum(seqlens)
    position_ids = np.concatenate([np.arange(n, dtype=np.int64) for n in seqlens]) if len(seqlens) else np.zeros(0, np.int64)
    return ids_nested, position_ids, indices, cu, int(seqlens.max()) if len(seqlens) else 0


# --------------------------------------------------------------------------------------
# Pooling + normalise (finetune/dense_pooling.py:48-55; finetune/modeling_hybrid.py:272-278)
# --------------------------------------------------------------------------------------


def lasttoken_pool_packed(last_hidden: np.ndarray, cu_seqlens: np.ndarray) -> np.ndarray:
    """In the packed layout 'lasttoken' is h[cu_seqlens[1:] - 1] (both branches of dense_pooling.py:48-55
    select the last non-pad token of each right-padded row)."""
    return last_hidden[np.asarray(cu_seqlens[1:], dtype=np.int64) - 1]


def lasttoken_pool_padded(last_hidden: np.ndarray, attention_mask: np.ndarray) -> np.ndarray:
    """Literal restatement of dense_pooling.py:48-55 on [B,S,H]."""
    B = last_hidden.shape[0]
    left_padding = attention_mask[:, -1].sum() == B
    if left_padding:
        return last_hidden[:, -1]
    idx = attention_mask.sum(axis=1) - 1
    return last_hidden[np.arange(B), idx]


POOLING_STRATEGIES = ("lasttoken", "cls", "mean", "second_to_last", "third_to_last", "avg_first_last", "avg_top2")


def pool_packed(last_hidden: np.ndarray, cu_seqlens: np.ndarray, strategy: str = "lasttoken", other_hidden: Optional[np.ndarray] = None) -> np.ndarray:
    """pooling() of finetune/dense_pooling.py:12-82 in the packed layout [T, H] (right-padded rows in the reference; both branches of the
    x-to-last strategies select the same token there): 'cls' = first token (:32-33), 'mean' = sum over the sequence's tokens / its length
    (:35-36; fp32, tokens added in order), 'lasttoken' / 'second_to_last' / 'third_to_last' = token len-1 / len-2 / len-3 (:48-79; the
    reference asserts len >= 2 / 3).  'avg_first_last' / 'avg_top2' (:38-46): the mean over the tokens of (other + last) / 2 with
    `other_hidden` = hidden_states[0] (the embedding rows) / hidden_states[-2] (the stream entering the final layer)."""
    cu = np.asarray(cu_seqlens, dtype=np.int64)
    if strategy in ("avg_first_last", "avg_top2"):
        assert other_hidden is not None and other_hidden.shape == last_hidden.shape
        half = (other_hidden.astype(np.float32) + last_hidden.astype(np.float32)) / np.float32(2.0)
        return pool_packed(half, cu, "mean")
    if strategy == "lasttoken":
        return lasttoken_pool_packed(last_hidden, cu_seqlens)
    if strategy == "cls":
        return last_hidden[cu[:-1]]
    if strategy in ("second_to_last", "third_to_last"):
        back = 2 if strategy == "second_to_last" else 3
        assert np.all(cu[1:] - cu[:-1] >= back), f"{strategy}: a sequence has fewer than {back} tokens"
        return last_hidden[cu[1:] - back]
    if strategy == "mean":
        out = np.zeros((len(cu) - 1, last_hidden.shape[1]), dtype=np.float32)
        for b in range(len(cu) - 1):
            acc = np.zeros(last_hidden.shape[1], dtype=np.float32)
            for t in range(cu[b], cu[b + 1]):
                acc += last_hidden[t].astype(np.float32)
            out[b] = acc / np.float32(cu[b + 1] - cu[b])
        return out
    raise NotImplementedError(strategy)


def pool_padded(last_hidden: np.ndarray, attention_mask: np.ndarray, strategy: str, hidden_states=None) -> np.ndarray:
    """The same on the reference's padded layout [B, S, H] + mask (literal restatement, for the goldens of pooling() itself);
    hidden_states: the tuple the two-layer strategies index ([0] / [-2], and [-1] as the last state, :38-46)."""
    B = last_hidden.shape[0]
    m = attention_mask.astype(np.int64)
    if strategy in ("avg_first_last", "avg_top2"):
        other = hidden_states[0] if strategy == "avg_first_last" else hidden_states[-2]
        return (((other + hidden_states[-1]) / np.float32(2.0) * m[..., None]).sum(1) / m.sum(-1)[..., None]).astype(np.float32)
    if strategy == "cls":
        return last_hidden[:, 0]
    if strategy == "mean":
        return ((last_hidden * m[..., None]).sum(1) / m.sum(-1)[..., None]).astype(np.float32)
    back = {"lasttoken": 1, "second_to_last": 2, "third_to_last": 3}[strategy]
    if m[:, -1].sum() == B:
        return last_hidden[:, -back]
    return last_hidden[np.arange(B), m.sum(axis=1) - back]


def l2_normalize(x: np.ndarray, eps: float = 1e-12) -> np.ndarray:
    """torch.nn.functional.normalize(p=2, dim=-1): x / max(||x||, eps)."""
    x = x.astype(np.float32)
    n = np.sqrt(np.sum(x * x, axis=-1, keepdims=True, dtype=np.float32))
    return (x / np.maximum(n, np.float32(eps))).astype(np.float32)


def encode_passage(cfg: EncoderConfig, w, ids, cu_seqlens, dense_shrink_dim: Optional[int] = None,
                   normalize: bool = True, bf16: bool = False, pooling: str = "lasttoken") -> np.ndarray:
    """finetune/modeling_hybrid.py:205-278 dense branch: forward -> pooling(strategy; 'lasttoken' in the released models) -> MRL slice ->
    F.normalize."""
    other = None
    if pooling in ("avg_first_last", "avg_top2"):
        # HF's hidden_states tuple (modeling_hybrid.py:257 asks for it): [0] = the embedding rows, [i] = the stream after layer i, and the
        # LAST entry is the final-norm output (= last_hidden_state), so [-2] is the stream after layer L - 1, unnormed (L = 1: the embeddings)
        h, layers = encoder_forward_packed(cfg, w, ids, cu_seqlens, bf16=bf16, return_layers=True)
        emb = w["embed_tokens.weight"][np.asarray(ids)].astype(np.float32)
        other = emb if pooling == "avg_first_last" or cfg.num_layers == 1 else layers[-2]
    else:
        h = encoder_forward_packed(cfg, w, ids, cu_seqlens, bf16=bf16)
    p = pool_packed(h, cu_seqlens, pooling, other)
    if dense_shrink_dim:
        p = p[..., :dense_shrink_dim]
    return l2_normalize(p) if normalize else p.astype(np.float32)


# --------------------------------------------------------------------------------------
# Query side: EmbeddingBag(mode='mean', padding_idx) (finetune/nonctx_emb_utils.py:197-219, :310-313;
# finetune/modeling_hybrid.py:472-490)
# --------------------------------------------------------------------------------------


def nonctx_offsets(lengths: list[int]) -> np.ndarray:
    """offsets = cumsum([0] + len[:-1])   (nonctx_emb_utils.py:217)."""
    return np.cumsum([0] + list(lengths[:-1])).astype(np.int64)


def embedding_bag_mean(table: np.ndarray, ids: np.ndarray, offsets: np.ndarray, padding_idx: Optional[int] = None) -> np.ndarray:
    """torch.nn.EmbeddingBag(mode='mean', padding_idx=p): mean over the non-padding ids of each bag; empty bag -> 0."""
    Q = len(offsets)
    out = np.zeros((Q, table.shape[1]), dtype=np.float32)
    ends = list(offsets[1:]) + [len(ids)]
    for i in range(Q):
        bag = ids[int(offsets[i]):int(ends[i])]
        if padding_idx is not None:
            bag = bag[bag != padding_idx]
        if len(bag):
            acc = np.zeros(table.shape[1], dtype=np.float32)
            for t in bag:  # sequential fp32 accumulation like the torch CPU kernel
                acc += table[int(t)].astype(np.float32)
            out[i] = acc / np.float32(len(bag))
    return out


def encode_query_emb(table, ids, offsets, padding_idx=None, dense_shrink_dim=None, normalize=True):
    e = embedding_bag_mean(table, ids, offsets, padding_idx)
    if dense_shrink_dim:
        e = e[..., :dense_shrink_dim]
    return l2_normalize(e) if normalize else e


def construct_embedding_bag(cfg: EncoderConfig, w, bos_id: Optional[int], eos_id: int, prompt_ids: list[int],
                            vocab_len: Optional[int] = None, bf16: bool = False) -> np.ndarray:
    """finetune/nonctx_emb_utils.py:239-313: for every tok in [0, len(tokenizer)) run the encoder on
    [bos] + prompt + [tok] + [eos] and keep last_hidden_state[:, -1] un-normalised, fp32."""
    V = vocab_len or cfg.vocab_size
    prefix = ([bos_id] if bos_id is not None else []) + list(prompt_ids)
    L = len(prefix) + 2
    ids = np.empty((V, L), dtype=np.int64)
    ids[:, :len(prefix)] = np.asarray(prefix, dtype=np.int64)[None, :] if prefix else 0
    ids[:, -2] = np.arange(V)
    ids[:, -1] = eos_id
    cu = (np.arange(V + 1) * L).astype(np.int32)
    h = encoder_forward_packed(cfg, w, ids.reshape(-1), cu, bf16=bf16)
    return h[cu[1:] - 1].astype(np.float32)


# --------------------------------------------------------------------------------------
# Flat inner-product index (retriever/faiss_index.py:20-73; faiss.IndexFlatIP published semantics)
# --------------------------------------------------------------------------------------


def flat_ip_topk(q: np.ndarray, X: np.ndarray, k: int) -> tuple[np.ndarray, np.ndarray]:
    """scores = q @ X.T (fp32), per-query top-k descending.  Ties broken by LOWER row id (deterministic rule
    this build defines; Faiss leaves tie order unspecified).  If k > N the tail is (-inf... faiss uses
    -3.4e38, id -1): we return score = -FLT_MAX, id = -1 like faiss' heap initialisation."""
    q = np.ascontiguousarray(q, dtype=np.float32)
    X = np.ascontiguousarray(X, dtype=np.float32)
    Q, N = q.shape[0], X.shape[0]
    S = (q @ X.T).astype(np.float32) if N else np.zeros((Q, 0), np.float32)
    D = np.full((Q, k), -np.finfo(np.float32).max, dtype=np.float32)
    I = np.full((Q, k), -1, dtype=np.int64)
    kk = min(k, N)
    if kk:
        order = np.lexsort((np.broadcast_to(np.arange(N), S.shape), -S), axis=-1)[:, :kk]
        D[:, :kk] = np.take_along_axis(S, order, axis=1)
        I[:, :kk] = order
    return D, I


def merge_topk(D_parts: list[np.ndarray], I_parts: list[np.ndarray], k: int) -> tuple[np.ndarray, np.ndarray]:
    """Merge per-shard (score, global row) lists -> top-k by (score desc, row asc); id -1 entries sink."""
    D = np.concatenate(D_parts, axis=1)
    I = np.concatenate(I_parts, axis=1)
    Q = D.shape[0]
    outD = np.full((Q, k), -np.finfo(np.float32).max, dtype=np.float32)
    outI = np.full((Q, k), -1, dtype=np.int64)
    for qi in range(Q):
        valid = I[qi] >= 0
        d, i = D[qi][valid], I[qi][valid]
        order = np.lexsort((i, -d))[:k]
        outD[qi, :len(order)] = d[order]
        outI[qi, :len(order)] = i[order]
    return outD, outI


class FlatIPIndexOracle:
    """faiss.IndexFlatIP surface used by retriever/faiss_index.py: add / search / reset / ntotal."""

    def __init__(self, d: int):
        self.d = d
        self._chunks: list[np.ndarray] = []
        self.ntotal = 0

    def add(self, x: np.ndarray):
        x = np.ascontiguousarray(x, dtype=np.float32)
        assert x.ndim == 2 and x.shape[1] == self.d
        self._chunks.append(x)
        self.ntotal += x.shape[0]

    def search(self, q: np.ndarray, k: int):
        X = np.concatenate(self._chunks, axis=0) if self._chunks else np.zeros((0, self.d), np.float32)
        return flat_ip_topk(q, X, k)

    def reset(self):
        self._chunks, self.ntotal = [], 0


# --------------------------------------------------------------------------------------
# search(): chunk loop + heap merge (retriever/hybrid_search.py:182-205, :273-358;
# retriever/faiss_search.py:143-173, :228-291)
# --------------------------------------------------------------------------------------


def sort_corpus_ids_longest_first(corpus: dict) -> list[str]:
    """hybrid_search.py:273-276 / faiss_search.py:214-216: sorted(..., key=len(text), reverse=True) (stable)."""
    return sorted(corpus, key=lambda k: len(corpus[k].get("text", "")) if isinstance(corpus[k], dict) else len(corpus[k]),
                  reverse=True)


def add_to_heap(sub_results: dict, heaps: dict, top_k: int, ignore_identical_ids: bool):
    """hybrid_search.py:182-205: per-query min-heap of (score, pid) capped at top_k.  The heap keeps the top_k LARGEST
    tuples pushed so far under Python's tuple order, i.e. (score, then pid) -- at equal scores the larger pid survives."""
    for qid, pid_to_score in sub_results.items():
        for pid, score in pid_to_score.items():
            if ignore_identical_ids and qid == pid:
                continue
            if qid not in heaps:
                heaps[qid] = []
            if len(heaps[qid]) < top_k:
                heapq.heappush(heaps[qid], (score, pid))
            else:
                heapq.heappushpop(heaps[qid], (score, pid))


def faiss_index_search(q: np.ndarray, X: np.ndarray, k: int, passage_ids: Optional[np.ndarray] = None,
                       reference_padding: bool = False) -> tuple[np.ndarray, np.ndarray]:
    """FaissIndex.search (retriever/faiss_index.py:27-40): index.search, then rows -> `_passage_ids[rows]`.
    reference_padding=True restates the reference literally: the numpy lookup turns the -1 padding id of a k > ntotal
    search into `_passage_ids[-1]`, the LAST passage.  False (the product's contract): padding stays -1."""
    D, I = flat_ip_topk(q, X, k)
    if passage_ids is not None:
        pid = np.asarray(passage_ids, dtype=np.int64)
        if reference_padding:
            I = pid[I.reshape(-1)].reshape(I.shape) if len(pid) else I
        else:
            I = np.where(I >= 0, pid[np.clip(I, 0, None)], -1) if len(pid) else I
    return D, I


def retrieve_with_emb(q: np.ndarray, query_ids: list, X: np.ndarray, corpus_ids: list, top_k: int,
                      reference_padding: bool = False) -> dict:
    """FlatIPFaissSearch.index + retrieve_with_emb (retriever/faiss_search.py:490-504, :143-173): rows -> ids through
    `rev_mapping`, `float(score)`, `dict(zip(doc_ids, scores))` per query.  With reference_padding the padding entries of a
    short index arrive as the last document with score -FLT_MAX and, being later in the zip, overwrite its real score."""
    D, I = faiss_index_search(q, X, top_k, passage_ids=np.arange(len(corpus_ids)), reference_padding=reference_padding)
    out = {}
    for qi, qid in enumerate(query_ids):
        pairs = [(corpus_ids[int(r)], float(sc)) for sc, r in zip(D[qi], I[qi]) if reference_padding or r >= 0]
        out[qid] = dict(pairs)
    return out


def search_chunks(query_emb: np.ndarray, query_ids: list[str], corpus_emb: np.ndarray, corpus_ids: list[str],
                  top_k: int, corpus_chunk_size: int, ignore_identical_ids: bool = False,
                  reference_padding: bool = False) -> dict[str, dict[str, float]]:
    """Dense part of HybridSearch.search / DenseRetrievalFaissSearch.search given already-encoded, already
    length-sorted embeddings: per chunk index -> retrieve_with_emb -> heap merge -> {qid: {pid: score}}
    (hybrid_search.py:301-358, faiss_search.py:228-291).  reference_padding: see retrieve_with_emb (a chunk shorter than
    top_k makes the reference lose that chunk's last document; the product does not reproduce that)."""
    heaps = {qid: [] for qid in query_ids}
    for s in range(0, len(corpus_ids), corpus_chunk_size):
        e = min(s + corpus_chunk_size, len(corpus_ids))
        sub = retrieve_with_emb(query_emb, query_ids, corpus_emb[s:e], corpus_ids[s:e], top_k, reference_padding)
        add_to_heap(sub, heaps, top_k, ignore_identical_ids)
    return {qid: {pid: score for score, pid in heaps[qid]} for qid in query_ids}     # _parse_heap_results, hybrid_search.py:347-355


# --------------------------------------------------------------------------------------
# LoRA merge (finetune/modeling_encoder.py:616-625 -> peft merge_and_unload: W + (alpha/r) * B @ A)
# --------------------------------------------------------------------------------------


def lora_merge(W: np.ndarray, A: np.ndarray, B: np.ndarray, alpha: float, r: int) -> np.ndarray:
    return (W.astype(np.float32) + np.float32(alpha / r) * (B.astype(np.float32) @ A.astype(np.float32))).astype(np.float32)


# --------------------------------------------------------------------------------------
# Sparse document vectors
# --------------------------------------------------------------------------------------
BF16_MIN = np.float32(-3.3895313892515355e38)      # torch.finfo(torch.bfloat16).min
F32_MIN = np.float32(np.finfo(np.float32).min)


def sparse_attention_mask(input_ids: np.ndarray, attention_mask: np.ndarray, sep_token_id: Optional[int], remove_prompt: bool = False) -> np.ndarray:
    """finetune/sparse_pooling.py:23-59 (get_sparse_attention_mask + get_prompt_mask) on the padded [B,S] layout: valid tokens
    minus column 0, minus each row's last valid token, minus (remove_prompt) everything up to and including the first
    sep token -- with the reference's quirks: rows without a sep lose only column 0; no masking at all when no row holds a
    sep, or when every row's first sep sits in the last COLUMN."""
    ids = np.asarray(input_ids)
    m = np.asarray(attention_mask).astype(bool).copy()
    B, S = ids.shape
    if remove_prompt and sep_token_id is not None and (ids == sep_token_id).any():
        pos = np.argmax((ids == sep_token_id).astype(np.int64), axis=-1)
        if not np.all(pos == S - 1):
            m[np.arange(S)[None, :] <= pos[:, None]] = False
    last = np.asarray(attention_mask).sum(1) - 1          # -1 (empty row) indexes the last column like torch does
    m[:, 0] = False
    m[np.arange(B), last] = False
    return m


def max_aggregate(hidden: np.ndarray, W: np.ndarray, bias: Optional[np.ndarray], mask: np.ndarray, bf16: bool = False) -> np.ndarray:
    """utils/max_linear_map.py:8-88 (MaxLinearMapperFunction.forward): out[b, v] = max over valid t of hidden[b, t] . W[v] (+ bias),
    starting from finfo(dtype).min (rows with no valid token keep it).  hidden [B,S,H], W [V,H] (= lm_head.weight), mask [B,S].
    bf16=True restates the autocast run: every logit is rounded to bf16 before the max (dtype of `input_sliced @ weight`)."""
    B, S, H = hidden.shape
    lo = BF16_MIN if bf16 else F32_MIN
    out = np.full((B, W.shape[0]), lo, np.float32)
    Wt = np.ascontiguousarray(W.T, dtype=np.float32)
    for t in range(S):
        logits = hidden[:, t].astype(np.float32) @ Wt
        if bias is not None:
            logits = logits + bias
        if bf16:
            logits = round_bf16(logits)
        logits = np.where(mask[:, t:t + 1], logits, lo)
        out = np.where(logits > out, logits, out)
    return out


def max_aggregate_packed(hidden: np.ndarray, cu_seqlens: np.ndarray, tok_mask: np.ndarray, W: np.ndarray, bias: Optional[np.ndarray] = None,
                         bf16: bool = True) -> np.ndarray:
    """Same on the packed layout: hidden [T,H], tok_mask [T] (the sparse attention mask selected at the valid tokens)."""
    B = len(cu_seqlens) - 1
    lo = BF16_MIN if bf16 else F32_MIN
    out = np.full((B, W.shape[0]), lo, np.float32)
    logits = hidden.astype(np.float32) @ np.ascontiguousarray(W.T, dtype=np.float32)
    if bias is not None:
        logits = logits + bias
    if bf16:
        logits = round_bf16(logits)
    for b in range(B):
        s, e = int(cu_seqlens[b]), int(cu_seqlens[b + 1])
        sel = logits[s:e][np.asarray(tok_mask[s:e]).astype(bool)]
        if sel.shape[0]:
            out[b] = np.maximum(out[b], sel.max(0))
    return out


def top_k_sampling(scores: np.ndarray, top_k: int, filter_value: float = 0.0, min_tokens_to_keep: int = 1) -> np.ndarray:
    """finetune/sparse_pooling.py:92-109: everything below the k-th largest value of the row -> filter_value (ties with the
    k-th value all survive)."""
    if top_k <= 0:
        return scores
    k = min(max(top_k, min_tokens_to_keep), scores.shape[-1])
    kth = np.sort(scores, axis=-1)[:, ::-1][:, k - 1:k]
    return np.where(scores < kth, np.float32(filter_value), scores)


def top_p_sampling(scores: np.ndarray, top_p: float, filter_value: float = 0.0, min_tokens_to_keep: int = 1) -> np.ndarray:
    """finetune/sparse_pooling.py:64-90: ascending sort, softmax, cumulative sum; entries with cumulative probability
    <= 1 - top_p are dropped, the last min_tokens_to_keep of the sorted order always stay."""
    if top_p <= 0 or top_p >= 1:
        return scores
    order = np.argsort(scores, axis=-1, kind="stable")
    srt = np.take_along_axis(scores, order, -1).astype(np.float32)
    e = np.exp(srt - srt.max(-1, keepdims=True))
    cum = np.cumsum((e / e.sum(-1, keepdims=True)).astype(np.float32), axis=-1, dtype=np.float32)
    remove_sorted = cum <= np.float32(1 - top_p)
    remove_sorted[:, -min_tokens_to_keep:] = False
    remove = np.zeros_like(remove_sorted)
    np.put_along_axis(remove, order, remove_sorted, -1)
    return np.where(remove, np.float32(filter_value), scores)


def sparsify(logits: np.ndarray, relu: bool = True, log1p: bool = True, top_p: float = 1.0, top_k: int = 0, min_tokens_to_keep: int = 8,
             bf16: bool = False) -> np.ndarray:
    """finetune/modeling_hybrid.py:176-203 (get_sparse_emb, the branches the asymmetric config uses): relu -> log1p -> top-p ->
    top-k.  bf16=True: the tensor is bf16 in the autocast run, so log1p's result is rounded to bf16."""
    x = logits.astype(np.float32)
    if relu:
        x = np.maximum(x, np.float32(0))
    if log1p:
        x = np.log1p(x).astype(np.float32)
        if bf16:
            x = round_bf16(x)
    x = top_p_sampling(x, top_p, min_tokens_to_keep=min_tokens_to_keep)
    x = top_k_sampling(x, top_k, min_tokens_to_keep=min_tokens_to_keep)
    return x


def quantize_sparse(reps: np.ndarray, quantization_factor: int = 100) -> np.ndarray:
    """finetune/sparse_converter_mixin.py:129-133: clamp(min=0) -> round-half-even(reps * q) -> int32."""
    return np.rint(np.maximum(reps.astype(np.float32), np.float32(0)) * np.float32(quantization_factor)).astype(np.int32)


def sparse_reps_to_json(reps: np.ndarray, quantization_factor: int = 100, vocab: Optional[dict] = None) -> list[dict]:
    """finetune/sparse_converter_mixin.py:105-160 (convert_sparse_reps_to_json_pt): {str(token id) | token: integer weight} for
    the non-zero quantised entries in ascending token id order; an empty vector becomes {"-1": 1} ({"[PAD]": 1} with a vocab)."""
    q = quantize_sparse(np.atleast_2d(reps), quantization_factor)
    out = []
    for row in q:
        nz = np.nonzero(row)[0]
        d = {(vocab[int(i)] if vocab is not None else str(int(i))): int(row[i]) for i in nz}
        if not d:
            d = {"[PAD]": 1} if vocab is not None else {"-1": 1}
        out.append(d)
    return out


def sparse_reps_to_pseudo_text(reps: np.ndarray, quantization_factor: int = 100, vocab: Optional[dict] = None) -> list[str]:
    """finetune/sparse_converter_mixin.py:162-189 (convert_sparse_reps_to_pseudo_text_pt): every token (ascending id) repeated its quantised weight
    times, space-joined -- what call_batch_encode hands the sparse engine for a QUERY vector (inference/exact_search_base.py:231-236); an empty
    vector becomes "-1"."""
    return [" ".join(tok for tok, freq in d.items() for _ in range(freq)) for d in sparse_reps_to_json(reps, quantization_factor, vocab)]


def keep_input_token_scores(agg: np.ndarray, ids: np.ndarray, cu_seqlens: np.ndarray, tok_mask: np.ndarray, filter_value: float = 0.0) -> np.ndarray:
    """`--sparse_pool_from_original_input_ids_qry / _psg` (modeling_hybrid.py:175-180): get_unique_token_ids over the sequence's tokens under the
    sparse attention mask (sparse_pooling.py:147-156), then get_scores_with_indices (:158-179) -- every vocabulary entry that is not one of the
    sequence's own (unmasked) tokens becomes filter_value BEFORE relu / log1p / top-p / top-k: a sparse vector without expansion terms."""
    out = np.full_like(agg, np.float32(filter_value), dtype=np.float32)
    for b in range(len(cu_seqlens) - 1):
        s, e = int(cu_seqlens[b]), int(cu_seqlens[b + 1])
        own = np.unique(np.asarray(ids[s:e])[np.asarray(tok_mask[s:e]).astype(bool)])
        out[b, own] = agg[b, own]
    return out


def encode_query_sparse(cfg: EncoderConfig, w, ids, cu_seqlens, tok_mask, lm_head: Optional[np.ndarray] = None, bf16: bool = True, **sparsify_kw):
    """HybridModel.encode_query's sparse branch (modeling_hybrid.py:404-438): the passage pipeline on the query's tokens; get_sparse_emb(is_query=True)
    differs only in WHICH top-p / top-k ratios apply (sparse_top_p_qry / sparse_top_k_qry, :189-200) -- pass them as top_p / top_k."""
    return encode_passage_sparse(cfg, w, ids, cu_seqlens, tok_mask, lm_head=lm_head, bf16=bf16, **sparsify_kw)


def encode_passage_sparse(cfg: EncoderConfig, w, ids, cu_seqlens, tok_mask, lm_head: Optional[np.ndarray] = None, bf16: bool = True,
                          pool_from_input_ids: bool = False, **sparsify_kw):
    """HybridModel.encode_passage's sparse branch (modeling_hybrid.py:280-323) on packed input: LM forward -> final-norm hidden
    states -> max aggregation with the (tied unless given) LM head -> (pool_from_input_ids: only the sequence's own tokens, :175-180) -> sparsify."""
    hidden = encoder_forward_packed(cfg, w, ids, cu_seqlens, bf16=bf16)
    W = lm_head if lm_head is not None else w["embed_tokens.weight"]
    agg = max_aggregate_packed(hidden, cu_seqlens, tok_mask, W, None, bf16=bf16)
    if pool_from_input_ids:
        agg = keep_input_token_scores(agg, ids, cu_seqlens, tok_mask)
    return sparsify(agg, bf16=bf16, **sparsify_kw)


# --------------------------------------------------------------------------------------
# Hit-list fusion -- retriever/score_fuse_utils.py:3-91, numpy float64 like the reference
# --------------------------------------------------------------------------------------
def fuse_scores_rrf(results_list: list[dict], k: int = 60) -> dict:
    """score_fuse_utils.py:3-45: per system and query, rank by descending score, add 1 / (k + rank) per passage."""
    fused: dict = {}
    for res in results_list:
        for qid, passages in res.items():
            out = fused.setdefault(str(qid), {})
            pids = list(passages.keys())
            order = np.argsort(-np.array([float(passages[p]) for p in pids], dtype=np.float64), kind="stable")
            for rank, j in enumerate(order, start=1):
                out[str(pids[j])] = out.get(str(pids[j]), 0.0) + float(1 / np.float64(k + rank))
    return fused


def fuse_scores_linear(results_list: list[dict], weights=(0.7, 0.3), eps: float = 1e-8) -> dict:
    """score_fuse_utils.py:47-91: (s - min) / (max - min + eps) * weight per system and query, summed per passage."""
    assert len(results_list) == len(weights)
    fused: dict = {}
    for res, wgt in zip(results_list, weights):
        for qid, passages in res.items():
            out = fused.setdefault(str(qid), {})
            pids = list(passages.keys())
            sc = np.array([float(passages[p]) for p in pids], dtype=np.float64)
            normed = (sc - sc.min()) / (sc.max() - sc.min() + eps) * wgt
            for p, v in zip(pids, normed):
                out[str(p)] = out.get(str(p), 0.0) + float(v)
    return fused
