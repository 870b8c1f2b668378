"""Shared helpers for the parity tests (test infrastructure: may import oracle/)."""
import os

import numpy as np

from oracle import lrx_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_model_golden(name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    kw = {}
    for k in g.files:
        if k.startswith("cfg."):
            v = g[k]
            kw[k[4:]] = v.item() if v.shape == () else v
    for k in ("rope_type",):
        kw[k] = str(kw[k])
    kw["qkv_bias"] = bool(kw["qkv_bias"])
    for k in ("vocab_size", "hidden_size", "num_layers", "num_q_heads", "num_kv_heads", "head_dim", "intermediate_size",
              "rope_original_max_position", "max_positions"):
        kw[k] = int(kw[k])
    cfg = O.EncoderConfig(**kw)
    w = O.random_weights(cfg, seed=int(g["weight_seed"]), std=float(g["weight_std"]), bf16=True)
    if any(k.startswith("w.") for k in g.files):  # fixture that also carries the literal HF state_dict
        for k in g.files:
            if k.startswith("w."):
                np.testing.assert_array_equal(g[k], w[k[2:]], err_msg=k)
    ids_nested, pos, indices, cu, max_len = O.pack_padded(g["input_ids"], g["attention_mask"])
    return cfg, w, g, ids_nested.astype(np.int32), cu, max_len


MODEL_GOLDENS = ["llama_tiny_l3rope", "llama_tiny_allfull", "qwen2_tiny", "llama_small_d128", "llama_small_d64"]


def min_cos(a, b):
    a = a / np.linalg.norm(a, axis=-1, keepdims=True)
    b = b / np.linalg.norm(b, axis=-1, keepdims=True)
    return float((a * b).sum(-1).min())


def load_search_ref():
    """tests/golden/search_ref.json (outputs of the reference's own searchers, gen_search_goldens.py) with arrays decoded."""
    import base64
    import json
    fx = json.load(open(os.path.join(GOLDEN, "search_ref.json")))
    for st in fx["sets"].values():
        for k in ("X", "emb_reps", "dense_reps"):
            a = st[k]
            st[k] = np.frombuffer(base64.b64decode(a["b64"]), dtype=a["dtype"]).reshape(a["shape"]).copy()
    return fx


def flat_ip_topk_fp64(q, X, k):
    """The product's definition of the result, evaluated on the host: every inner product accumulated in fp64 and rounded ONCE to fp32, hits
    by (score descending, row ascending).  For tests whose rows tie inside the noise of an fp32 summation (O.flat_ip_topk is a numpy sgemm)."""
    S = (np.asarray(q, np.float64) @ np.asarray(X, np.float64).T).astype(np.float32)
    order = np.lexsort((np.broadcast_to(np.arange(S.shape[1]), S.shape), -S), axis=-1)[:, :k]
    return np.take_along_axis(S, order, axis=1), order


def load_query_modes():
    """tests/golden/query_modes.npz (gen_query_goldens.py: the reference's encode_query in its LM-encoded modes on the llama_small_d64
    model) -> (oracle cfg, weights, fixture, meta dict)."""
    import json
    cfg, w, _, _, _, _ = load_model_golden("llama_small_d64")
    g = np.load(os.path.join(GOLDEN, "query_modes.npz"))
    return cfg, w, g, json.loads(bytes(g["meta_json"]).decode())
