"""Host-side mirror of the reference's model-side interfaces for the dense asymmetric path.

    EncodeCollator      <- inference/exact_search_base.py:267-437   (text -> token ids; here: packed, no second tokenisation)
    LrxHybridModel      <- finetune/modeling_hybrid.py:205-278 (encode_passage), :327-500 (encode_query, emb branch),
                           finetune/emb_bag_mixin.py / nonctx_emb_utils.py:239-313 (EmbeddingBag construction)
    LrxExactSearchModel <- inference/exact_search_base.py:42-200 + exact_search_torchrpc.py:122-295
                           (encode_queries / encode_corpus / encode with prompts; one process per GPU, no RPC)

Same names for arguments and result keys (`dense_reps`, `emb_reps`), same text formatting rules; tensors stay on the GPU.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Optional

import numpy as np
import torch

from . import ops
from .encoder import LrxEncoder


# ------------------------------------------------------------------------------------------------------------------
# text -> token ids
# ------------------------------------------------------------------------------------------------------------------
def format_text(item: dict, prepend_prompt: bool = False) -> str:
    """`title + " " + text` when a non-empty title exists; the prompt (if the item carries one) is string-prepended
    with no separator (exact_search_base.py:289-309)."""
    text = item["title"] + " " + item["text"] if item.get("title") else item["text"]
    if prepend_prompt and "prompt" in item:
        text = item["prompt"] + text
    return text


@dataclass
class EncodeCollator:
    """Tokenises a list of {"title"?, "text", "prompt"?} items.

    Documents: specials added (`<bos> A <eos>` template of the tokenizer), truncation 'only_first' to p_max_len (EOS is
    kept as the last token), output PACKED: `input_ids` int32 [T], `cu_seqlens` int32 [B+1], `max_seqlen` -- plus, on
    request, the reference's right-padded `[B, S]` ids/mask for callers that want the original layout.
    Queries (noncontextual_query_embedding): raw text, no prompt, no specials, truncated to q_max_len;
    `nonctx_tok_emb_input_ids` int64 [sum len], `nonctx_tok_emb_offsets` int64 [Q] = cumsum([0] + len[:-1])
    (nonctx_emb_utils.py:197-219).  Queries that go through the LM (symmetric dense vector, `hybrid_use_dense_vector`; the LM's input
    embedding layer as the bag, `noncontextual_query_embedding=False`): `prompt + text` with specials, truncation 'only_first' to
    q_max_len, packed like the documents -- the reference tokenises that for EVERY query batch (exact_search_base.py:333-345); here it
    is made on request (`query_lm_inputs`) and rides next to the EmbeddingBag fields in the same dict."""
    tokenizer: object
    encode_is_query: bool
    q_max_len: int = 512
    p_max_len: int = 512
    noncontextual_query_embedding: bool = True
    return_padded: bool = False
    sparse_mask: bool = False                 # also emit `sparse_mask` uint8 [T] (get_sparse_attention_mask) for the sparse branch
    sep_token_id: Optional[int] = None
    add_sep_token: bool = False
    query_lm_inputs: Optional[bool] = None    # queries: also emit the packed `prompt + text` LM inputs (None = only when no EmbeddingBag fields are made)

    def __call__(self, texts: list[dict]) -> dict:
        bag = {}
        if self.encode_is_query and self.noncontextual_query_embedding:
            enc = self.tokenizer([format_text(t) for t in texts], max_length=self.q_max_len, truncation=True,
                                 add_special_tokens=False, return_attention_mask=False)["input_ids"]
            lens = [len(e) for e in enc]
            flat = np.concatenate([np.asarray(e, dtype=np.int64) for e in enc]) if sum(lens) else np.zeros(0, np.int64)
            offsets = np.cumsum([0] + lens[:-1]).astype(np.int64)
            bag = {"nonctx_tok_emb_input_ids": torch.from_numpy(flat), "nonctx_tok_emb_offsets": torch.from_numpy(offsets)}
            if not self.query_lm_inputs:
                return bag
        max_len = self.q_max_len if self.encode_is_query else self.p_max_len
        enc = self.tokenizer([format_text(t, prepend_prompt=True) for t in texts], max_length=max_len, truncation="only_first",
                             padding=False, add_special_tokens=True, return_attention_mask=False)["input_ids"]
        lens = np.asarray([len(e) for e in enc], dtype=np.int64)
        if (lens == 0).any():
            raise ValueError("EncodeCollator: a document tokenised to zero tokens (tokenizer adds no special tokens?)")
        out = {"input_ids": torch.from_numpy(np.concatenate([np.asarray(e, dtype=np.int32) for e in enc])),
               "cu_seqlens": torch.from_numpy(np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)),
               "max_seqlen": int(lens.max())}
        if self.sparse_mask:
            out["sparse_mask"] = torch.from_numpy(sparse_token_mask(out["input_ids"].numpy(), out["cu_seqlens"].numpy(), self.sep_token_id,
                                                                    self.add_sep_token))
        if self.return_padded:
            pad = self.tokenizer.pad_token_id
            ids = np.full((len(enc), int(lens.max())), pad, dtype=np.int64)
            mask = np.zeros_like(ids)
            for i, e in enumerate(enc):
                ids[i, :len(e)], mask[i, :len(e)] = e, 1
            out["padded_input_ids"], out["padded_attention_mask"] = torch.from_numpy(ids), torch.from_numpy(mask)
        out.update(bag)
        return out


def sparse_token_mask(input_ids: np.ndarray, cu_seqlens: np.ndarray, sep_token_id: Optional[int] = None, remove_prompt: bool = False) -> np.ndarray:
    """get_sparse_attention_mask (finetune/sparse_pooling.py:23-59) on the packed layout -> uint8 [T]: every token except each
    sequence's first and last one and, with remove_prompt, everything up to and including the sequence's first sep token.
    Reference quirks kept: a sequence without a sep loses nothing more; no prompt masking when no sequence of the batch
    holds a sep, or when every sequence's first sep is in the last column of the padded batch (= all sequences full length
    with their only sep at the end)."""
    ids = np.asarray(input_ids)
    cu = np.asarray(cu_seqlens, dtype=np.int64)
    lens = np.diff(cu)
    T = int(cu[-1])
    seg = np.repeat(np.arange(len(lens)), lens)
    pos = np.arange(T) - cu[seg]
    mask = np.ones(T, dtype=bool)
    if remove_prompt and sep_token_id is not None and T > 0:
        is_sep = ids[:T] == sep_token_id
        if is_sep.any():
            BIG = np.iinfo(np.int64).max
            first = np.full(len(lens), BIG, dtype=np.int64)
            np.minimum.at(first, seg, np.where(is_sep, pos, BIG))
            first[first == BIG] = 0                               # argmax of an all-False row
            if not np.all(first == int(lens.max()) - 1):
                mask &= pos > first[seg]
    mask &= pos != 0
    mask &= pos != lens[seg] - 1
    return mask.astype(np.uint8)


def pack_padded_batch(input_ids: torch.Tensor, attention_mask: torch.Tensor):
    """[B,S] ids + mask (the reference's psg dict) -> packed (ids int32 [T], cu_seqlens int32 [B+1], max_seqlen);
    the integer work of utils/nested_input.py:15-39."""
    mask = attention_mask.bool()
    lens = mask.sum(dim=1)
    ids = input_ids[mask].to(torch.int32).contiguous()
    cu = torch.zeros(lens.numel() + 1, dtype=torch.int32, device=input_ids.device)
    cu[1:] = torch.cumsum(lens, 0)
    return ids, cu, int(lens.max().item())


def _top_p_filter(scores: torch.Tensor, top_p: float, min_tokens_to_keep: int) -> torch.Tensor:
    """top_p_sampling (finetune/sparse_pooling.py:64-90) with device tensor ops; an option that is off (top_p = 1) in the
    published configuration, so it is not a kernel."""
    srt, idx = torch.sort(scores, descending=False, stable=True)
    cum = srt.softmax(dim=-1).cumsum(dim=-1)
    rem = cum <= (1 - top_p)
    rem[..., -min_tokens_to_keep:] = False
    remove = torch.zeros_like(rem).scatter(1, idx, rem)
    return scores.masked_fill(remove, 0.0)


# ------------------------------------------------------------------------------------------------------------------
# per-batch operators (B3)
# ------------------------------------------------------------------------------------------------------------------
class LrxHybridModel:
    """encode_passage / encode_query of the reference's HybridModel (one tied encoder, lasttoken pooling): documents through the LM;
    queries as the asymmetric EmbeddingBag vector (`hybrid_use_emb_vector` + `noncontextual_query_embedding`, the default here), as the
    symmetric dense vector (`hybrid_use_dense_vector`: the query through the same LM, modeling_hybrid.py:363-401) or as the mean of the
    LM's input embeddings (`hybrid_use_emb_vector` without `noncontextual_query_embedding`, the reference's ablation, :476-486);
    score_function cos_sim -> normalize=True."""

    def __init__(self, encoder: LrxEncoder, normalize: bool = True, dense_shrink_dim: Optional[int] = None,
                 pad_token_id: Optional[int] = None, encode_sparse: bool = False, sep_token_id: Optional[int] = None,
                 add_sep_token: bool = False, sparse_use_relu: bool = True, sparse_use_log_saturation: bool = True,
                 sparse_top_k_psg: int = 0, sparse_top_p_psg: float = 1.0, sparse_min_tokens_to_keep: int = 8,
                 sparse_round_bf16: bool = True, hybrid_use_dense_vector: bool = False, hybrid_use_emb_vector: bool = True,
                 noncontextual_query_embedding: bool = True, pooling_strategy: Optional[str] = None, hybrid_use_sparse_vector: bool = False,
                 sparse_top_k_qry: int = 0, sparse_top_p_qry: float = 1.0, hybrid_use_token_id_vector: Optional[bool] = None,
                 sparse_pool_from_original_input_ids_qry: bool = False, sparse_pool_from_original_input_ids_psg: bool = False):
        """The sparse_* / add_sep_token / sep_token_id / hybrid_use_* / noncontextual_query_embedding fields carry the reference's
        ModelArguments of the same names (finetune/arguments.py:175-290); encode_sparse = hybrid_use_sparse_vector or
        hybrid_use_token_id_vector."""
        self.encoder = encoder
        # `--pooling_strategy` (finetune/arguments.py:85-90; it overrides pooling_strategy_qry / _psg, :329-331): the strategy of the dense
        # document vector and of the LM-encoded dense query vector.  None is served as the released models' 'lasttoken'.
        self.pooling_strategy = pooling_strategy or "lasttoken"
        from . import _lib as _l
        if self.pooling_strategy not in _l.POOLING:
            raise NotImplementedError(f"--pooling_strategy {self.pooling_strategy}: served are {sorted(_l.POOLING)}")
        if encode_sparse and self.pooling_strategy != "lasttoken":
            raise NotImplementedError("dense + sparse document vectors in one pass (lrx_encode_packed_sparse) pool the dense vector from the last token only")
        self.hybrid_use_dense_vector = hybrid_use_dense_vector
        # `--hybrid_use_sparse_vector`: LM-encoded sparse QUERY vectors (LM head max aggregation on the query's tokens, modeling_hybrid.py:404-438;
        # the `spr` / `den_spr` query modes); the document side produces its sparse vector whenever encode_sparse is on
        self.hybrid_use_sparse_vector = hybrid_use_sparse_vector
        # `--hybrid_use_token_id_vector`: parameter-free sparse queries (token-id counts); None = whenever the sparse half is on without LM-head queries
        self.hybrid_use_token_id_vector = (encode_sparse and not hybrid_use_sparse_vector) if hybrid_use_token_id_vector is None else hybrid_use_token_id_vector
        self.sparse_top_k_qry = sparse_top_k_qry
        self.sparse_top_p_qry = sparse_top_p_qry
        # `--sparse_pool_from_original_input_ids_qry / _psg` (modeling_hybrid.py:175-180): the aggregated logits keep only the entries of the
        # sequence's own tokens (sparse attention mask) before relu / log1p / top-k -- a sparse vector without expansion terms
        self.sparse_pool_from_original_input_ids_qry = bool(sparse_pool_from_original_input_ids_qry)
        self.sparse_pool_from_original_input_ids_psg = bool(sparse_pool_from_original_input_ids_psg)
        self.hybrid_use_emb_vector = hybrid_use_emb_vector
        self.noncontextual_query_embedding = noncontextual_query_embedding
        self._lm_emb_table: Optional[torch.Tensor] = None   # fp32 copy of embed_tokens, made on first use by the input-embedding bag
        self.normalize = normalize
        self.dense_shrink_dim = dense_shrink_dim
        self.pad_token_id = pad_token_id
        self.encode_sparse = encode_sparse
        self.sep_token_id = sep_token_id
        self.add_sep_token = add_sep_token
        self.sparse_use_relu = sparse_use_relu
        self.sparse_use_log_saturation = sparse_use_log_saturation
        self.sparse_top_k_psg = sparse_top_k_psg
        self.sparse_top_p_psg = sparse_top_p_psg
        self.sparse_min_tokens_to_keep = sparse_min_tokens_to_keep
        self.sparse_round_bf16 = sparse_round_bf16
        self.emb_bag: Optional[torch.Tensor] = None      # fp32 [V, H] on the GPU (weight of the reference's nn.EmbeddingBag)
        self.emb_bag_prompt: Optional[str] = None

    @property
    def device(self):
        return self.encoder.device

    def encode_passage(self, psg: Optional[dict], normalize: Optional[bool] = None, out: Optional[torch.Tensor] = None,
                       encode_sparse: Optional[bool] = None, **kwargs):
        """psg: packed {"input_ids" [T] i32, "cu_seqlens" [B+1] i32, "max_seqlen"} or the reference's padded
        {"input_ids" [B,S], "attention_mask" [B,S]}.  -> {"dense_reps": fp32 [B, D] (GPU; `out` lets the caller pass index rows)}
        and, with encode_sparse, {"sparse_reps": fp32 [B, V]} (modeling_hybrid.py:280-323)."""
        if psg is None:
            return None
        normalize = self.normalize if normalize is None else normalize
        if encode_sparse or (encode_sparse is None and self.encode_sparse):
            return self._encode_passage_sparse(psg, bool(normalize), out)
        ids, cu, max_len = self._packed_lm_inputs(psg, "encode_passage")
        reps = self.encoder.encode_packed(ids, cu, max_len, out=out, out_dim=self.dense_shrink_dim, normalize=bool(normalize),
                                          pooling=self.pooling_strategy)
        return {"dense_reps": reps}

    def _packed_lm_inputs(self, batch: dict, who: str):
        """(ids int32 [T], cu_seqlens int32 [B+1], max_seqlen) on the GPU from a packed batch or from the reference's padded ids + mask"""
        if "input_ids" not in batch:
            raise KeyError(f"{who}: the batch carries no `input_ids` (LM inputs)")
        ids = batch["input_ids"]
        if "cu_seqlens" in batch:
            return (ids.to(self.device, dtype=torch.int32, non_blocking=True), batch["cu_seqlens"].to(self.device, dtype=torch.int32, non_blocking=True),
                    int(batch["max_seqlen"]))
        if batch.get("attention_mask") is None:
            raise KeyError(f"{who}: padded input needs attention_mask")
        return pack_padded_batch(ids.to(self.device), batch["attention_mask"].to(self.device))

    def _encode_passage_sparse(self, psg: dict, normalize: bool, out: Optional[torch.Tensor], top_k: Optional[int] = None,
                               top_p: Optional[float] = None, want_dense: bool = True, pool_from_input_ids: Optional[bool] = None):
        """dense (optional) + sparse vectors of a batch in one pass.  top_k / top_p / pool_from_input_ids: the sampling ratios and the
        own-tokens-only switch of get_sparse_emb (modeling_hybrid.py:175-200) -- the passage ones by default, the *_qry ones when encode_query calls."""
        top_k = self.sparse_top_k_psg if top_k is None else top_k
        top_p = self.sparse_top_p_psg if top_p is None else top_p
        own_only = self.sparse_pool_from_original_input_ids_psg if pool_from_input_ids is None else bool(pool_from_input_ids)
        ids = psg["input_ids"]
        if "cu_seqlens" in psg:
            cu_host, ids_host = psg["cu_seqlens"], ids
            max_len = int(psg["max_seqlen"])
        else:
            ids_host, cu_host, max_len = pack_padded_batch(ids.cpu(), psg["attention_mask"].cpu())
        tok_mask = psg.get("sparse_mask")
        if tok_mask is None:       # the collator normally ships it; built here from the host copy of the ids otherwise
            tok_mask = torch.from_numpy(sparse_token_mask(ids_host.cpu().numpy(), cu_host.cpu().numpy(), self.sep_token_id, self.add_sep_token))
        ids_dev, cu_dev = ids_host.to(self.device, dtype=torch.int32), cu_host.to(self.device, dtype=torch.int32)
        mask_dev = tok_mask.to(self.device, dtype=torch.uint8).contiguous()
        dense, sparse = self.encoder.encode_packed_sparse(
            ids_dev, cu_dev, max_len, tok_mask=mask_dev, dense_dim=self.dense_shrink_dim, normalize=normalize, want_dense=want_dense,
            relu=self.sparse_use_relu and not own_only, log1p=self.sparse_use_log_saturation and not own_only, round_bf16=self.sparse_round_bf16,
            top_k=0 if (0 < top_p < 1 or own_only) else top_k, min_tokens_to_keep=self.sparse_min_tokens_to_keep)
        if own_only:
            # get_unique_token_ids + get_scores_with_indices (sparse_pooling.py:147-179) on the raw aggregated logits: every vocabulary entry that is
            # not one of the sequence's own unmasked tokens becomes 0; relu / log1p (and top-k, unless top-p comes first) follow on the result
            lens = (cu_dev[1:] - cu_dev[:-1]).long()
            rows = torch.repeat_interleave(torch.arange(lens.numel(), device=self.device), lens)
            sel = mask_dev.bool() & (ids_dev >= 0) & (ids_dev < sparse.shape[1])
            keep = torch.zeros(sparse.shape, dtype=torch.bool, device=self.device)
            keep[rows[sel], ids_dev[sel].long()] = True
            sparse.masked_fill_(~keep, 0.0)
            ops.sparsify_(sparse, relu=self.sparse_use_relu, log1p=self.sparse_use_log_saturation, round_bf16=self.sparse_round_bf16,
                          top_k=0 if 0 < top_p < 1 else top_k, min_tokens_to_keep=self.sparse_min_tokens_to_keep)
        if 0 < top_p < 1:
            # optional nucleus filter (sparse_pooling.py:64-90), off in the published configuration: plain device tensor ops,
            # then the top-k threshold kernel on its result (the reference's order: top-p before top-k)
            sparse = _top_p_filter(sparse, top_p, self.sparse_min_tokens_to_keep)
            ops.sparsify_(sparse, relu=False, log1p=False, top_k=top_k, min_tokens_to_keep=self.sparse_min_tokens_to_keep)
        if not want_dense:
            return {"sparse_reps": sparse}
        if out is not None:
            out[:dense.shape[0]].copy_(dense)
            dense = out[:dense.shape[0]]
        return {"dense_reps": dense, "sparse_reps": sparse}

    def convert_sparse_reps_to_json(self, reps: torch.Tensor, quantization_factor: int = 100, convert_id_to_token: bool = False,
                                    vocab_dict: Optional[dict] = None) -> list[dict]:
        """[{token id (str) | token: integer weight}] per row (finetune/sparse_converter_mixin.py:25-60, :105-160): weights =
        round(max(x, 0) * quantization_factor), zeros dropped, an empty vector becomes {"-1": 1} / {"[PAD]": 1}.  Quantise +
        compaction run on the GPU (lrx_sparse_compact); only the non-zeros come to the host."""
        if reps.dim() == 1:
            reps = reps.unsqueeze(0)
        reps = reps.to(self.device, dtype=torch.float32).contiguous()
        ids, w, cnt = ops.sparse_compact(reps, quantization_factor)
        cnt = cnt.cpu().numpy()
        ncap = int(cnt.max()) if cnt.size else 0
        ids, w = ids[:, :ncap].cpu().numpy(), w[:, :ncap].cpu().numpy()
        if convert_id_to_token and vocab_dict is None:
            raise ValueError("convert_id_to_token needs vocab_dict {id: token}")
        out = []
        for b in range(reps.shape[0]):
            n = int(cnt[b])
            if n == 0:
                out.append({"[PAD]": 1} if convert_id_to_token else {"-1": 1})
            elif convert_id_to_token:
                out.append({vocab_dict[int(i)]: int(v) for i, v in zip(ids[b, :n], w[b, :n])})
            else:
                out.append({str(int(i)): int(v) for i, v in zip(ids[b, :n], w[b, :n])})
        return out

    def convert_sparse_reps_to_pseudo_text(self, reps: torch.Tensor, quantization_factor: int = 100, convert_id_to_token: bool = False,
                                           vocab_dict: Optional[dict] = None) -> list[str]:
        """Each token repeated its quantised weight times, space-joined (finetune/sparse_converter_mixin.py:63-101, :162-189): the form
        call_batch_encode gives a sparse QUERY vector (inference/exact_search_base.py:231-236); an empty vector becomes "-1"."""
        return [" ".join(tok for tok, freq in d.items() for _ in range(freq))
                for d in self.convert_sparse_reps_to_json(reps, quantization_factor, convert_id_to_token, vocab_dict)]

    def encode_query(self, qry: Optional[dict], normalize: Optional[bool] = None, encode_dense: Optional[bool] = None,
                     encode_emb_reps: Optional[bool] = None, encode_sparse: Optional[bool] = None, **kwargs):
        """modeling_hybrid.py:327-500 -> a dict with, as enabled (argument override, else the model's flag -- the reference's rule :362-366):
          `dense_reps` fp32 [Q, D]: the query (`prompt + text`, specials) through the LM, lasttoken pooling, slice, normalise (:363-401);
          `emb_reps`   fp32 [Q, D]: EmbeddingBag mean over nonctx_tok_emb_input_ids / nonctx_tok_emb_offsets (:472-474), or -- without
                       `noncontextual_query_embedding` -- the mean of the LM's input embeddings over the query's tokens (:476-486);
                       slice, normalise (:487-490).
        LM inputs: packed {"input_ids" [T], "cu_seqlens" [Q+1], "max_seqlen"} or the reference's padded ids + attention_mask."""
        if qry is None:
            return None
        normalize = self.normalize if normalize is None else normalize
        encode_dense = bool(encode_dense or (encode_dense is None and self.hybrid_use_dense_vector))
        encode_emb = bool(encode_emb_reps or (encode_emb_reps is None and self.hybrid_use_emb_vector))
        encode_spr = bool(encode_sparse or (encode_sparse is None and self.hybrid_use_sparse_vector))
        out = {}
        if encode_spr:
            # `sparse_reps` fp32 [Q, V] (:404-438): LM head max aggregation over the query's tokens (sparse attention mask), relu / log1p, the *_qry
            # sampling ratios; the dense vector of the same forward comes out of the same pass (last-token pooling)
            if encode_dense and self.pooling_strategy != "lasttoken":
                raise NotImplementedError("dense + sparse query vectors in one pass pool the dense vector from the last token only")
            out.update(self._encode_passage_sparse(qry, bool(normalize), None, top_k=self.sparse_top_k_qry, top_p=self.sparse_top_p_qry, want_dense=encode_dense,
                                                   pool_from_input_ids=self.sparse_pool_from_original_input_ids_qry))
            encode_dense = False
        lm_in = self._packed_lm_inputs(qry, "encode_query") if (encode_dense or (encode_emb and not self.noncontextual_query_embedding)) else None
        if encode_dense:
            out["dense_reps"] = self.encoder.encode_packed(lm_in[0], lm_in[1], lm_in[2], out_dim=self.dense_shrink_dim, normalize=bool(normalize),
                                                           pooling=self.pooling_strategy)
        if encode_emb and self.noncontextual_query_embedding:
            if self.emb_bag is None:
                raise AssertionError("Please load or construct an EmbeddingBag before encoding queries")
            ids = qry["nonctx_tok_emb_input_ids"].to(self.device, dtype=torch.int64)
            offs = qry["nonctx_tok_emb_offsets"].to(self.device, dtype=torch.int64)
            out["emb_reps"] = ops.embedding_bag_mean(self.emb_bag, ids, offs, padding_idx=self.pad_token_id, out_dim=self.dense_shrink_dim,
                                                     normalize=bool(normalize))
        elif encode_emb:
            # the LM's own input embedding layer as the bag: every real token of the query counts (the attention mask is what the packed
            # layout already encodes), no padding_idx -- `pooling(inputs_embeds, attention_mask, 'mean')`
            if self._lm_emb_table is None:
                self._lm_emb_table = self.encoder.embed.float().contiguous()
            ids, cu = lm_in[0].to(torch.int64), lm_in[1]
            out["emb_reps"] = ops.embedding_bag_mean(self._lm_emb_table, ids, cu[:-1].to(torch.int64).contiguous(), padding_idx=None,
                                                     out_dim=self.dense_shrink_dim, normalize=bool(normalize))
        return out

    # -- EmbeddingBag construction (nonctx_emb_utils.py:239-313) --------------------------------------------------------
    def construct_embedding_bag(self, tokenizer, prompt: Optional[str] = None, batch_size: int = 5000, vocab_len: Optional[int] = None,
                                shared_prefix: bool = True, vocab_range: Optional[tuple[int, int]] = None):
        """For every tok in [0, len(tokenizer)): encode [bos] + prompt + [tok] + [eos] and keep the final hidden state of
        the last position, un-normalised, fp32 (nonctx_emb_utils.py:239-313).

        shared_prefix=True (default): [bos] + prompt is the same for every token, so it is encoded once and only the
        [tok, eos] suffixes go through the layers (lrx_encode_prefixed) - (P+3)/2 times fewer FLOPs, same table.
        shared_prefix=False: every sequence in full through lrx_encode_packed (the reference's literal loop).
        vocab_range=(lo, hi): build only those rows (one rank's slice, see construct_embedding_bag_distributed)."""
        V = vocab_len or len(tokenizer)
        lo, hi = vocab_range if vocab_range is not None else (0, V)
        assert 0 <= lo <= hi <= V
        bos, eos = tokenizer.bos_token_id, tokenizer.eos_token_id
        add_bos = bos is not None and bos in tokenizer.encode("", add_special_tokens=True)
        prefix = ([bos] if add_bos else []) + (tokenizer.encode(prompt, add_special_tokens=False) if prompt else [])
        H = self.encoder.cfg.hidden_size
        table = torch.empty(hi - lo, H, dtype=torch.float32, device=self.device)
        if shared_prefix:
            pre = torch.tensor(prefix, dtype=torch.int32, device=self.device)
            # 2 suffix tokens per row: six times the row count of the full path fits the same activation workspace
            step = batch_size * max(1, (len(prefix) + 2) // 2)
            for s in range(lo, hi, step):
                e = min(s + step, hi)
                suf = torch.empty(e - s, 2, dtype=torch.int32, device=self.device)
                suf[:, 0] = torch.arange(s, e, dtype=torch.int32, device=self.device)
                suf[:, 1] = eos
                self.encoder.encode_prefixed(pre, suf, out=table[s - lo:e - lo], normalize=False)
        else:
            L = len(prefix) + 2
            base = torch.empty(batch_size, L, dtype=torch.int32, device=self.device)
            if prefix:
                base[:, :len(prefix)] = torch.tensor(prefix, dtype=torch.int32, device=self.device)
            base[:, -1] = eos
            for s in range(lo, hi, batch_size):
                e = min(s + batch_size, hi)
                n = e - s
                base[:n, -2] = torch.arange(s, e, dtype=torch.int32, device=self.device)
                cu = (torch.arange(n + 1, device=self.device, dtype=torch.int64) * L).to(torch.int32)
                self.encoder.encode_packed(base[:n].reshape(-1), cu, L, out=table[s - lo:e - lo], normalize=False)
        if vocab_range is None:
            self.emb_bag, self.emb_bag_prompt = table, prompt
        return table

    def construct_embedding_bag_distributed(self, tokenizer, prompt: Optional[str] = None, batch_size: int = 5000,
                                            vocab_len: Optional[int] = None, group=None, shared_prefix: bool = True):
        """One vocabulary slice per rank, then an all-gather of the fp32 rows so every rank holds the whole table (the
        reference builds the full table on every rank; the rows are independent so the build shards with one collective)."""
        import torch.distributed as dist
        V = vocab_len or len(tokenizer)
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
            return self.construct_embedding_bag(tokenizer, prompt, batch_size, vocab_len, shared_prefix)
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        per = (V + world - 1) // world
        lo, hi = min(rank * per, V), min((rank + 1) * per, V)
        H = self.encoder.cfg.hidden_size
        mine = torch.zeros(per, H, dtype=torch.float32, device=self.device)
        if hi > lo:
            mine[:hi - lo] = self.construct_embedding_bag(tokenizer, prompt, batch_size, V, shared_prefix, vocab_range=(lo, hi))
        full = torch.empty(world * per, H, dtype=torch.float32, device=self.device)
        dist.all_gather_into_tensor(full, mine, group=group)
        self.emb_bag, self.emb_bag_prompt = full[:V].contiguous(), prompt
        return self.emb_bag

    def save_embedding_bag(self, path: str):
        """`torch.save(emb_bag.weight, path)`: the `*.emb_bag.pt` artefact of scripts/cache_emb_bag.ipynb."""
        assert self.emb_bag is not None, "construct_embedding_bag first"
        torch.save(self.emb_bag.detach().cpu(), path)

    def load_embedding_bag(self, weight, prompt: Optional[str] = None):
        """`torch.save(emb_bag.weight, '*.emb_bag.pt')` artefacts of scripts/cache_emb_bag.ipynb (tensor or path)."""
        if isinstance(weight, (str, bytes)) or hasattr(weight, "__fspath__"):
            weight = torch.load(weight, map_location="cpu", weights_only=True)
        if weight.dim() != 2 or weight.shape[1] != self.encoder.cfg.hidden_size:
            raise ValueError(f"EmbeddingBag table {tuple(weight.shape)} does not match hidden size {self.encoder.cfg.hidden_size}")
        self.emb_bag = weight.to(self.device, dtype=torch.float32).contiguous()
        self.emb_bag_prompt = prompt


# ------------------------------------------------------------------------------------------------------------------
# encode_queries / encode_corpus (B2)
# ------------------------------------------------------------------------------------------------------------------
def _as_items(texts) -> list[dict]:
    try:
        import datasets
        if isinstance(texts, datasets.Dataset):
            return texts.to_list()
    except ImportError:  # pragma: no cover
        pass
    if isinstance(texts, list):
        assert len(texts) > 0, "Empty lists."
        if isinstance(texts[0], str):
            return [{"text": t} for t in texts]
        if isinstance(texts[0], dict):
            return texts
        raise NotImplementedError(f"Unrecognized texts[0] type {type(texts[0])}")
    raise NotImplementedError(f"Unrecognized type {type(texts)}")


def _prefetch_batches(coll, items: list, batch_size: int, depth: int = 2):
    """(start, end, collated batch) in order, collated one worker thread ahead of the consumer: the tokenizer (Rust, releases the
    GIL) works on batch i+1.. while the caller copies and launches batch i.  The reference gets the same overlap from DataLoader
    worker processes (exact_search_torchrpc.py:185-200); order is preserved (row i of the output is input i)."""
    import queue
    import threading
    spans = [(s, min(s + batch_size, len(items))) for s in range(0, len(items), batch_size)]
    if len(spans) <= 1:
        for s, e in spans:
            yield s, e, coll(items[s:e])
        return
    q: "queue.Queue" = queue.Queue(maxsize=depth)
    stop = threading.Event()

    def work():
        try:
            for s, e in spans:
                if stop.is_set():
                    return
                q.put((s, e, coll(items[s:e])))
        except BaseException as ex:   # hand the failure to the consumer instead of dying silently
            q.put(ex)

    th = threading.Thread(target=work, name="lrx-collate", daemon=True)
    th.start()
    try:
        for _ in spans:
            got = q.get()
            if isinstance(got, BaseException):
                raise got
            yield got
    finally:
        stop.set()
        while th.is_alive():          # unblock a producer waiting on a full queue (consumer stopped early)
            try:
                q.get_nowait()
            except queue.Empty:
                th.join(0.01)


def _merge_packed(batches: list) -> dict:
    """Concatenate consecutive packed batches (collator output) into one: ids and masks appended, cu_seqlens re-based."""
    if len(batches) == 1:
        return batches[0]
    out = {"input_ids": torch.cat([b["input_ids"] for b in batches]), "max_seqlen": max(int(b["max_seqlen"]) for b in batches)}
    cu, base = [batches[0]["cu_seqlens"]], int(batches[0]["cu_seqlens"][-1])
    for b in batches[1:]:
        cu.append(b["cu_seqlens"][1:] + base)
        base += int(b["cu_seqlens"][-1])
    out["cu_seqlens"] = torch.cat(cu).to(torch.int32)
    if "sparse_mask" in batches[0]:
        out["sparse_mask"] = torch.cat([b["sparse_mask"] for b in batches])
    return out


def _token_budget_batches(gen, max_tokens: int, max_docs: int):
    """Merge consecutive (start, end, packed batch) items while the merged batch stays within `max_tokens` tokens and `max_docs`
    documents: short documents (the tail of a longest-first sorted corpus) then reach the encoder in batches of the same token
    count as a batch of full-length ones -- the GEMMs keep their tile count instead of shrinking with the documents.  Outputs are
    batch-invariant (tests), so the rows are the same as without merging."""
    pend, tok = [], 0
    for s, e, b in gen:
        t = int(b["cu_seqlens"][-1]) if "cu_seqlens" in b else None
        if t is None or max_tokens <= 0:                      # padded layout or merging off: pass through
            if pend:
                yield pend[0][0], pend[-1][1], _merge_packed([x[2] for x in pend])
                pend, tok = [], 0
            yield s, e, b
            continue
        if pend and (tok + t > max_tokens or (e - pend[0][0]) > max_docs):
            yield pend[0][0], pend[-1][1], _merge_packed([x[2] for x in pend])
            pend, tok = [], 0
        pend.append((s, e, b))
        tok += t
    if pend:
        yield pend[0][0], pend[-1][1], _merge_packed([x[2] for x in pend])


@dataclass
class LrxExactSearchModel:
    """DRES-style adapter: `encode_queries`, `encode_corpus`, `encode`; mutable `query_prompt`, `corpus_prompt`,
    `encoding_kwargs` (set per task by eval/evaluate_mteb.py:96-98).  Results: dict with `emb_reps` (queries) /
    `dense_reps` (corpus) like the reference's HybridModel path; row i of the output is input i."""
    model: LrxHybridModel
    tokenizer: object
    q_max_len: int = 512
    p_max_len: int = 512
    append_prompt_sep: bool = False
    eval_batch_size_embedding_bag: int = 5000
    query_prompt: Optional[str] = None
    corpus_prompt: Optional[str] = None
    encoding_kwargs: dict = field(default_factory=dict)
    token_id_vector_type: str = "sum"          # 'sum' | 'bow' (finetune/arguments.py:203-211): the parameter-free sparse query
    noncontextual_prompt_prefix: Optional[str] = None   # finetune/arguments.py: prepended to the query prompt inside the EmbeddingBag table
    max_batch_tokens: int = 131072             # encode(): consecutive batches are merged up to this many tokens (0 = off; 256 x 512)
    max_batch_docs: int = 2048                 # ... and this many documents (bounds the [docs, vocab] sparse activations)
    # the reference's EncoderModel returns a bare Tensor, its HybridModel a dict (exact_search_torchrpc.py:288-295: "if emb is single emb
    # type, we should return the emb alone"): True = hand back the only representation itself
    single_tensor_output: bool = False
    # What encode() / encode_queries() do when the device counters of the calls they made are non-zero (read ONCE per call, after the last
    # batch was enqueued: one blocking 8-byte copy per corpus chunk).  lrx_device_error_count (token ids outside the embedding table, an
    # attention work list that did not fit) always raises: those rows are not the model's.  lrx_device_saturation_count (q|k|v or shadow
    # elements that were NaN or beyond fp16's +-65504 and were stored as +-65504 -- the precondition of the fp16 attention operands,
    # csrc/lrx_gemm.hip:f2h_bits): "raise" (default), "warn" (log at WARNING with the count) or "ignore" (do not read either counter).
    on_fp16_saturation: str = "raise"

    def __post_init__(self):
        # every id the tokenizer can produce needs an embedding row (the reference grows the matrix with resize_emb,
        # utils/data_utils.py:273-281; loader.encoder_from_pretrained(tokenizer=...) does the same).  The gather kernel zero-fills and
        # counts out-of-range ids instead of reading outside the table, but a model that reaches it is misconfigured: fail here.
        enc = getattr(self.model, "encoder", None)
        if enc is not None and self.tokenizer is not None and hasattr(self.tokenizer, "__len__") and len(self.tokenizer) > enc.cfg.vocab_size:
            raise ValueError(f"tokenizer has {len(self.tokenizer)} tokens but the encoder only {enc.cfg.vocab_size} embedding rows "
                             "(load the encoder with loader.encoder_from_pretrained(..., tokenizer=tokenizer))")

    def _check_device_counters(self, what: str):
        """Make the library's device-side counters loud on the product path (nobody read them).  Synchronises."""
        if self.on_fp16_saturation == "ignore" or not isinstance(self.model, LrxHybridModel):     # (host-side tests drive the adapter with stand-ins)
            return
        if self.on_fp16_saturation not in ("raise", "warn"):
            raise ValueError(f"on_fp16_saturation={self.on_fp16_saturation!r}: 'raise', 'warn' or 'ignore'")
        from . import _lib
        lib = _lib.lib()
        bad, sat = int(lib.lrx_device_error_count(1)), int(lib.lrx_device_saturation_count(1))
        if bad < 0 or sat < 0:
            raise _lib.LrxError(f"{what}: reading the device counters failed")
        if bad:
            raise _lib.LrxError(f"{what}: {bad} device-side input error(s) (token ids outside the embedding table or an attention work list that "
                                "did not fit): the rows of this call are not the model's")
        if sat:
            msg = (f"{what}: {sat} wave instruction(s) met q|k|v / projection-operand / embedding elements that were NaN or beyond fp16's +-65504 and "
                   "stored them as +-65504: the embeddings of this call are not the model's (broken checkpoint, or activations outside the range the "
                   "fp16 operands assume -- EncoderConfig(operand_dtype='bf16') keeps the projection operands in bf16; q|k|v stay fp16)")
            if self.on_fp16_saturation == "raise":
                raise _lib.LrxError(msg)
            import logging
            logging.getLogger(__name__).warning(msg)

    def token_id_reps(self, items: list[dict]) -> list[dict]:
        """Parameter-free sparse query vectors (exact_search_base.py:380-431): raw text with a leading whitespace, no specials,
        {str(token id): count} ('sum') or {str(token id): 1} ('bow')."""
        from collections import Counter
        enc = self.tokenizer([" " + format_text(t) for t in items], max_length=self.q_max_len, truncation=True, add_special_tokens=False)["input_ids"]
        if self.token_id_vector_type == "bow":
            return [{str(t): 1 for t in set(e)} for e in enc]
        if self.token_id_vector_type == "sum":
            return [{str(k): v for k, v in Counter(e).items()} for e in enc]
        raise NotImplementedError(self.token_id_vector_type)

    def parse_texts(self, texts, prompt: Optional[str] = None) -> list[dict]:
        items = _as_items(texts)
        if prompt and not any("prompt" in it for it in items):   # `prompt` only applies when no prompt column exists
            if self.append_prompt_sep:
                prompt = prompt + self.tokenizer.sep_token + " "
            items = [dict(it, prompt=prompt) for it in items]
        return items

    def encode_queries(self, queries, batch_size: int, show_progress_bar: bool = True, convert_to_tensor: bool = True, **kwargs):
        """-> {"emb_reps"?, "dense_reps"?, "sparse_reps"?, "token_id_reps"?} by the model's flags (exact_search_base.py:94-122 +
        exact_search_torchrpc.py:139-170).  `sparse_reps` (`--hybrid_use_sparse_vector`): the LM-head query vectors as the quantised pseudo text
        call_batch_encode makes of them (inference/exact_search_base.py:231-236)."""
        items = self.parse_texts(queries, prompt=self.query_prompt)        # the `prompt` column the LM-encoded query vectors prepend
        hm = self.model
        use_dense = bool(getattr(hm, "hybrid_use_dense_vector", False))
        use_spr = bool(getattr(hm, "hybrid_use_sparse_vector", False))
        use_emb = bool(getattr(hm, "hybrid_use_emb_vector", True))
        nonctx = bool(getattr(hm, "noncontextual_query_embedding", True))
        if use_emb and nonctx:
            # the query prompt lives inside the EmbeddingBag table (exact_search_torchrpc.py:139-160): the model's query_prompt, else the
            # `prompt` column of the first query, with noncontextual_prompt_prefix in front; rebuilt when it changes
            prompt = self.query_prompt or (items[0].get("prompt") if items and isinstance(items[0], dict) else None)
            if self.noncontextual_prompt_prefix:
                prompt = self.noncontextual_prompt_prefix + prompt if prompt else self.noncontextual_prompt_prefix
            if hm.emb_bag is None or hm.emb_bag_prompt != prompt:
                hm.construct_embedding_bag(self.tokenizer, prompt=prompt, batch_size=self.eval_batch_size_embedding_bag)
                self._check_device_counters("construct_embedding_bag")
        need_lm = use_dense or use_spr or (use_emb and not nonctx)
        coll = EncodeCollator(self.tokenizer, encode_is_query=True, q_max_len=self.q_max_len, p_max_len=self.p_max_len,
                              noncontextual_query_embedding=use_emb and nonctx, query_lm_inputs=need_lm, sparse_mask=use_spr,
                              sep_token_id=getattr(hm, "sep_token_id", None), add_sep_token=bool(getattr(hm, "add_sep_token", False)))
        outs: dict = {}
        spr_text: list[str] = []
        for s in (range(0, len(items), batch_size) if (need_lm or (use_emb and nonctx)) else ()):     # (token-id-only models tokenise in token_id_reps)
            for k, v in hm.encode_query(coll(items[s:s + batch_size])).items():
                if k == "sparse_reps":    # [batch, vocab] fp32 on the GPU -> quantised pseudo text per query (only the non-zeros come to the host)
                    spr_text.extend(hm.convert_sparse_reps_to_pseudo_text(v, quantization_factor=100))
                else:
                    outs.setdefault(k, []).append(v)
        if need_lm:                       # (the EmbeddingBag lookup has no fp16 operand; the table build checks its own rows below)
            self._check_device_counters("encode_queries")
        res = {}
        for k, parts in outs.items():
            reps = torch.cat(parts, 0)
            res[k] = reps if convert_to_tensor else reps.cpu().numpy()
        if use_spr:
            res["sparse_reps"] = spr_text
        if getattr(hm, "hybrid_use_token_id_vector", hm.encode_sparse):
            res["token_id_reps"] = self.token_id_reps(items)
        return self._unwrap(res)

    def _unwrap(self, res: dict):
        if self.single_tensor_output:
            assert len(res) == 1, f"Not single representations: {list(res)}"
            return next(iter(res.values()))
        return res

    def encode_corpus(self, corpus, batch_size: int, show_progress_bar: bool = True, convert_to_tensor: bool = True,
                      out: Optional[torch.Tensor] = None, **kwargs):
        return self.encode(corpus, batch_size, show_progress_bar, convert_to_tensor, out=out, **kwargs)

    def encode(self, sentences, batch_size: int, show_progress_bar: bool = True, convert_to_tensor: bool = True,
               out: Optional[torch.Tensor] = None, **kwargs):
        items = self.parse_texts(sentences, prompt=self.corpus_prompt)
        sparse = self.model.encode_sparse
        # the reference's own launch (torch RPC, this rank drives, a model registered on every worker): a direct encode call fans the
        # documents out in contiguous spans and assembles the rows in input order, as PytorchRPCExactSearchModel._encode does
        # (exact_search_torchrpc.py:185-295).  Search keeps its shards on the workers instead (rpc_shards.index_chunk) and passes
        # rpc_fanout=False for the pieces a worker encodes itself.
        if kwargs.pop("rpc_fanout", True) and out is None and len(items) >= 2 * batch_size:
            from . import rpc_shards
            names = rpc_shards.rpc_workers()
            if len(names) > 1 and rpc_shards._WORKER.get("model") is self:
                return self._unwrap(rpc_shards.encode_fanout(names, items, batch_size, convert_to_tensor, self.model.device))
        coll = EncodeCollator(self.tokenizer, encode_is_query=False, q_max_len=self.q_max_len, p_max_len=self.p_max_len, sparse_mask=sparse,
                              sep_token_id=self.model.sep_token_id, add_sep_token=self.model.add_sep_token)
        D = self.model.dense_shrink_dim or self.model.encoder.cfg.hidden_size
        if out is None:
            out = torch.empty(len(items), D, dtype=torch.float32, device=self.model.device)
        sparse_json: list[dict] = []
        for s, e, batch in _token_budget_batches(_prefetch_batches(coll, items, batch_size), self.max_batch_tokens, self.max_batch_docs):
            r = self.model.encode_passage(batch, out=out[s:e])
            if sparse:   # quantised {token id: weight} per document, what call_batch_encode hands to the sparse engine
                sparse_json.extend(self.model.convert_sparse_reps_to_json(r["sparse_reps"], quantization_factor=100))
        self._check_device_counters("encode")
        reps = out[:len(items)]
        res = {"dense_reps": reps if convert_to_tensor else reps.cpu().numpy()}
        if sparse:
            res["sparse_reps"] = sparse_json
        return self._unwrap(res)
