"""Synthetic checkpoints with the statistics of a TRAINED Llama / Qwen2 backbone (no released weights are reachable offline).

`LrxEncoder.random_init(..., profile="trained_like")` uses this.  N(0, 0.02) weights (profile "gaussian") are the easy case for
16-bit rounding: q.k logits with sigma < 1 (softmax nearly uniform), no outlier channel, biases of 0.02.  The forward this build must
match is finetune/modeling_hybrid.py:248-278 on the released adapters (README.md:40-46), whose backbones show, as every trained LLM of
these families does:

  * peaky attention -- pre-softmax logits with a standard deviation of 5-10 across the keys of a query;
  * "massive activations": a handful of residual channels 50-100 x the typical one on every token, far larger still on the first
    token, which the other tokens attend to (attention sink);
  * heavy-tailed RMSNorm weights that squash exactly those channels;
  * (Qwen2) q / k / v biases of O(10-100), concentrated on the slowly rotating rotary pairs, a few beyond 100.

The generator builds the model layer by layer and CALIBRATES each layer on a few documents with an fp32 torch forward of the weights
it has just rounded to bf16 (weight synthesis, not the product path): the q / k scale by bisection on the measured logit spread, the
O-projection and down-projection scales on the size of what they add to the residual stream.  It returns the HF-named state dict and
the statistics it measured, so that a test can assert the regime rather than trust the recipe."""
from __future__ import annotations

import math

import torch

SINK_TOKEN = {128256: 128000, 151936: 151643, 152064: 151643}     # <|begin_of_text|> (Llama-3), <|endoftext|> (Qwen2.5): first token of the test documents


def sink_token(cfg) -> int:
    return SINK_TOKEN.get(cfg.vocab_size, 1)


def _rope(x, cos, sin):
    """x [T, heads, d] in HF's rotate_half layout; cos / sin [T, d/2] fp32"""
    d2 = x.shape[-1] // 2
    x1, x2 = x[..., :d2], x[..., d2:]
    c, s = cos[:, None, :], sin[:, None, :]
    return torch.cat([x1 * c - x2 * s, x2 * c + x1 * s], -1)


def _rms(x, eps):
    return x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps)


def trained_like_state_dict(cfg, seed: int = 0, device=None, logit_sigma=(5.0, 10.0), content_sigma=(0.75, 2.0), n_massive: int = 6, massive=(50.0, 100.0),
                            sink_boost: float = 8.0, sink_logit: float = 2.0, attn_add: float = 0.18, mlp_add: float = 0.28, bias_sigma: float = 20.0, pseudo_bias_sigma: float = 4.0, calib_lens=(96, 64, 33)):
    """-> (state_dict with HF names (bf16, on `device`), stats dict).  cfg: lightretriever_amd.EncoderConfig."""
    from .encoder import rope_tables
    device = device or torch.device("cuda", torch.cuda.current_device())
    gen = torch.Generator(device=device).manual_seed(seed)
    bf = torch.bfloat16
    H, d, I, nq, nkv, V = cfg.hidden_size, cfg.head_dim, cfg.intermediate_size, cfg.num_q_heads, cfg.num_kv_heads, cfg.vocab_size
    grp = nq // nkv

    def rn(*shape):
        return torch.randn(*shape, generator=gen, device=device, dtype=torch.float32)

    def ru(lo, hi, *shape):
        return lo + (hi - lo) * torch.rand(*shape, generator=gen, device=device, dtype=torch.float32)

    def q16(t):                                   # checkpoints are bf16: everything the calibration multiplies with is the rounded value
        return t.to(bf)

    def norm_weight(squash):
        g = torch.exp(0.5 * rn(H)) * 0.6                                      # log-normal body
        big = torch.randperm(H, generator=gen, device=device)[:8]
        g[big] *= ru(3.0, 6.0, 8)                                             # heavy tail
        g[mass_ch] = ru(0.02, 0.08, n_massive) if squash else g[mass_ch]      # trained norms squash the massive channels
        return q16(g)

    sd = {}
    sink = sink_token(cfg)
    # ---- embeddings: N(0, 0.02) body, n_massive channels at 50-100 sigma with a fixed sign on every token, x sink_boost on the sink token
    mass_ch = torch.randperm(H, generator=gen, device=device)[:n_massive]
    emb = rn(V, H) * 0.02
    mass_val = 0.02 * ru(massive[0], massive[1], n_massive) * torch.where(rn(n_massive) > 0, 1.0, -1.0)
    emb[:, mass_ch] = mass_val[None, :] * (1.0 + 0.1 * rn(V, n_massive))
    emb[sink, mass_ch] *= sink_boost
    sd["embed_tokens.weight"] = q16(emb)
    del emb

    # ---- calibration documents (sink token first), packed
    lens = list(calib_lens)
    ids = torch.cat([torch.cat([torch.tensor([sink], device=device), torch.randint(1000, V - 1000, (n - 1,), generator=gen, device=device)]) for n in lens])
    pos = torch.cat([torch.arange(n, device=device) for n in lens])
    cos_t, sin_t = rope_tables(cfg)
    cos, sin = cos_t.to(device)[pos], sin_t.to(device)[pos]
    T = ids.numel()
    doc = torch.repeat_interleave(torch.arange(len(lens), device=device), torch.tensor(lens, device=device))
    allowed = (doc[:, None] == doc[None, :]) & (pos[:, None] >= pos[None, :])          # [T, T] causal within a document
    n_keys = allowed.sum(-1)
    x = sd["embed_tokens.weight"].float()[ids]
    typ0 = x.abs().median().item()
    stats = {"profile": "trained_like", "seed": seed, "massive_channels": mass_ch.tolist(), "layers": []}

    # heavy (slowly rotating) rotary pairs of a head: the last quarter of the d/2 frequencies, both halves of each pair
    heavy = torch.zeros(d, dtype=torch.bool, device=device)
    j0 = d // 2 - d // 8
    heavy[j0:d // 2] = True
    heavy[d // 2 + j0:] = True

    def qk_bias(n_heads):
        b = rn(n_heads, d)
        b[:, heavy] *= bias_sigma
        for h in range(n_heads):                                              # a few entries beyond 100 per projection
            if h % max(1, n_heads // 3) == 0:
                jj = j0 + int(torch.randint(0, d // 8, (1,), generator=gen, device=device))
                b[h, jj] = float(ru(100.0, 300.0, 1)) * (1.0 if h % 2 else -1.0)
        return b

    def logits_of(q, k):
        """[nq, T, T] fp32 logits (causal pairs only are meaningful)"""
        qh = _rope(q.view(T, nq, d), cos, sin).permute(1, 0, 2)
        kh = _rope(k.view(T, nkv, d), cos, sin).permute(1, 0, 2).repeat_interleave(grp, 0)
        return qh @ kh.transpose(1, 2) / math.sqrt(d), qh, kh

    def spread(lg):
        """standard deviation of the logits across the keys of a query (per-query mean removed: a constant cancels in softmax)"""
        m = allowed[None].float()
        mean = (lg * m).sum(-1, keepdim=True) / n_keys[None, :, None]
        var = (((lg - mean) * m) ** 2).sum(-1) / n_keys[None, :].clamp(min=2)
        return var[:, n_keys >= 8].mean().sqrt().item()

    for li in range(cfg.num_layers):
        p = f"layers.{li}."
        typ = x.abs().median().item()                                          # the typical channel of the stream entering this layer
        g1 = norm_weight(squash=True)
        h = _rms(x, cfg.rms_eps) * g1.float()
        wq, wk, wv = rn(nq * d, H) / math.sqrt(H), rn(nkv * d, H) / math.sqrt(H), rn(nkv * d, H) / math.sqrt(H)
        bq = bk = bv = None
        if cfg.qkv_bias:
            hv = heavy.repeat(nq)
            wq[hv] *= 0.1                                                      # heavy dims are bias-dominated (near-constant "positional" dims)
            wk[heavy.repeat(nkv)] *= 0.1
            bq, bk = q16(qk_bias(nq).reshape(-1)), q16(qk_bias(nkv).reshape(-1))
            bv = rn(nkv * d)
            vh = torch.rand(nkv * d, generator=gen, device=device) < 0.25
            bv[vh] *= bias_sigma
            bv = q16(bv)
        # q / k scale: bisection on the measured spread of the logits (weights only are scaled; the biases stay).  CONTENT part first --
        # the bilinear q_w . k_w term is what turns a perturbation of the stream into a perturbation of the attention pattern; a trained
        # model is not chaotic (its bf16 run stays within 1e-3 .. 1e-2 of its fp32 run), so this part gets sigma 1.5-3 ...
        target = float(ru(logit_sigma[0], logit_sigma[1], 1))
        c_target = float(ru(content_sigma[0], content_sigma[1], 1))
        c_all, c_sink = h[pos > 0].mean(0), h[pos == 0].mean(0)
        r = c_sink - (c_sink @ c_all) / (c_all @ c_all) * c_all
        e_q, e_k = c_all / (c_all @ c_all), r / (c_sink @ r)
        cq = (h @ e_q)[:, None]                                               # ~1 on every token
        if bq is not None:
            bqf, bkf = bq.float()[None, :], bk.float()[None, :]
            pq = pk = None
        else:                                                                  # Llama has no bias parameters: a trained model builds the same constant q / k
            pq, pk = (qk_bias(nq) * (pseudo_bias_sigma / bias_sigma)).reshape(-1), (qk_bias(nkv) * (pseudo_bias_sigma / bias_sigma)).reshape(-1)
            pq, pk = pq.clamp(-8 * pseudo_bias_sigma, 8 * pseudo_bias_sigma), pk.clamp(-8 * pseudo_bias_sigma, 8 * pseudo_bias_sigma)
            bqf, bkf = cq * pq[None, :], cq * pk[None, :]                      # components out of the massive channels ("massive activations act as biases")
        hq, hk = h @ wq.T, h @ wk.T
        lg0, _, _ = logits_of(bqf.expand(T, -1), bkf.expand(T, -1))            # the part of the logits that does not depend on the tokens' content
        lo_s, hi_s = 0.02, 60.0
        for _ in range(18):
            s = math.sqrt(lo_s * hi_s)
            lg, _, _ = logits_of(s * hq + bqf, s * hk + bkf)
            lo_s, hi_s = (s, hi_s) if spread(lg - lg0) < c_target else (lo_s, s)
        s = math.sqrt(lo_s * hi_s)
        wq, wk = wq * s, wk * s
        if pq is not None:
            wq += pq[:, None] * e_q[None, :]
            wk += pk[:, None] * e_q[None, :]
            bqf = bkf = 0.0
        # ... and the rest of the spread is POSITIONAL, as in trained heads (previous-token / local heads, attention sink): rank-one terms
        # on the direction common to every token's normalised hidden state (the massive channels provide it), so that every q and every k
        # carries the same large component on a band of rotary pairs: sum_p a_p^2 cos(w_p (t - j)), a recency kernel peaked at j = t.
        # attention sink: in half of the kv heads every query carries a common component u (on the slowly rotating dims, so that it survives
        # RoPE at any distance) that only the FIRST token's key answers -- a rank-one term on the direction that separates the sink token's
        # normalised hidden state from the other tokens'.
        amp = math.sqrt(sink_logit * target * math.sqrt(d))
        for kvh in range(0, nkv, 2):
            u = torch.zeros(d, device=device)
            u[heavy] = rn(int(heavy.sum()))
            qb = bq.float() if bq is not None else pq
            if qb is not None:                                                 # ... orthogonal to the group's q biases: the sink key must not also collect b_q . u
                B = qb.view(nq, d)[kvh * grp:(kvh + 1) * grp] * heavy[None, :]
                u -= torch.linalg.lstsq(B.T, u[:, None]).solution[:, 0] @ B
            u /= u.norm()
            wk[kvh * d:(kvh + 1) * d] += amp * u[:, None] * e_k[None, :]
            for qh_ in range(kvh * grp, (kvh + 1) * grp):
                wq[qh_ * d:(qh_ + 1) * d] += amp * u[:, None] * e_q[None, :]
        band = torch.zeros(nkv, d, device=device)
        band[1::2, d // 16:d // 16 + d // 4] = 0.5 + torch.rand(nkv // 2, d // 4, generator=gen, device=device)   # odd kv heads; mid-frequency pairs (first halves: phase 0)
        if nkv == 1:
            band[0, d // 16:d // 16 + d // 4] = 1.0
        band /= band.norm(dim=-1, keepdim=True).clamp(min=1e-20)
        hq, hk = h @ wq.T + bqf, h @ wk.T + bkf
        lo_a, hi_a = 0.0, 80.0
        for _ in range(16):
            a_ = 0.5 * (lo_a + hi_a)
            lg, _, _ = logits_of(hq + a_ * cq * band.repeat_interleave(grp, 0).reshape(1, -1), hk + a_ * cq * band.reshape(1, -1))
            lo_a, hi_a = (a_, hi_a) if spread(lg) < target else (lo_a, a_)
        a_ = 0.5 * (lo_a + hi_a)
        wk += a_ * band.reshape(-1)[:, None] * e_q[None, :]
        wq += a_ * band.repeat_interleave(grp, 0).reshape(-1)[:, None] * e_q[None, :]
        wq, wk, wv = q16(wq), q16(wk), q16(wv * 1.5)
        q = h @ wq.float().T + (bq.float() if bq is not None else 0.0)
        k = h @ wk.float().T + (bk.float() if bk is not None else 0.0)
        v = h @ wv.float().T + (bv.float() if bv is not None else 0.0)
        lg, qh, kh = logits_of(q, k)
        sp = spread(lg)
        pr = torch.softmax(lg.masked_fill(~allowed[None], float("-inf")), -1)
        vh_ = v.view(T, nkv, d).permute(1, 0, 2).repeat_interleave(grp, 0)
        att = (pr @ vh_).permute(1, 0, 2).reshape(T, nq * d)
        wo = rn(H, nq * d) / math.sqrt(nq * d)
        ao = att @ wo.T
        wo = q16(wo * (attn_add * typ / max(ao.abs().median().item(), 1e-20)))
        x = x + att @ wo.float().T
        g2 = norm_weight(squash=True)
        h2 = _rms(x, cfg.rms_eps) * g2.float()
        wg, wu = rn(I, H) / math.sqrt(H), rn(I, H) / math.sqrt(H)
        wg = q16(wg * (1.5 / max((h2 @ wg.T).std().item(), 1e-20)))           # gate pre-activations with sigma 1.5: part of them saturate
        wu = q16(wu * (1.0 / max((h2 @ wu.T).std().item(), 1e-20)))
        act = torch.nn.functional.silu(h2 @ wg.float().T) * (h2 @ wu.float().T)
        wd = rn(H, I) / math.sqrt(I)
        mo = act @ wd.T
        wd *= mlp_add * typ / max(mo.abs().median().item(), 1e-20)
        if li < 3:                                                             # the early MLPs write the massive channels (Sun et al. 2024)
            wd[mass_ch] *= 6.0
        wd = q16(wd)
        x = x + act @ wd.float().T
        sd[p + "self_attn.q_proj.weight"], sd[p + "self_attn.k_proj.weight"], sd[p + "self_attn.v_proj.weight"] = wq, wk, wv
        if cfg.qkv_bias:
            sd[p + "self_attn.q_proj.bias"], sd[p + "self_attn.k_proj.bias"], sd[p + "self_attn.v_proj.bias"] = bq, bk, bv
        sd[p + "self_attn.o_proj.weight"] = wo
        sd[p + "mlp.gate_proj.weight"], sd[p + "mlp.up_proj.weight"], sd[p + "mlp.down_proj.weight"] = wg, wu, wd
        sd[p + "input_layernorm.weight"], sd[p + "post_attention_layernorm.weight"] = g1, g2
        not_first = pos > 0
        stats["layers"].append({
            "logit_sigma": round(sp, 3), "top1_prob": round(pr.max(-1).values[:, n_keys >= 8].mean().item(), 4),
            "content_sigma": round(c_target, 3), "qk_weight_scale": round(s, 4), "sink_mass": round(pr[:, not_first & (n_keys >= 8), :][..., pos == 0].sum(-1).mean().item(), 4),
            "max_abs_q": round(qh.abs().max().item(), 2), "max_abs_k": round(kh.abs().max().item(), 2), "max_abs_v": round(v.abs().max().item(), 2),
            "max_abs_bias": round(max(bq.float().abs().max().item(), bk.float().abs().max().item(), bv.float().abs().max().item()), 1) if bq is not None else 0.0,
            "stream_max_over_median": round(x.abs().max().item() / max(x.abs().median().item(), 1e-20), 1),
            "stream_max": round(x.abs().max().item(), 2)})
    gf = norm_weight(squash=True)
    sd["norm.weight"] = gf
    L = stats["layers"]
    stats["summary"] = {
        "logit_sigma_min": min(l["logit_sigma"] for l in L), "logit_sigma_max": max(l["logit_sigma"] for l in L),
        "top1_prob_mean": round(sum(l["top1_prob"] for l in L) / len(L), 4), "sink_mass_mean": round(sum(l["sink_mass"] for l in L) / len(L), 4),
        "max_abs_qkv": max(max(l["max_abs_q"], l["max_abs_k"], l["max_abs_v"]) for l in L), "max_abs_bias": max(l["max_abs_bias"] for l in L),
        "stream_max_over_median_max": max(l["stream_max_over_median"] for l in L), "stream_max": max(l["stream_max"] for l in L),
        "embedding_typical": round(typ0, 5)}
    return sd, stats


# ---------------------------------------------------------------------------------------------------------------------------------------
# Synthetic CORPORA for the search legs.  iid Gaussian rows are the easy case for a sample-thresholded filter; embedding corpora are
# clustered (topics), carry exact duplicates, and are often stored grouped by source.
# ---------------------------------------------------------------------------------------------------------------------------------------
def clustered_corpus(slot: torch.Tensor, n_clusters: int = 1000, intra_cos: float = 0.9, dup_frac: float = 0.01, seed: int = 0,
                     order: str = "shuffled", chunk_rows: int = 65536) -> dict:
    """Fills `slot` ([n, d] fp32 view of an index shard, on the GPU) with unit rows x = sqrt(c) centre + sqrt(1 - c) u (u a random unit
    vector): von-Mises-Fisher-like clusters whose members have cosine ~c = intra_cos to each other; dup_frac of the rows are exact copies
    of another row.  order: "shuffled" (cluster membership iid over the rows) or "by_cluster" (the rows of a cluster are contiguous -- a
    corpus stored by topic / source: a strided block sample misses whole clusters).  -> {"centres": [n_clusters, d], "assign": [n] int64}"""
    n, d = slot.shape
    dev = slot.device
    gen = torch.Generator(device=dev).manual_seed(seed)
    centres = torch.nn.functional.normalize(torch.randn(n_clusters, d, generator=gen, device=dev), dim=-1)
    if order == "by_cluster":
        assign = (torch.arange(n, device=dev) * n_clusters // n).to(torch.int64)
    elif order == "shuffled":
        assign = torch.randint(0, n_clusters, (n,), generator=gen, device=dev)
    else:
        raise ValueError(order)
    a, b = math.sqrt(intra_cos), math.sqrt(1.0 - intra_cos)
    for s in range(0, n, chunk_rows):
        e = min(s + chunk_rows, n)
        u = torch.nn.functional.normalize(torch.randn(e - s, d, generator=gen, device=dev), dim=-1)
        x = torch.nn.functional.normalize(a * centres[assign[s:e]] + b * u, dim=-1)
        n_dup = int(dup_frac * (e - s))
        if n_dup:
            dst = torch.randperm(e - s, generator=gen, device=dev)[:n_dup]
            src = torch.randint(0, e - s, (n_dup,), generator=gen, device=dev)
            x[dst] = x[src]                                                   # exact copies (ties: the lower row wins)
            assign[s:e][dst] = assign[s:e][src]
        slot[s:e] = x
    return {"centres": centres, "assign": assign}


def cluster_queries(centres: torch.Tensor, n_queries: int, query_cos: float = 0.9, seed: int = 1) -> torch.Tensor:
    """Unit queries near randomly chosen cluster centres (cosine ~query_cos to the centre)."""
    dev = centres.device
    gen = torch.Generator(device=dev).manual_seed(seed)
    c = centres[torch.randint(0, centres.shape[0], (n_queries,), generator=gen, device=dev)]
    u = torch.nn.functional.normalize(torch.randn(n_queries, centres.shape[1], generator=gen, device=dev), dim=-1)
    return torch.nn.functional.normalize(math.sqrt(query_cos) * c + math.sqrt(1.0 - query_cos) * u, dim=-1)
