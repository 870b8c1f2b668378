"""Document-side encoder on MI355X: weight packing + the lrx_encode_* entry points.

Host-side mirror of what the reference does around its LM call: load weights (finetune/modeling_encoder.py:602-633, LoRA
merge :616-625 restated as W + (alpha/r) B A), build the RoPE table (transformers modeling_rope_utils: default and
llama3 scaling), then run HybridModel.encode_passage's dense branch (finetune/modeling_hybrid.py:205-278) -- here one
call into liblrx.so on a packed (ids, cu_seqlens) batch."""
from __future__ import annotations

import os

import ctypes as C
import math
from dataclasses import dataclass
from typing import Optional

import torch

from . import _lib


# When the fp32 residual stream is switched on by default: ALWAYS, since round 5 (threshold 0; 60 000 in round 3, 40 000 in round 4).
# Measured at full depth against the HF fp32 model, Gaussian weights (tests/test_gpu_encoder.py::test_full_depth_hf_parity,
# profiles/r03_full_depth_parity.jsonl), 1 - cos on the bf16 stream grows like ~9e-9 x layers x hidden_size (1B 2.8e-4 ... 8B 1.28e-3); on
# TRAINED-LIKE weights (synth.py; 64 documents x 3 seeds, profiles/r04_trained_like_parity.jsonl) the 16-layer Llama-3.2-1B -- the last
# backbone left on the bf16 stream -- read 0.7e-3 .. 2.0e-3 there against 2.1e-4 .. 5.4e-4 on the precise stream: the bf16 stream misses
# the 1e-3 bar of BASELINE.json's north_star on the headline model, so no default configuration runs it any more (price: ~3 % of the 1B's
# docs/s, bench.py -> configs.encode_llama32_1b_bf16_stream).  `EncoderConfig(precise_stream=False)` still selects it explicitly.
PRECISE_FROM_LAYERS_X_HIDDEN = 0
# The fp32 stream's GEMM operands (round 6).  With bf16 operands the worst of 2 048 trained-like documents reads 4.6e-4 on Llama-3.2-1B, 8.1e-4
# (MRL-256: 1.0e-3) on Llama-3.2-3B and 1.11e-3 on Llama-3.1-8B -- 3 documents over the 1e-3 bar (profiles/r06_trained_like_tail_2048.jsonl) --
# and 70 % of that distance is ONE rounding: the QKV projection's A operand bf16(x * gamma) (tools/exp/rounding_fp16_o_act.py).  Default
# "fp16_qkv": that operand and wqkv in fp16 (f16 MFMA for the QKV projection only): 1B 9.7e-5, 3B 3.2e-4, 8B 2.4e-4 on the same samples for
# -0.3 % docs/s (same box, alternating runs).  "fp16": every projection's operands (activations, attention / SwiGLU outputs, the four weight
# matrices): 1-3e-5 for -3.5 ... -3.9 % (the f16 multiplier array draws more power at the package cap) -- on request, for callers that need the
# fp32 model's embeddings to five digits.  "bf16": round 5's arithmetic.


@dataclass
class EncoderConfig:
    vocab_size: int
    hidden_size: int
    num_layers: int
    num_q_heads: int
    num_kv_heads: int
    head_dim: int
    intermediate_size: int
    rms_eps: float = 1e-5
    rope_theta: float = 500000.0
    rope_type: str = "default"
    rope_factor: float = 32.0
    rope_low_freq_factor: float = 1.0
    rope_high_freq_factor: float = 4.0
    rope_original_max_position: int = 8192
    qkv_bias: bool = False
    max_positions: int = 512
    fold_norm: bool = True      # RMSNorm weights folded into the next projection at load time (lrx_encoder_config.norm_folded)
    # fp32 residual stream + exact weights, the norm weight on the bf16 activation operand (lrx_encoder_config.precise_stream): None = the
    # default (PRECISE_FROM_LAYERS_X_HIDDEN = 0: on for every backbone -- the bf16 stream and the folded weights spend the 1e-3 cosine budget)
    precise_stream: Optional[bool] = None
    # GEMM operands of the fp32-stream pipeline (lrx_encoder_config.precise_stream = 1 / 2): "bf16", or "fp16" -- activations fp16(x * gamma),
    # attention / SwiGLU outputs and the four projection weights in fp16 (converted once at load: exact for every |w| in [6.1e-5, 65504]), the
    # f16 MFMA at the bf16 rate.  Three more mantissa bits on every operand: 1 - cos against the HF fp32 model drops 14-42 x on trained-like
    # weights (tools/exp/rounding_fp16_o_act.py: 8B 3.4e-4 -> 8e-6); the range is fp16's, watched by the saturation counter the product path
    # reads.  "fp16_qkv" (precise_stream = 3): the QKV projection only -- the default (None) of the fp32 stream, unless a weight leaves fp16's
    # range (then "bf16", with a warning); the bf16 stream is bf16.
    operand_dtype: Optional[str] = None

    def operand_mode(self) -> str:
        """"bf16" | "fp16" (every projection) | "fp16_qkv" (the QKV projection only: its bf16 operand is 70 % of the pipeline's distance to the fp32
        model -- 8B 3.4e-4 -> 1.0e-4, 1B 1.1e-4 -> 2.4e-5 in the restatement -- and its fp16 form costs < 0.2 % of a step)."""
        if not self.use_precise_stream():
            return "bf16"
        if self.operand_dtype is not None:
            return self.operand_dtype
        return "fp16_qkv"

    def use_f16_operands(self) -> bool:
        return self.operand_mode() == "fp16"

    def use_precise_stream(self) -> bool:
        if self.precise_stream is not None:
            return bool(self.precise_stream)
        return self.num_layers * self.hidden_size >= PRECISE_FROM_LAYERS_X_HIDDEN

    @staticmethod
    def llama32_1b(max_positions: int = 512) -> "EncoderConfig":
        return EncoderConfig(128256, 2048, 16, 32, 8, 64, 8192, 1e-5, 500000.0, "llama3", 32.0, 1.0, 4.0, 8192, False, max_positions)

    @staticmethod
    def llama31_8b(max_positions: int = 512) -> "EncoderConfig":
        return EncoderConfig(128256, 4096, 32, 32, 8, 128, 14336, 1e-5, 500000.0, "llama3", 8.0, 1.0, 4.0, 8192, False, max_positions)

    @staticmethod
    def llama32_3b(max_positions: int = 512) -> "EncoderConfig":
        """Llama-3.2-3B (backbone of lightretriever-llama3.2-3b): 24 q / 8 kv heads of 128, tied embeddings."""
        return EncoderConfig(128256, 3072, 28, 24, 8, 128, 8192, 1e-5, 500000.0, "llama3", 32.0, 1.0, 4.0, 8192, False, max_positions)

    @staticmethod
    def qwen25_1_5b(max_positions: int = 512) -> "EncoderConfig":
        """Qwen2.5-1.5B: the backbone of the released lightretriever-qwen2.5-1.5b adapters (scripts/*_infer.ipynb)."""
        return EncoderConfig(151936, 1536, 28, 12, 2, 128, 8960, 1e-6, 1000000.0, "default", 1.0, 1.0, 4.0, 8192, True, max_positions)

    @staticmethod
    def qwen25_3b(max_positions: int = 512) -> "EncoderConfig":
        """Qwen2.5-3B (backbone of lightretriever-qwen2.5-3b): 16 q / 2 kv heads of 128 (GQA group 8)."""
        return EncoderConfig(151936, 2048, 36, 16, 2, 128, 11008, 1e-6, 1000000.0, "default", 1.0, 1.0, 4.0, 8192, True, max_positions)

    @staticmethod
    def qwen25_7b(max_positions: int = 512) -> "EncoderConfig":
        return EncoderConfig(152064, 3584, 28, 28, 4, 128, 18944, 1e-6, 1000000.0, "default", 1.0, 1.0, 4.0, 8192, True, max_positions)

    @staticmethod
    def from_hf_dict(c: dict, max_positions: int = 512) -> "EncoderConfig":
        """Fields of an HF Llama/Qwen2 config.json (transformers 4.x `rope_scaling` or 5.x `rope_parameters`)."""
        rp = c.get("rope_parameters") or c.get("rope_scaling") or {}
        rt = rp.get("rope_type", rp.get("type", "default")) or "default"
        nq = c["num_attention_heads"]
        return EncoderConfig(
            vocab_size=c["vocab_size"], hidden_size=c["hidden_size"], num_layers=c["num_hidden_layers"], num_q_heads=nq,
            num_kv_heads=c.get("num_key_value_heads", nq), head_dim=c.get("head_dim") or c["hidden_size"] // nq,
            intermediate_size=c["intermediate_size"], rms_eps=c.get("rms_norm_eps", 1e-5),
            rope_theta=float(rp.get("rope_theta", c.get("rope_theta", 10000.0))), rope_type=rt,
            rope_factor=float(rp.get("factor", 1.0)), rope_low_freq_factor=float(rp.get("low_freq_factor", 1.0)),
            rope_high_freq_factor=float(rp.get("high_freq_factor", 4.0)),
            rope_original_max_position=int(rp.get("original_max_position_embeddings", 8192)),
            qkv_bias=c.get("model_type", "") == "qwen2" or bool(c.get("attention_bias", False)), max_positions=max_positions)

    def flops_per_doc(self, seq_len: int) -> float:
        """Algorithmic bf16 FLOPs per document, F(S) = 2 S P_ne + 2 L S^2 H."""
        d = self.head_dim
        p_layer = (self.hidden_size * (self.num_q_heads + 2 * self.num_kv_heads) * d + self.num_q_heads * d * self.hidden_size
                   + 3 * self.hidden_size * self.intermediate_size + 2 * self.hidden_size)
        p_ne = self.num_layers * p_layer + self.hidden_size
        return 2.0 * seq_len * p_ne + 2.0 * self.num_layers * seq_len * seq_len * self.hidden_size


def rope_tables(cfg: EncoderConfig) -> tuple[torch.Tensor, torch.Tensor]:
    """cos/sin [max_positions, d/2] fp32 as LlamaRotaryEmbedding computes them (fp32 position * inv_freq; default and llama3 scaling).  HF's
    bf16 run rounds them to bf16 afterwards; the fused QKV epilogue rotates the fp32 accumulators with the fp32 values (round 3)."""
    d = cfg.head_dim
    inv = 1.0 / (torch.tensor(cfg.rope_theta, dtype=torch.float32) ** (torch.arange(0, d, 2, dtype=torch.int64).float() / d))
    if cfg.rope_type == "llama3":
        old = cfg.rope_original_max_position
        low_wl, high_wl = old / cfg.rope_low_freq_factor, old / cfg.rope_high_freq_factor
        wavelen = 2 * math.pi / inv
        inv_l = torch.where(wavelen > low_wl, inv / cfg.rope_factor, inv)
        smooth = (old / wavelen - cfg.rope_low_freq_factor) / (cfg.rope_high_freq_factor - cfg.rope_low_freq_factor)
        smoothed = (1 - smooth) * inv_l / cfg.rope_factor + smooth * inv_l
        medium = ~(wavelen < high_wl) & ~(wavelen > low_wl)
        inv = torch.where(medium, smoothed, inv_l)
    elif cfg.rope_type != "default":
        raise NotImplementedError(f"rope_type {cfg.rope_type}")
    pos = torch.arange(cfg.max_positions, dtype=torch.float32)
    freqs = pos[:, None] * inv[None, :].float()
    return freqs.cos().contiguous(), freqs.sin().contiguous()


def interleave_gate_up(gate: torch.Tensor, up: torch.Tensor) -> torch.Tensor:
    """[I,H],[I,H] -> [2I,H] with 16-row groups alternating gate/up (the layout lrx_gemm's SwiGLU epilogue expects:
    rows [32j, 32j+16) = gate[16j..], rows [32j+16, 32j+32) = up[16j..])."""
    I, H = gate.shape
    assert I % 16 == 0
    return torch.stack([gate.view(I // 16, 16, H), up.view(I // 16, 16, H)], dim=1).reshape(2 * I, H).contiguous()


def lora_merge(W: torch.Tensor, A: torch.Tensor, B: torch.Tensor, alpha: float, r: int) -> torch.Tensor:
    """peft merge_and_unload for one Linear: W + (alpha / r) * B @ A   (finetune/modeling_encoder.py:616-625)."""
    return (W.float() + (alpha / r) * (B.float() @ A.float())).to(W.dtype)


class LrxEncoder:
    """Weights resident in HBM as bf16 + a reusable workspace; encode_packed() is the B3 operator of SURVEY.md 8b."""

    def __init__(self, cfg: EncoderConfig, state_dict: dict, device: Optional[torch.device] = None):
        """state_dict: HF names without the leading `model.` (embed_tokens.weight, layers.N.self_attn.q_proj.weight, ...);
        values are torch tensors or numpy arrays (any float dtype) -- cast to bf16 and packed here."""
        _lib.require_gpu()
        self.lib = _lib.lib()
        self.cfg = cfg
        self.device = device or torch.device("cuda", torch.cuda.current_device())
        bf = torch.bfloat16

        def g(name):
            v = state_dict[name]
            if not isinstance(v, torch.Tensor):
                v = torch.from_numpy(v)
            return v

        def dev(t):
            return t.to(device=self.device, dtype=bf).contiguous()

        self.embed = dev(g("embed_tokens.weight"))
        self.final_norm = dev(g("norm.weight"))
        cos, sin = rope_tables(cfg)
        self.rope_cos, self.rope_sin = cos.to(self.device), sin.to(self.device)
        from .ops import rotary_pair_order
        perm = rotary_pair_order(cfg.num_q_heads, cfg.num_kv_heads, cfg.head_dim).to(self.device)
        self.precise = cfg.use_precise_stream()
        self.layers = []
        for i in range(cfg.num_layers):
            p = f"layers.{i}."
            wqkv = torch.cat([g(p + "self_attn.q_proj.weight"), g(p + "self_attn.k_proj.weight"), g(p + "self_attn.v_proj.weight")], 0)
            bqkv = None
            if cfg.qkv_bias:
                bqkv = dev(torch.cat([g(p + "self_attn.q_proj.bias"), g(p + "self_attn.k_proj.bias"), g(p + "self_attn.v_proj.bias")], 0))
            self.layers.append(dict(
                wqkv=dev(wqkv), bqkv=bqkv, wo=dev(g(p + "self_attn.o_proj.weight")),
                wgu=dev(interleave_gate_up(g(p + "mlp.gate_proj.weight"), g(p + "mlp.up_proj.weight"))),
                wdown=dev(g(p + "mlp.down_proj.weight")), ln1=dev(g(p + "input_layernorm.weight")),
                ln2=dev(g(p + "post_attention_layernorm.weight"))))
        # What the C structs get (the logical-order originals stay for hf_state_dict / inspection): wqkv / bqkv with the q and k heads' rows in
        # rotary-pair order; with the folded norm (and no precise stream) W' = W diag(gamma): fp32 product, one rounding to bf16.
        fold = cfg.fold_norm and not self.precise
        for L in self.layers:
            wq = L["wqkv"].float() * L["ln1"].float()[None, :] if fold else L["wqkv"]
            L["wqkv_c"] = wq[perm].to(bf).contiguous()
            L["bqkv_c"] = L["bqkv"][perm].contiguous() if L["bqkv"] is not None else None
            L["wgu_c"] = (L["wgu"].float() * L["ln2"].float()[None, :]).to(bf).contiguous() if fold else L["wgu"]
        # fp16 operands (fp32 stream only): the four projection matrices as fp16 -- checked once that nothing left fp16's range
        if cfg.operand_dtype not in (None, "bf16", "fp16", "fp16_qkv"):
            raise ValueError(f"operand_dtype {cfg.operand_dtype!r}: 'bf16', 'fp16' or 'fp16_qkv'")
        if cfg.operand_dtype in ("fp16", "fp16_qkv") and not self.precise:
            raise ValueError(f"operand_dtype={cfg.operand_dtype!r} belongs to the fp32 residual stream (precise_stream)")
        self.operand_mode = cfg.operand_mode()
        for L in self.layers:
            L["wo_c"], L["wdown_c"] = L["wo"], L["wdown"]
        if self.operand_mode != "bf16":
            h = torch.float16
            bad = torch.zeros((), dtype=torch.bool, device=self.device)
            keys = (("wqkv_c", "wqkv_c"), ("wgu_c", "wgu_c"), ("wo_c", "wo"), ("wdown_c", "wdown")) if self.operand_mode == "fp16" else (("wqkv_c", "wqkv_c"),)
            for L in self.layers:
                for key, src in keys:
                    w32 = L[src].float()
                    L[key] = L[src].to(h).contiguous()
                    # fp16 must hold the matrix: nothing beyond +-65504, and nothing that matters below its subnormals (a checkpoint whose
                    # scale lives in the norm weights instead of the projection): the conversion may move the matrix by 1e-3 of its norm at most
                    bad |= ~torch.isfinite(L[key]).all() | ((L[key].float() - w32).norm() > 1e-3 * w32.norm())
                    del w32
            if bool(bad):
                if cfg.operand_dtype is not None:
                    raise ValueError(f"operand_dtype={cfg.operand_dtype!r}: a projection matrix does not survive the conversion to fp16 "
                                     "(|w| > 65504, non-finite, or a scale below fp16's subnormals)")
                import warnings
                warnings.warn("a projection matrix does not survive the conversion to fp16 (range): the encoder keeps bf16 GEMM operands", RuntimeWarning)
                self.operand_mode = "bf16"
                for L in self.layers:
                    L["wqkv_c"], L["wgu_c"], L["wo_c"], L["wdown_c"] = L["wqkv"][perm].contiguous(), L["wgu"], L["wo"], L["wdown"]
        self.operand_f16 = self.operand_mode == "fp16"
        # LM head for the sparse branch: tied to the embedding unless the checkpoint carries its own (`lm_head.weight`)
        self.lm_head = dev(g("lm_head.weight")) if "lm_head.weight" in state_dict else None
        self._build_c_structs()
        self._ws = None

    @classmethod
    def random_init(cls, cfg: EncoderConfig, seed: int = 0, std: float = 0.02, device: Optional[torch.device] = None,
                    profile: str = "gaussian", **synth_kwargs) -> "LrxEncoder":
        """Random weights of the real architecture generated directly on the GPU (benchmarks: no checkpoints offline).  profile "gaussian":
        every matrix N(0, std); "trained_like": synth.trained_like_state_dict (peaky attention, massive-activation channels, attention
        sink, heavy-tailed norm weights, Qwen-scale q/k/v biases) -- the statistics it measured while calibrating are kept as `synth_stats`."""
        _lib.require_gpu()
        device = device or torch.device("cuda", torch.cuda.current_device())
        if profile == "trained_like":
            from .synth import trained_like_state_dict
            sd, stats = trained_like_state_dict(cfg, seed=seed, device=device, **synth_kwargs)
            enc = cls(cfg, sd, device)
            enc.synth_stats = stats
            return enc
        if profile != "gaussian":
            raise ValueError(f"unknown weight profile {profile!r}")
        gen = torch.Generator(device=device).manual_seed(seed)
        H, d, I = cfg.hidden_size, cfg.head_dim, cfg.intermediate_size

        def rn(*shape, s=std):
            return (torch.randn(*shape, generator=gen, device=device, dtype=torch.float32) * s).to(torch.bfloat16)

        sd = {"embed_tokens.weight": rn(cfg.vocab_size, H), "norm.weight": (1.0 + rn(H)).to(torch.bfloat16)}
        for i in range(cfg.num_layers):
            p = f"layers.{i}."
            sd[p + "self_attn.q_proj.weight"] = rn(cfg.num_q_heads * d, H)
            sd[p + "self_attn.k_proj.weight"] = rn(cfg.num_kv_heads * d, H)
            sd[p + "self_attn.v_proj.weight"] = rn(cfg.num_kv_heads * d, H)
            sd[p + "self_attn.o_proj.weight"] = rn(H, cfg.num_q_heads * d)
            sd[p + "mlp.gate_proj.weight"] = rn(I, H)
            sd[p + "mlp.up_proj.weight"] = rn(I, H)
            sd[p + "mlp.down_proj.weight"] = rn(H, I)
            sd[p + "input_layernorm.weight"] = (1.0 + rn(H)).to(torch.bfloat16)
            sd[p + "post_attention_layernorm.weight"] = (1.0 + rn(H)).to(torch.bfloat16)
            if cfg.qkv_bias:
                sd[p + "self_attn.q_proj.bias"] = rn(cfg.num_q_heads * d)
                sd[p + "self_attn.k_proj.bias"] = rn(cfg.num_kv_heads * d)
                sd[p + "self_attn.v_proj.bias"] = rn(cfg.num_kv_heads * d)
        return cls(cfg, sd, device)

    def hf_state_dict(self) -> dict:
        """The packed weights mapped back to HF parameter names (un-fused q/k/v, de-interleaved gate/up): lets tests load
        the very same values into a transformers model."""
        c = self.cfg
        H, d, I = c.hidden_size, c.head_dim, c.intermediate_size
        nq, nkv = c.num_q_heads * d, c.num_kv_heads * d
        sd = {"embed_tokens.weight": self.embed, "norm.weight": self.final_norm}
        for i, L in enumerate(self.layers):
            p = f"layers.{i}."
            sd[p + "self_attn.q_proj.weight"], sd[p + "self_attn.k_proj.weight"], sd[p + "self_attn.v_proj.weight"] = \
                L["wqkv"][:nq], L["wqkv"][nq:nq + nkv], L["wqkv"][nq + nkv:]
            if L["bqkv"] is not None:
                sd[p + "self_attn.q_proj.bias"], sd[p + "self_attn.k_proj.bias"], sd[p + "self_attn.v_proj.bias"] = \
                    L["bqkv"][:nq], L["bqkv"][nq:nq + nkv], L["bqkv"][nq + nkv:]
            sd[p + "self_attn.o_proj.weight"] = L["wo"]
            gu = L["wgu"].view(I // 16, 2, 16, H)
            sd[p + "mlp.gate_proj.weight"], sd[p + "mlp.up_proj.weight"] = gu[:, 0].reshape(I, H), gu[:, 1].reshape(I, H)
            sd[p + "mlp.down_proj.weight"] = L["wdown"]
            sd[p + "input_layernorm.weight"], sd[p + "post_attention_layernorm.weight"] = L["ln1"], L["ln2"]
        return sd

    def _build_c_structs(self):
        c = self.cfg
        self._ccfg = _lib.EncoderConfigC(c.vocab_size, c.hidden_size, c.num_layers, c.num_q_heads, c.num_kv_heads, c.head_dim,
                                         c.intermediate_size, c.rms_eps, int(c.qkv_bias), c.max_positions, int(c.fold_norm and not self.precise),
                                         {"fp16": 2, "fp16_qkv": 3}.get(self.operand_mode, int(self.precise)))
        arr = (_lib.LayerWeightsC * c.num_layers)()
        for i, L in enumerate(self.layers):
            arr[i] = _lib.LayerWeightsC(L["wqkv_c"].data_ptr(), L["bqkv_c"].data_ptr() if L["bqkv_c"] is not None else None,
                                        L["wo_c"].data_ptr(), L["wgu_c"].data_ptr(), L["wdown_c"].data_ptr(), L["ln1"].data_ptr(),
                                        L["ln2"].data_ptr())
        self._clayers = arr
        self._cw = _lib.EncoderWeightsC(self.embed.data_ptr(), self.final_norm.data_ptr(), self.rope_cos.data_ptr(),
                                        self.rope_sin.data_ptr(), C.cast(arr, C.POINTER(_lib.LayerWeightsC)))
        self._handle = None        # (a handle built before this call points at the structs just replaced)

    def workspace_bytes(self, total_tokens: int, n_seqs: int) -> int:
        return int(self.lib.lrx_encode_workspace_bytes(C.byref(self._ccfg), total_tokens, n_seqs))

    def _workspace(self, total_tokens: int, n_seqs: int) -> torch.Tensor:
        need = self.workspace_bytes(total_tokens, n_seqs)
        if self._ws is None or self._ws.numel() < need:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws

    @staticmethod
    def _check_batch(ids: torch.Tensor, cu_seqlens: torch.Tensor):
        if ids.dtype != torch.int32 or cu_seqlens.dtype != torch.int32:
            raise TypeError("ids and cu_seqlens must be int32 device tensors")
        if not (ids.is_cuda and cu_seqlens.is_cuda and ids.is_contiguous() and cu_seqlens.is_contiguous()):
            raise ValueError("ids and cu_seqlens must be contiguous CUDA tensors")

    @property
    def handle(self) -> int:
        """Address of an lrx_encoder_handle {cfg*, weights*} for torch.ops.lrx.encode_packed (valid while this encoder lives)."""
        if getattr(self, "_handle", None) is None:
            class _H(C.Structure):
                _fields_ = [("cfg", C.c_void_p), ("w", C.c_void_p)]
            self._handle = _H(C.addressof(self._ccfg), C.addressof(self._cw))
        return C.addressof(self._handle)

    def encode_packed(self, ids: torch.Tensor, cu_seqlens: torch.Tensor, max_seqlen: int, out: Optional[torch.Tensor] = None,
                      out_dim: Optional[int] = None, normalize: bool = True, pooling: str = "lasttoken") -> torch.Tensor:
        """ids int32 [T], cu_seqlens int32 [B+1] (device).  Writes fp32 [B, out_dim] rows into `out` (e.g. a slice of the
        index shard: no host round trip) and returns it.  pooling: `--pooling_strategy` of the reference (finetune/dense_pooling.py:12-82):
        'lasttoken' (the released models), 'cls', 'mean', 'second_to_last', 'third_to_last', 'avg_first_last', 'avg_top2' (the last two pool
        over a second hidden state: the embedding rows / the stream as it enters the final layer)."""
        if pooling not in _lib.POOLING:
            raise NotImplementedError(f"pooling strategy {pooling!r}: served are {sorted(_lib.POOLING)}")
        self._check_batch(ids, cu_seqlens)
        T, B = ids.numel(), cu_seqlens.numel() - 1
        D = out_dim or self.cfg.hidden_size
        if out is None:
            out = torch.empty(B, D, dtype=torch.float32, device=self.device)
        if out.dtype != torch.float32 or out.shape[0] < B or out.shape[1] < D or out.stride(1) != 1:
            raise ValueError("out must be fp32 [>=B, >=out_dim] with unit inner stride")
        ws = self._workspace(T, B)
        # rows of a live index shard: the last kernel also writes their fp16 shadow and raises the shard's bounds (no second pass)
        from .index import shard_of
        hit = shard_of(out) if D == out.shape[1] else None
        shadow, srow0, bounds = hit[0].shard_sink(hit[1], B) if hit is not None else (None, 0, None)
        _lib.check(self.lib.lrx_encode_packed_pooled(C.byref(self._ccfg), C.byref(self._cw), _lib.ptr(ids), _lib.ptr(cu_seqlens), B, T,
                                                     int(max_seqlen), _lib.POOLING[pooling], _lib.ptr(out), out.stride(0), D, int(normalize),
                                                     _lib.ptr(shadow), int(srow0), _lib.ptr(bounds), _lib.ptr(ws), ws.numel(), _lib.current_stream()))
        return out[:B, :D]

    def encode_hidden(self, ids: torch.Tensor, cu_seqlens: torch.Tensor, max_seqlen: int) -> torch.Tensor:
        """last_hidden_state (after the final norm) bf16 [T, H]."""
        self._check_batch(ids, cu_seqlens)
        T, B = ids.numel(), cu_seqlens.numel() - 1
        out = torch.empty(T, self.cfg.hidden_size, dtype=torch.bfloat16, device=self.device)
        ws = self._workspace(T, B)
        _lib.check(self.lib.lrx_encode_hidden(C.byref(self._ccfg), C.byref(self._cw), _lib.ptr(ids), _lib.ptr(cu_seqlens), B, T,
                                              int(max_seqlen), _lib.ptr(out), _lib.ptr(ws), ws.numel(), _lib.current_stream()))
        return out

    def encode_packed_sparse(self, ids: torch.Tensor, cu_seqlens: torch.Tensor, max_seqlen: int, tok_mask: Optional[torch.Tensor] = None,
                             dense_dim: Optional[int] = None, normalize: bool = True, want_dense: bool = True, relu: bool = True,
                             log1p: bool = True, round_bf16: bool = True, top_k: int = 0, min_tokens_to_keep: int = 8):
        """Dense + sparse document vectors in one pass (lrx_encode_packed_sparse).  tok_mask: uint8 [T] on the GPU (None = drop
        each sequence's first and last token).  -> (dense fp32 [B,D] or None, sparse fp32 [B,V])."""
        self._check_batch(ids, cu_seqlens)
        T, B = ids.numel(), cu_seqlens.numel() - 1
        if tok_mask is not None and not (tok_mask.is_cuda and tok_mask.dtype == torch.uint8 and tok_mask.numel() == T and tok_mask.is_contiguous()):
            raise ValueError("tok_mask must be a contiguous uint8 CUDA tensor with one entry per token")
        D = dense_dim or self.cfg.hidden_size
        dense = torch.empty(B, D, dtype=torch.float32, device=self.device) if want_dense else None
        sparse = torch.empty(B, self.cfg.vocab_size, dtype=torch.float32, device=self.device)
        ws = self._workspace(T, B)
        lm_head = getattr(self, "lm_head", None)
        _lib.check(self.lib.lrx_encode_packed_sparse(
            C.byref(self._ccfg), C.byref(self._cw), _lib.ptr(lm_head) if lm_head is not None else None, None, _lib.ptr(ids), _lib.ptr(cu_seqlens),
            _lib.ptr(tok_mask) if tok_mask is not None else None, B, T, int(max_seqlen), _lib.ptr(dense) if want_dense else None,
            dense.stride(0) if want_dense else 0, D, int(normalize), _lib.ptr(sparse), sparse.stride(0), int(relu), int(log1p), int(round_bf16),
            int(top_k), int(min_tokens_to_keep), _lib.ptr(ws), ws.numel(), _lib.current_stream()))
        return dense, sparse

    def encode_prefixed(self, prefix_ids: torch.Tensor, suffix_ids: torch.Tensor, out: Optional[torch.Tensor] = None,
                        normalize: bool = False) -> torch.Tensor:
        """Sequences `prefix_ids + suffix_ids[i]` for every row i of suffix_ids [n, S2]; the shared prefix is encoded once
        (lrx_encode_prefixed).  Returns fp32 [n, H]: final-norm hidden state of each sequence's last token."""
        if prefix_ids.dtype != torch.int32 or suffix_ids.dtype != torch.int32:
            raise TypeError("prefix_ids and suffix_ids must be int32 device tensors")
        if not (prefix_ids.is_cuda and suffix_ids.is_cuda and prefix_ids.is_contiguous() and suffix_ids.is_contiguous()):
            raise ValueError("prefix_ids and suffix_ids must be contiguous CUDA tensors")
        if suffix_ids.dim() != 2:
            raise ValueError("suffix_ids must be [n_seqs, suffix_len]")
        n, S2 = suffix_ids.shape
        P1, H = prefix_ids.numel(), self.cfg.hidden_size
        if out is None:
            out = torch.empty(n, H, dtype=torch.float32, device=self.device)
        if out.dtype != torch.float32 or out.stride(-1) != 1 or out.shape[0] < n or out.shape[1] != H:
            raise ValueError("out must be fp32 [>=n, H] with unit inner stride")
        need = self.lib.lrx_encode_prefixed_workspace_bytes(C.byref(self._ccfg), P1, n, S2)
        if self._ws is None or self._ws.numel() < need:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        _lib.check(self.lib.lrx_encode_prefixed(C.byref(self._ccfg), C.byref(self._cw), _lib.ptr(prefix_ids) if P1 else None, P1,
                                                _lib.ptr(suffix_ids), n, S2, _lib.ptr(out), out.stride(0), H, int(normalize),
                                                _lib.ptr(self._ws), self._ws.numel(), _lib.current_stream()))
        return out[:n]

    # profiling hooks used by bench.py ------------------------------------------------------------------------
    def set_profiling(self, on: bool):
        self.lib.lrx_set_profiling(int(on))

    def get_profile(self) -> dict:
        n = _lib.LRX_PROF_CLASSES
        ms, fl, la = (C.c_float * n)(), (C.c_double * n)(), (C.c_int32 * n)()
        self.lib.lrx_get_profile(ms, fl, la)
        return {name: {"ms": ms[i], "flops": fl[i], "launches": la[i]} for i, name in enumerate(_lib.PROF_CLASS_NAMES)}
