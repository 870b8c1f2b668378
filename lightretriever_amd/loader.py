"""Checkpoint + tokenizer loading for the MI355X encoder (scope row a-13).

Mirrors what the reference does at load time, without `peft` / HF model classes:
  * EncoderModel._load_model (finetune/modeling_encoder.py:602-633): a checkpoint dir is either a plain HF model
    (config.json + *.safetensors) or a LoRA adapter dir (adapter_config.json + adapter_model.safetensors) whose
    `base_model_name_or_path` is loaded first and merged: W += (lora_alpha / r) * B @ A  (peft merge_and_unload).
  * load_tokenizer (utils/data_utils.py:29-117): right padding, pad / sep special tokens (defaults by model family,
    arguments.py:283-310), optional Lowercase normaliser in front, `<bos>*n A <eos>*m` TemplateProcessing.
"""
from __future__ import annotations

import json
import os
from typing import Optional

import torch

from .encoder import EncoderConfig, LrxEncoder, lora_merge


def _read_safetensors(dirname: str, stem: str = "model") -> dict:
    from safetensors.torch import load_file
    single = os.path.join(dirname, f"{stem}.safetensors")
    if os.path.exists(single):
        return load_file(single)
    index = os.path.join(dirname, f"{stem}.safetensors.index.json")
    if os.path.exists(index):
        out = {}
        for shard in sorted(set(json.load(open(index))["weight_map"].values())):
            out.update(load_file(os.path.join(dirname, shard)))
        return out
    raise FileNotFoundError(f"no {stem}.safetensors[.index.json] under {dirname}")


def _strip(name: str) -> Optional[str]:
    """HF causal-LM parameter name -> name relative to the inner model (`model.` prefix dropped); lm_head is unused."""
    for pre in ("base_model.model.model.", "base_model.model.", "model."):
        if name.startswith(pre):
            return name[len(pre):]
    return None if name.startswith("lm_head") else name


def load_hf_checkpoint(path: str, max_positions: int = 512) -> tuple[EncoderConfig, dict]:
    """-> (EncoderConfig, state_dict with names like `layers.0.self_attn.q_proj.weight`), LoRA merged if `path` is an
    adapter directory."""
    adapter_cfg = os.path.join(path, "adapter_config.json")
    if os.path.exists(adapter_cfg):
        acfg = json.load(open(adapter_cfg))
        base = acfg["base_model_name_or_path"]
        if not os.path.isdir(base):
            raise FileNotFoundError(f"LoRA base model {base!r} is not a local directory (no network here)")
        cfg, sd = load_hf_checkpoint(base, max_positions)
        scale_alpha, r = float(acfg["lora_alpha"]), int(acfg["r"])
        ad = _read_safetensors(path, "adapter_model")
        pairs = {}
        for k, v in ad.items():
            k = k.replace(".default", "")
            if ".lora_A." in k or ".lora_B." in k:
                mod, which = (k.split(".lora_A.")[0], "A") if ".lora_A." in k else (k.split(".lora_B.")[0], "B")
                pairs.setdefault(_strip(mod), {})[which] = v
        for mod, ab in pairs.items():
            name = mod + ".weight"
            if name not in sd:
                raise KeyError(f"LoRA target {name} not found in the base model")
            sd[name] = lora_merge(sd[name], ab["A"], ab["B"], scale_alpha, r)
        return cfg, sd
    cfg = EncoderConfig.from_hf_dict(json.load(open(os.path.join(path, "config.json"))), max_positions)
    sd = {}
    for k, v in _read_safetensors(path).items():
        n = _strip(k)
        if n is not None:
            sd[n] = v
    return cfg, sd


def encoder_from_pretrained(path: str, max_positions: int = 512, device: Optional[torch.device] = None) -> LrxEncoder:
    cfg, sd = load_hf_checkpoint(path, max_positions)
    return LrxEncoder(cfg, sd, device)


# ------------------------------------------------------------------------------------------------------------------
_PAD_BY_FAMILY = {"qwen": "<|im_end|>", "llama": "<|reserved_special_token_0|>", "mistral-7b-v0.1": "<unk>", "mistral-7b-v0.3": "[control_8]"}
_SEP_BY_FAMILY = {"qwen": "<|im_start|>", "llama": "<|reserved_special_token_1|>", "mistral-7b-v0.1": "<s>", "mistral-7b-v0.3": "[/INST]", "gemma": "<bos>"}


def default_special_tokens(model_name_or_path: str, pad_token: str = "<|pad|>", sep_token: str = "<|sep|>") -> tuple[str, str]:
    low = (model_name_or_path or "").lower()
    if pad_token == "<|pad|>":
        pad_token = next((v for k, v in _PAD_BY_FAMILY.items() if k in low), pad_token)
    if sep_token == "<|sep|>":
        sep_token = next((v for k, v in _SEP_BY_FAMILY.items() if k in low), sep_token)
    return pad_token, sep_token


def load_tokenizer(path: str, lowercase: bool = False, add_bos_num: int = -1, add_eos_num: int = -1, add_pad_token: bool = True,
                   pad_token: str = "<|pad|>", add_sep_token: bool = False, sep_token: str = "<|sep|>"):
    from tokenizers import normalizers, processors
    from transformers import AutoTokenizer
    tok = AutoTokenizer.from_pretrained(path, use_fast=True)
    tok.padding_side = "right"
    if add_bos_num > 0 and tok.bos_token is None:
        tok.add_special_tokens({"bos_token": "<|bos|>"})
    if add_eos_num > 0 and tok.eos_token is None:
        tok.add_special_tokens({"eos_token": "<|eos|>"})
    if add_pad_token and tok.pad_token is None:
        tok.add_special_tokens({"pad_token": pad_token})
    if add_sep_token and tok.sep_token is None:
        tok.add_special_tokens({"sep_token": sep_token})
    bt = tok.backend_tokenizer
    if lowercase:
        cur = bt.normalizer
        has_lower = cur is not None and "Lowercase" in str(cur)
        if cur is None:
            bt.normalizer = normalizers.Lowercase()
        elif not has_lower:
            bt.normalizer = normalizers.Sequence([normalizers.Lowercase(), cur])
    if add_bos_num >= 0 or add_eos_num >= 0:
        # `<bos>*n A <eos>*m`; pairs: `<bos>*n A B <eos>*m` (no separator between A and B, like the reference)
        bos = [tok.bos_token] * max(add_bos_num, 0)
        eos = [tok.eos_token] * max(add_eos_num, 0)
        specials = [(t, tok.convert_tokens_to_ids(t)) for t in dict.fromkeys(bos + eos)]
        tmpl = processors.TemplateProcessing(single=" ".join(bos + ["$A"] + eos), pair=" ".join(bos + ["$A", "$B"] + eos),
                                             special_tokens=specials)
        cur = bt.post_processor
        if cur is None or "TemplateProcessing" in str(type(cur)) or "TemplateProcessing" in str(cur)[:40]:
            bt.post_processor = tmpl
        else:
            bt.post_processor = processors.Sequence([cur, tmpl])
    return tok
