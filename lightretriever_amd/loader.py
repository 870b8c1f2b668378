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

from .encoder import EncoderConfig, LrxEncoder


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
    """HF causal-LM parameter name -> name relative to the inner model (`model.` prefix dropped; the CausalLM head keeps its own
    name `lm_head.*`: the sparse branch projects with it, finetune/modeling_hybrid.py:72-86 get_lm_head)."""
    for pre in ("base_model.model.", ):
        if name.startswith(pre):
            name = name[len(pre):]
    if name.startswith("lm_head."):
        return name
    return name[len("model."):] if name.startswith("model.") else name


def _lora_scale(acfg: dict, module: str) -> float:
    """peft's scaling for one target module: lora_alpha / r (alpha / sqrt(r) with use_rslora), r and alpha overridable per module
    through rank_pattern / alpha_pattern (keys are module-name suffixes or regexes, peft/tuners/lora/model.py)."""
    import math
    import re

    def pick(pattern: dict, default):
        for key, val in (pattern or {}).items():
            if re.match(rf"(.*\.)?({key})$", module):
                return val
        return default

    r = int(pick(acfg.get("rank_pattern"), acfg["r"]))
    alpha = float(pick(acfg.get("alpha_pattern"), acfg["lora_alpha"]))
    return alpha / math.sqrt(r) if acfg.get("use_rslora") else alpha / r


def merge_lora_adapter(sd: dict, adapter: dict, acfg: dict) -> dict:
    """peft merge_and_unload on a plain state dict (finetune/modeling_encoder.py:616-625): W += scale * B @ A for every LoRA pair,
    `modules_to_save` / saved embedding layers replace the base tensor.  Options that change the arithmetic are applied
    (use_rslora, rank_pattern, alpha_pattern, fan_in_fan_out); anything this merge does not implement raises -- an adapter tensor
    is never dropped silently."""
    if acfg.get("use_dora"):
        raise NotImplementedError("LoRA adapter with use_dora=True (weight-decomposed LoRA) is not supported by this loader")
    if acfg.get("bias", "none") != "none":
        raise NotImplementedError(f"LoRA adapter trained with bias={acfg['bias']!r}: bias deltas are not merged by this loader")
    pairs, unused = {}, []
    for k, v in adapter.items():
        k = k.replace(".default.", ".").replace(".default", "")
        if ".lora_A." in k or ".lora_B." in k:
            mod, which = (k.split(".lora_A.")[0], "A") if ".lora_A." in k else (k.split(".lora_B.")[0], "B")
            pairs.setdefault(_strip(mod), {})[which] = v
        elif ".modules_to_save." in k:                   # a fully trained copy of the module (e.g. embed_tokens / lm_head after adding tokens)
            name = _strip(k.replace(".modules_to_save.", "."))
            sd[name] = v
        elif ".original_module." in k:
            continue                                     # peft's frozen copy next to modules_to_save
        elif _strip(k) in sd or _strip(k) in ("lm_head.weight", "embed_tokens.weight"):   # embedding layers saved with the adapter (save_embedding_layers)
            sd[_strip(k)] = v
        else:
            unused.append(k)
    if unused:
        raise NotImplementedError(f"adapter tensors this loader cannot merge (lora_embedding_*, DoRA magnitudes, ...): {unused[:5]}"
                                  f"{' ...' if len(unused) > 5 else ''}")
    for mod, ab in pairs.items():
        name = mod + ".weight"
        if name == "lm_head.weight" and name not in sd:  # LoRA on the head of a tied model: the merged head is its own tensor
            sd[name] = sd["embed_tokens.weight"].clone()
        if name not in sd:
            raise KeyError(f"LoRA target {name} not found in the base model")
        if "A" not in ab or "B" not in ab:
            raise KeyError(f"LoRA target {mod}: lora_A / lora_B pair incomplete")
        delta = ab["B"].float() @ ab["A"].float()
        if acfg.get("fan_in_fan_out"):
            delta = delta.T
        sd[name] = (sd[name].float() + _lora_scale(acfg, mod) * delta).to(sd[name].dtype)
    return sd


def resize_embeddings(cfg: EncoderConfig, sd: dict, n_tokens: int, pad_to_multiple_of: Optional[int] = None) -> None:
    """resize_emb (utils/data_utils.py:273-281): a tokenizer that gained special tokens needs as many embedding rows.  The reference
    lets HF draw the new rows at random (mean-resizing); here they are the mean of the existing rows (the centre of that
    distribution), deterministic.  A no-op for the released checkpoints, whose special tokens pre-exist."""
    rows = sd["embed_tokens.weight"].shape[0]
    if n_tokens <= rows:
        return
    new_rows = n_tokens if not pad_to_multiple_of else -(-n_tokens // pad_to_multiple_of) * pad_to_multiple_of
    import logging
    logging.getLogger(__name__).warning("tokenizer has %d tokens but the checkpoint %d embedding rows: growing to %d (new rows = mean row)",
                                        n_tokens, rows, new_rows)
    for name in ("embed_tokens.weight", "lm_head.weight"):
        if name in sd:
            w = sd[name]
            extra = w.float().mean(0, keepdim=True).to(w.dtype).expand(new_rows - rows, -1)
            sd[name] = torch.cat([w, extra], 0)
    cfg.vocab_size = new_rows


def _resolve_local_model(name: str) -> str:
    """A LoRA adapter names its base model the way it was trained (`Qwen/Qwen2.5-1.5B`, scripts/asymmetric_dense_infer.ipynb cell 8 hands
    that string to AutoModelForCausalLM.from_pretrained): a local directory is taken as is, a hub id is looked up in the local HF cache
    -- never downloaded."""
    if os.path.isdir(name):
        return name
    try:
        from huggingface_hub import snapshot_download
        return snapshot_download(name, local_files_only=True)
    except Exception as e:
        raise FileNotFoundError(f"LoRA base model {name!r} is neither a local directory nor in the local HF hub cache (nothing is "
                                f"downloaded here): {type(e).__name__}") from e


def load_hf_checkpoint(path: str, max_positions: int = 512, n_tokens: Optional[int] = None,
                       pad_to_multiple_of: Optional[int] = None) -> tuple[EncoderConfig, dict]:
    """-> (EncoderConfig, state_dict with names like `layers.0.self_attn.q_proj.weight`, plus `lm_head.weight` when the checkpoint
    has an untied head), LoRA merged if `path` is an adapter directory; embeddings grown to n_tokens (= len(tokenizer)) if needed."""
    adapter_cfg = os.path.join(path, "adapter_config.json")
    if os.path.exists(adapter_cfg):
        acfg = json.load(open(adapter_cfg))
        base = _resolve_local_model(acfg["base_model_name_or_path"])
        cfg, sd = load_hf_checkpoint(base, max_positions)
        sd = merge_lora_adapter(sd, _read_safetensors(path, "adapter_model"), acfg)
        cfg.vocab_size = sd["embed_tokens.weight"].shape[0]
    else:
        hf = json.load(open(os.path.join(path, "config.json")))
        cfg = EncoderConfig.from_hf_dict(hf, max_positions)
        sd = {_strip(k): v for k, v in _read_safetensors(path).items()}
        # HF's to_diff_dict leaves `tie_word_embeddings` out of config.json when it equals the class default (tied for PretrainedConfig
        # and the Gemma-style configs), so an ABSENT key says nothing: then the checkpoint decides -- no lm_head.weight = tied.  Only an
        # explicit `false` with the tensor missing is an error.
        tie = hf.get("tie_word_embeddings")
        if tie is None:
            tie = "lm_head.weight" not in sd
        if tie:
            sd.pop("lm_head.weight", None)               # tied: the head IS the embedding matrix (LrxEncoder falls back to it)
        elif "lm_head.weight" not in sd:
            raise KeyError(f"{path}: config.json says tie_word_embeddings=false but the checkpoint has no lm_head.weight")
        if sd["embed_tokens.weight"].shape[0] != cfg.vocab_size:
            raise ValueError(f"{path}: embed_tokens has {sd['embed_tokens.weight'].shape[0]} rows, config.json vocab_size={cfg.vocab_size}")
    if n_tokens is not None:
        resize_embeddings(cfg, sd, n_tokens, pad_to_multiple_of)
    return cfg, sd


def encoder_from_pretrained(path: str, max_positions: int = 512, device: Optional[torch.device] = None, tokenizer=None,
                            pad_to_multiple_of: Optional[int] = None, precise_stream: Optional[bool] = None,
                            operand_dtype: Optional[str] = None) -> LrxEncoder:
    """tokenizer: when given, the embedding matrix is grown to len(tokenizer) like the reference's resize_emb, and the result is
    checked: every id the tokenizer can produce must have a row.  precise_stream / operand_dtype: the arithmetic switches of EncoderConfig
    (None = the defaults: fp32 residual stream, the QKV projection's operands in fp16)."""
    cfg, sd = load_hf_checkpoint(path, max_positions, n_tokens=len(tokenizer) if tokenizer is not None else None,
                                 pad_to_multiple_of=pad_to_multiple_of)
    if tokenizer is not None and len(tokenizer) > cfg.vocab_size:
        raise ValueError(f"tokenizer has {len(tokenizer)} tokens but the model only {cfg.vocab_size} embedding rows")
    if precise_stream is not None:
        cfg.precise_stream = bool(precise_stream)
    if operand_dtype is not None:
        cfg.operand_dtype = operand_dtype
    return LrxEncoder(cfg, sd, device)


def load_model_args(path: str) -> dict:
    """model_args.yaml written next to a trained retriever by EncoderModel.save (finetune/modeling_encoder.py:820-822) and read back
    by EncoderModel._load_model_args (:635-656) when HybridModel.load(path) is called without arguments: the flags the checkpoint
    was trained with (pooling, normalisation, tokenizer surgery, MRL dims, sparse options ...)."""
    import yaml
    f = os.path.join(path, "model_args.yaml")
    if not os.path.exists(f):
        raise FileNotFoundError(f"{f} not found: pass the model arguments explicitly")

    class _Loader(yaml.SafeLoader):
        pass
    # yaml.dump(model_args.__dict__) tags non-plain values (torch dtypes, tuples, enums); they never matter on this path
    _Loader.add_multi_constructor("tag:yaml.org,2002:python/", lambda loader, suffix, node: None)
    d = yaml.load(open(f), Loader=_Loader) or {}
    d["model_name_or_path"] = path                       # the reference re-points the args at the directory they were loaded from
    return d


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
