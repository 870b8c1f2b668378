"""CPU tests of checkpoint / tokenizer loading (scope row a-13): HF safetensors dir, LoRA adapter merge, tokenizer surgery
against the fixture produced by the reference's load_tokenizer, and the `lightretriever.*` import-path shim."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import lrx_oracle as O
from helpers import GOLDEN


def tiny_hf(tmp_path, bias=False):
    from transformers import LlamaConfig, LlamaForCausalLM, Qwen2Config, Qwen2ForCausalLM
    torch.manual_seed(0)
    if bias:
        cfg = Qwen2Config(vocab_size=50, hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=2,
                          num_key_value_heads=1, rope_parameters={"rope_type": "default", "rope_theta": 1e6}, tie_word_embeddings=True)
        m = Qwen2ForCausalLM(cfg)
    else:
        cfg = LlamaConfig(vocab_size=50, hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=1,
                          head_dim=32, rope_parameters={"rope_type": "llama3", "rope_theta": 5e5, "factor": 8.0, "low_freq_factor": 1.0,
                                                        "high_freq_factor": 4.0, "original_max_position_embeddings": 64}, tie_word_embeddings=True)
        m = LlamaForCausalLM(cfg)
    d = str(tmp_path / ("qwen" if bias else "llama"))
    m.save_pretrained(d, safe_serialization=True)
    return m, d


@pytest.mark.parametrize("bias", [False, True])
def test_load_hf_checkpoint(tmp_path, bias):
    from lightretriever_amd.loader import load_hf_checkpoint
    m, d = tiny_hf(tmp_path, bias)
    cfg, sd = load_hf_checkpoint(d, max_positions=128)
    assert (cfg.hidden_size, cfg.num_layers, cfg.num_q_heads, cfg.num_kv_heads, cfg.intermediate_size) == (64, 2, 2, 1, 128)
    assert cfg.qkv_bias == bias and cfg.rope_type == ("default" if bias else "llama3") and cfg.head_dim == 32
    ref = {k: v for k, v in m.model.state_dict().items()}
    assert set(ref) <= set(sd)
    for k, v in ref.items():
        assert torch.equal(sd[k], v), k
    ocfg = O.EncoderConfig(**{k: v for k, v in cfg.__dict__.items() if k != "fold_norm"})
    assert set(O.weight_names(ocfg)) <= set(sd)          # every tensor the encoder needs is present


def test_lora_adapter_merge(tmp_path):
    from safetensors.torch import save_file
    from lightretriever_amd.loader import load_hf_checkpoint
    m, base = tiny_hf(tmp_path)
    r, alpha = 4, 8
    g = torch.Generator().manual_seed(1)
    ad, want = {}, {}
    for i in range(2):
        for mod, (o, inn) in {"self_attn.q_proj": (64, 64), "self_attn.v_proj": (32, 64), "mlp.down_proj": (64, 128)}.items():
            A, B = torch.randn(r, inn, generator=g) * 0.1, torch.randn(o, r, generator=g) * 0.1
            pre = f"base_model.model.model.layers.{i}.{mod}"
            ad[pre + ".lora_A.weight"], ad[pre + ".lora_B.weight"] = A, B
            W = m.model.state_dict()[f"layers.{i}.{mod}.weight"]
            want[f"layers.{i}.{mod}.weight"] = O.lora_merge(W.numpy(), A.numpy(), B.numpy(), alpha, r)
    adir = tmp_path / "adapter"
    adir.mkdir()
    save_file(ad, str(adir / "adapter_model.safetensors"))
    json.dump({"base_model_name_or_path": base, "r": r, "lora_alpha": alpha, "target_modules": ["q_proj", "v_proj", "down_proj"]},
              open(adir / "adapter_config.json", "w"))
    cfg, sd = load_hf_checkpoint(str(adir))
    for k, w in want.items():
        np.testing.assert_allclose(sd[k].numpy(), w, rtol=1e-6, atol=1e-7)
    untouched = "layers.0.self_attn.k_proj.weight"
    assert torch.equal(sd[untouched], m.model.state_dict()[untouched])


def test_tokenizer_surgery_matches_reference():
    from lightretriever_amd.loader import load_tokenizer
    fx = json.load(open(os.path.join(GOLDEN, "tokenizer_surgery.json")))
    tok = load_tokenizer(os.path.join(GOLDEN, "tok_raw"), lowercase=True, add_bos_num=1, add_eos_num=1, add_pad_token=True,
                         pad_token="<|reserved_special_token_0|>", add_sep_token=True, sep_token="<|reserved_special_token_1|>")
    assert (tok.pad_token_id, tok.sep_token_id, tok.bos_token_id, tok.eos_token_id, tok.padding_side) == \
        (fx["pad"], fx["sep"], fx["bos"], fx["eos"], fx["padding_side"])
    enc = tok(fx["texts"], max_length=fx["max_length"], truncation="only_first", padding=True, add_special_tokens=True)
    assert enc["input_ids"] == fx["input_ids"] and enc["attention_mask"] == fx["attention_mask"]
    assert tok(fx["texts"][0], add_special_tokens=False)["input_ids"] == fx["nospecial"]


def test_special_token_defaults_by_family():
    from lightretriever_amd.loader import default_special_tokens
    assert default_special_tokens("/ckpt/lightretriever-llama3.2-1b") == ("<|reserved_special_token_0|>", "<|reserved_special_token_1|>")
    assert default_special_tokens("/ckpt/Qwen2.5-1.5B") == ("<|im_end|>", "<|im_start|>")
    assert default_special_tokens("/ckpt/other", "<|p|>", "<|s|>") == ("<|p|>", "<|s|>")


def test_reference_import_paths_resolve():
    """The names eval/eval_utils.py:19-22,51,61,69 and eval/eval_arguments.py:5 import."""
    from lightretriever.inference.arguments import InferenceArguments
    from lightretriever.inference.utils import DEVICE_TYPE, DIST_BACKEND
    from lightretriever.inference.exact_search_torchrpc import PytorchRPCExactSearchModel
    from lightretriever.inference.rerank import RerankerModel
    from lightretriever.inference.dummy import DummyModel
    from lightretriever.retriever.hybrid_search import HybridSearch
    from lightretriever.retriever.faiss_search import FlatIPFaissSearch
    from lightretriever.retriever.anserini_search import AnseriniSearch
    assert DEVICE_TYPE in ("cuda", "cpu") and DIST_BACKEND in ("nccl", "gloo")
    args = InferenceArguments(model_name_or_path="/x/llama", score_function="dot")
    assert args.normalize is False and args.dtype == torch.bfloat16
    for cls in (RerankerModel, DummyModel, AnseriniSearch):
        with pytest.raises(NotImplementedError):
            cls()
    assert HybridSearch.name() == "hybrid_search" and FlatIPFaissSearch.name() == "faiss_search"
    assert callable(PytorchRPCExactSearchModel)
    with pytest.raises(NotImplementedError):
        InferenceArguments(model_name_or_path="/x", pooling_strategy="mean")
