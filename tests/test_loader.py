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
    ocfg = O.EncoderConfig(**{k: v for k, v in cfg.__dict__.items() if k not in ("fold_norm", "precise_stream", "operand_dtype")})
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
    args = InferenceArguments(model_name_or_path="/x/llama", score_function="dot", bf16=True)
    assert args.normalize is False and args.dtype == torch.bfloat16
    for cls in (RerankerModel, DummyModel, AnseriniSearch):
        with pytest.raises(NotImplementedError):
            cls()
    assert HybridSearch.name() == "hybrid_search" and FlatIPFaissSearch.name() == "faiss_search"
    assert callable(PytorchRPCExactSearchModel)
    with pytest.raises(NotImplementedError):
        InferenceArguments(model_name_or_path="/x", pooling_strategy="none")                # ('none' returns no vector; every other strategy is served)


# ---- round 2: loader hardening ------------------------------------------------------------------
def tiny_untied(tmp_path):
    from transformers import LlamaConfig, LlamaForCausalLM
    torch.manual_seed(3)
    cfg = LlamaConfig(vocab_size=50, hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=1,
                      head_dim=32, rope_parameters={"rope_type": "default", "rope_theta": 1e4}, tie_word_embeddings=False)
    m = LlamaForCausalLM(cfg)
    d = str(tmp_path / "untied")
    m.save_pretrained(d, safe_serialization=True)
    return m, d


def test_untied_lm_head_is_kept_and_tied_head_is_not_duplicated(tmp_path):
    """checkpoints with tie_word_embeddings=false (Llama-3.1-8B, Qwen2.5-7B) project the sparse branch with their
    own head (finetune/modeling_hybrid.py:72-86 get_lm_head); tied ones with the embedding matrix."""
    from lightretriever_amd.loader import load_hf_checkpoint
    m, d = tiny_untied(tmp_path)
    cfg, sd = load_hf_checkpoint(d)
    assert "lm_head.weight" in sd and torch.equal(sd["lm_head.weight"], m.lm_head.weight.detach())
    assert not torch.equal(sd["lm_head.weight"], sd["embed_tokens.weight"])
    mt, dt = tiny_hf(tmp_path)
    _, sdt = load_hf_checkpoint(dt)
    assert "lm_head.weight" not in sdt
    # an untied config whose head tensor is missing must not fall back silently
    import safetensors.torch as st
    bad = tmp_path / "nohead"
    bad.mkdir()
    tensors = {k: v.contiguous() for k, v in st.load_file(os.path.join(d, "model.safetensors")).items() if not k.startswith("lm_head")}
    st.save_file(tensors, str(bad / "model.safetensors"))
    json.dump(json.load(open(os.path.join(d, "config.json"))), open(bad / "config.json", "w"))
    with pytest.raises(KeyError, match="lm_head"):
        load_hf_checkpoint(str(bad))
    # config.json WITHOUT the key (HF's to_diff_dict drops values equal to the class default): the checkpoint decides --
    # no lm_head.weight = tied, a separate head stays
    for src, name, has_head in ((str(bad), "nokey_tied", False), (d, "nokey_untied", True)):
        nk = tmp_path / name
        nk.mkdir()
        c = json.load(open(os.path.join(src, "config.json")))
        c.pop("tie_word_embeddings")
        json.dump(c, open(nk / "config.json", "w"))
        st.save_file({k: v.contiguous() for k, v in st.load_file(os.path.join(src, "model.safetensors")).items()}, str(nk / "model.safetensors"))
        _, sdn = load_hf_checkpoint(str(nk))
        assert ("lm_head.weight" in sdn) == has_head


def _adapter(tmp_path, base, tensors, **cfg):
    from safetensors.torch import save_file
    adir = tmp_path / ("adapter_" + "_".join(sorted(cfg)) if cfg else "adapter_plain")
    adir.mkdir(exist_ok=True)
    save_file({k: v.contiguous() for k, v in tensors.items()}, str(adir / "adapter_model.safetensors"))
    json.dump(dict({"base_model_name_or_path": base, "r": 4, "lora_alpha": 8}, **cfg), open(adir / "adapter_config.json", "w"))
    return str(adir)


def test_lora_options_that_change_the_merge(tmp_path):
    """use_rslora, rank_pattern / alpha_pattern, fan_in_fan_out are honoured like peft's merge_and_unload."""
    from lightretriever_amd.loader import load_hf_checkpoint
    m, base = tiny_hf(tmp_path)
    g = torch.Generator().manual_seed(2)
    A4, B4 = torch.randn(4, 64, generator=g) * 0.1, torch.randn(64, 4, generator=g) * 0.1
    A2, B2 = torch.randn(2, 64, generator=g) * 0.1, torch.randn(32, 2, generator=g) * 0.1
    q, v = "base_model.model.model.layers.0.self_attn.q_proj", "base_model.model.model.layers.1.self_attn.v_proj"
    tensors = {q + ".lora_A.weight": A4, q + ".lora_B.weight": B4, v + ".lora_A.weight": A2, v + ".lora_B.weight": B2}
    Wq = m.model.state_dict()["layers.0.self_attn.q_proj.weight"].float()
    Wv = m.model.state_dict()["layers.1.self_attn.v_proj.weight"].float()
    # rslora: alpha / sqrt(r); rank_pattern gives v_proj r = 2, alpha_pattern alpha = 3
    _, sd = load_hf_checkpoint(_adapter(tmp_path, base, tensors, use_rslora=True, rank_pattern={"v_proj": 2}, alpha_pattern={"layers.1.self_attn.v_proj": 3}))
    torch.testing.assert_close(sd["layers.0.self_attn.q_proj.weight"], Wq + (8 / 2.0) * (B4 @ A4))
    torch.testing.assert_close(sd["layers.1.self_attn.v_proj.weight"], Wv + (3 / 2 ** 0.5) * (B2 @ A2))
    # fan_in_fan_out: the stored weight is the transpose, so is the delta
    At, Bt = torch.randn(4, 64, generator=g) * 0.1, torch.randn(64, 4, generator=g) * 0.1
    _, sd = load_hf_checkpoint(_adapter(tmp_path, base, {q + ".lora_A.weight": At, q + ".lora_B.weight": Bt}, fan_in_fan_out=True))
    torch.testing.assert_close(sd["layers.0.self_attn.q_proj.weight"], Wq + 2.0 * (Bt @ At).T)
    for bad in ({"use_dora": True}, {"bias": "all"}):
        with pytest.raises(NotImplementedError):
            load_hf_checkpoint(_adapter(tmp_path, base, tensors, **bad))


def test_adapter_tensors_are_never_dropped_silently(tmp_path):
    """modules_to_save / saved embedding layers are loaded; tensors the merge cannot consume raise."""
    from lightretriever_amd.loader import load_hf_checkpoint
    m, base = tiny_hf(tmp_path)
    g = torch.Generator().manual_seed(4)
    emb = torch.randn(53, 64, generator=g)           # embedding grown by three tokens and trained (modules_to_save)
    norm = torch.randn(64, generator=g)
    cfg, sd = load_hf_checkpoint(_adapter(tmp_path, base, {"base_model.model.model.embed_tokens.modules_to_save.default.weight": emb,
                                                           "base_model.model.model.norm.modules_to_save.weight": norm,
                                                           "base_model.model.model.embed_tokens.original_module.weight": emb * 0}))
    assert torch.equal(sd["embed_tokens.weight"], emb) and torch.equal(sd["norm.weight"], norm) and cfg.vocab_size == 53
    with pytest.raises(NotImplementedError, match="lora_embedding_A"):
        load_hf_checkpoint(_adapter(tmp_path, base, {"base_model.model.model.embed_tokens.lora_embedding_A": torch.zeros(4, 50)}, target_modules=["embed_tokens"]))


def test_resize_embeddings_like_resize_emb(tmp_path):
    """utils/data_utils.py:273-281: len(tokenizer) > rows grows embed_tokens (and an untied head), optionally to a multiple."""
    from lightretriever_amd.loader import load_hf_checkpoint
    m, d = tiny_untied(tmp_path)
    cfg, sd = load_hf_checkpoint(d, n_tokens=50)
    assert cfg.vocab_size == 50 and sd["embed_tokens.weight"].shape[0] == 50                 # no-op when the tokens pre-exist
    cfg, sd = load_hf_checkpoint(d, n_tokens=53, pad_to_multiple_of=8)
    assert cfg.vocab_size == 56 and sd["embed_tokens.weight"].shape == (56, 64) and sd["lm_head.weight"].shape == (56, 64)
    assert torch.equal(sd["embed_tokens.weight"][:50], m.model.embed_tokens.weight.detach())
    torch.testing.assert_close(sd["embed_tokens.weight"][52], m.model.embed_tokens.weight.detach().mean(0))


def test_model_args_yaml_resume(tmp_path, caplog):
    """HybridModel.load(path) without arguments (finetune/modeling_encoder.py:635-656): the flags come from model_args.yaml."""
    import yaml
    from lightretriever_amd.inference import arguments_from_checkpoint
    from lightretriever_amd.loader import load_model_args
    d = tmp_path / "ckpt-llama"
    d.mkdir()
    saved = {"model_name_or_path": "/training/box/path", "pooling_strategy": "lasttoken", "score_function": "cos_sim", "normalize": True,
             "dense_shrink_dim": 256, "lowercase": True, "add_bos_num": 1, "add_eos_num": 1, "add_sep_token": True, "hybrid_use_emb_vector": True,
             "noncontextual_query_embedding": True, "hybrid_use_token_id_vector": True, "sparse_use_relu": True, "sparse_top_k_psg": 512,
             "clloss_coef": 1.0, "matryoshka_dims": [256, 512], "gc_q_chunk_size": 32}                        # training-only keys are ignored
    with open(d / "model_args.yaml", "w") as f:
        yaml.dump(saved, f, indent=2)
        f.write("torch_dtype: !!python/object/apply:torch._utils._rebuild_dtype [bfloat16]\n")              # non-plain tags do not break the resume
    raw = load_model_args(str(d))
    assert raw["model_name_or_path"] == str(d) and raw["dense_shrink_dim"] == 256
    args = arguments_from_checkpoint(str(d), p_max_len=256)
    assert (args.model_name_or_path, args.dense_shrink_dim, args.lowercase, args.add_bos_num, args.add_sep_token, args.p_max_len) == (str(d), 256, True, 1, True, 256)
    assert args.normalize is True and args.encode_sparse and args.sparse_top_k_psg == 512
    assert args.pad_token == "<|reserved_special_token_0|>"                                                   # family default from the directory name
    with pytest.raises(FileNotFoundError):
        load_model_args(str(tmp_path))
    assert args.model_type == "HybridModel"                                                                   # (this mirrors HybridModel.load)
    # the released checkpoints were trained with the symmetric dense vector as well (scripts/finetune_example.sh:47): since round 5 it is
    # served as saved (queries through the LM next to the EmbeddingBag vector)
    saved["hybrid_use_dense_vector"] = True
    yaml.dump(saved, open(d / "model_args.yaml", "w"))
    args = arguments_from_checkpoint(str(d))
    assert args.hybrid_use_dense_vector and args.hybrid_use_emb_vector and args.noncontextual_query_embedding
    # round 6: LM-head sparse QUERY vectors (`hybrid_use_sparse_vector`, modeling_hybrid.py:404-438) are served as saved -- next to token-id
    # queries when the checkpoint has both, alone when it has only them; an explicit override still wins
    saved["hybrid_use_sparse_vector"] = True
    yaml.dump(saved, open(d / "model_args.yaml", "w"))
    args = arguments_from_checkpoint(str(d))
    assert args.hybrid_use_sparse_vector and args.hybrid_use_token_id_vector and args.encode_sparse
    saved["hybrid_use_token_id_vector"] = False
    yaml.dump(saved, open(d / "model_args.yaml", "w"))
    args = arguments_from_checkpoint(str(d))
    assert args.hybrid_use_sparse_vector and not args.hybrid_use_token_id_vector and args.encode_sparse
    args = arguments_from_checkpoint(str(d), hybrid_use_sparse_vector=False)
    assert not args.encode_sparse and args.hybrid_use_dense_vector
    saved["hybrid_use_sparse_vector"], saved["hybrid_use_token_id_vector"] = False, True
    saved["untie_encoder"] = True                                                                             # an unsupported saved flag still fails loudly
    yaml.dump(saved, open(d / "model_args.yaml", "w"))
    with pytest.raises(NotImplementedError):
        arguments_from_checkpoint(str(d))
