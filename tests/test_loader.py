"""Checkpoint ingestion, CPU side: format detection, HF config.json / GGUF metadata -> config POD, SafeTensors / GGUF
container parsing.  These are the only reference behaviours on this path that the reference's OWN tests pin, so they are replayed here one
for one:
  * /root/reference/src/loader/detect.rs:168-271            (9 format-detection tests)
  * /root/reference/src/loader/safetensors/config.rs:247-276 (Llama-3.2 rope_scaling config)
The parsers live in libblazr_hip.so (bz_loader.hip) and need no GPU."""
import json
import os
import struct

import numpy as np
import pytest

from blazr_amd import _lib as L
from blazr_amd import runtime, synth
import ckpt_writer as W


# ---- loader/detect.rs tests ---------------------------------------------------------------------------------------------------------------
def test_detect_safetensors_file(tmp_path):                       # detect.rs:168-180
    f = tmp_path / "model.safetensors"
    f.write_bytes(b"dummy")
    s = runtime.detect_model_source(str(f))
    assert s["format"] == "safetensors" and s["weights_path"] == str(f)


def test_detect_gguf_file(tmp_path):                              # detect.rs:182-195
    f = tmp_path / "model.gguf"
    f.write_bytes(b"dummy")
    s = runtime.detect_model_source(str(f))
    assert s["format"] == "gguf" and s["weights_path"] == str(f) and s["config_path"] is None


def test_detect_safetensors_directory(tmp_path):                  # detect.rs:197-208
    (tmp_path / "model.safetensors").write_bytes(b"dummy")
    (tmp_path / "config.json").write_bytes(b"{}")
    s = runtime.detect_model_source(str(tmp_path))
    assert s["format"] == "safetensors" and s["config_path"] is not None


def test_detect_gguf_directory(tmp_path):                         # detect.rs:210-219
    (tmp_path / "model-q4.gguf").write_bytes(b"dummy")
    assert runtime.detect_model_source(str(tmp_path))["format"] == "gguf"


def test_detect_safetensors_with_config_in_parent(tmp_path):      # detect.rs:221-233
    (tmp_path / "config.json").write_bytes(b"{}")
    f = tmp_path / "model.safetensors"
    f.write_bytes(b"dummy")
    s = runtime.detect_model_source(str(f))
    assert s["format"] == "safetensors" and s["config_path"] is not None


def test_detect_unsupported_extension(tmp_path):                  # detect.rs:235-245
    f = tmp_path / "model.bin"
    f.write_bytes(b"dummy")
    with pytest.raises(L.BlazrHipError):
        runtime.detect_model_source(str(f))


def test_detect_nonexistent_path():                               # detect.rs:247-251
    with pytest.raises(L.BlazrHipError):
        runtime.detect_model_source("/nonexistent/path/model")


def test_detect_empty_directory(tmp_path):                        # detect.rs:253-260
    with pytest.raises(L.BlazrHipError):
        runtime.detect_model_source(str(tmp_path))


def test_safetensors_preferred_over_gguf(tmp_path):               # detect.rs:262-271
    (tmp_path / "model.safetensors").write_bytes(b"dummy")
    (tmp_path / "model.gguf").write_bytes(b"dummy")
    assert runtime.detect_model_source(str(tmp_path))["format"] == "safetensors"


def test_detect_sharded_directory(tmp_path):                      # detect.rs:80-90 (first shard)
    for i in (1, 2):
        (tmp_path / ("model-%05d-of-00002.safetensors" % i)).write_bytes(b"dummy")
    s = runtime.detect_model_source(str(tmp_path))
    assert s["format"] == "safetensors" and s["weights_path"].endswith("model-00001-of-00002.safetensors")


# ---- loader/safetensors/detect_arch.rs:200-315 (7 tests of detect_architecture_from_names) -------------------------------------------------
def _hf_transformer_names(n):
    names = ["model.embed_tokens.weight", "model.norm.weight", "lm_head.weight"]
    for i in range(n):
        p = "model.layers.%d." % i
        names += [p + s for s in ("self_attn.q_proj.weight", "self_attn.k_proj.weight", "self_attn.v_proj.weight", "self_attn.o_proj.weight", "mlp.gate_proj.weight",
                                  "mlp.up_proj.weight", "mlp.down_proj.weight", "input_layernorm.weight", "post_attention_layernorm.weight")]
    return names


def _hf_mla_moe_names(n):
    names = ["model.embed_tokens.weight", "model.norm.weight", "lm_head.weight"]
    for i in range(n):
        p = "model.layers.%d." % i
        names += [p + s for s in ("self_attn.w_dkv.weight", "self_attn.w_q.weight", "self_attn.w_o.weight", "moe.gate.weight", "moe.experts.0.up_proj.weight",
                                  "moe.experts.0.down_proj.weight", "input_layernorm.weight")]
    return names


def test_hf_llama_32_layers():
    c = runtime.detect_architecture_from_names(_hf_transformer_names(32))
    assert c["format"] == "HuggingFace" and c["num_layers"] == 32 and not c["tie_word_embeddings"]
    assert all(t == "StandardTransformer" for t in c["layer_types"])


def test_hf_small_model_2_layers():
    c = runtime.detect_architecture_from_names(_hf_transformer_names(2))
    assert c["num_layers"] == 2 and c["format"] == "HuggingFace"


def test_hf_tied_embeddings():
    names = [n for n in _hf_transformer_names(4) if n != "lm_head.weight"]
    assert runtime.detect_architecture_from_names(names)["tie_word_embeddings"]


def test_hf_deepseek_mla_moe():
    c = runtime.detect_architecture_from_names(_hf_mla_moe_names(8))
    assert c["format"] == "HuggingFace" and c["num_layers"] == 8 and all(t == "MlaWithMoe" for t in c["layer_types"])


def test_hf_hybrid_transformer_and_mamba():
    names = ["model.embed_tokens.weight", "model.norm.weight", "lm_head.weight"]
    names += ["model.layers.0." + s for s in ("self_attn.q_proj.weight", "self_attn.k_proj.weight", "mlp.gate_proj.weight")]
    names += ["model.layers.1." + s for s in ("mamba2.mixer.A_log", "mamba2.mixer.conv1d.weight")]
    c = runtime.detect_architecture_from_names(names)
    assert c["num_layers"] == 2 and c["layer_types"] == ["StandardTransformer", "Mamba2"]


def test_hf_no_layers_errors():
    with pytest.raises(L.BlazrHipError):
        runtime.detect_architecture_from_names(["model.embed_tokens.weight", "model.norm.weight"])


# ---- loader/safetensors/config.rs:247-276 -------------------------------------------------------------------------------------------------
def test_huggingface_config_rope_scaling():
    text = '''{
        "architectures": ["LlamaForCausalLM"],
        "vocab_size": 128256,
        "hidden_size": 2048,
        "num_hidden_layers": 16,
        "num_attention_heads": 32,
        "num_key_value_heads": 8,
        "max_position_embeddings": 131072,
        "rope_theta": 500000.0,
        "rope_scaling": {
            "rope_type": "llama3",
            "factor": 32.0,
            "low_freq_factor": 1.0,
            "high_freq_factor": 4.0,
            "original_max_position_embeddings": 8192
        }
    }'''
    c, q = runtime.config_from_hf_json(text)
    assert c.rope_scaling == L.ROPE_LLAMA3 and c.rope_factor == 32.0 and c.rope_low_freq_factor == 1.0 and c.rope_high_freq_factor == 4.0
    assert (c.vocab, c.hidden, c.n_layers, c.n_heads, c.n_kv_heads, c.head_dim, c.max_seq_len) == (128256, 2048, 16, 32, 8, 64, 131072)
    assert c.rope_theta == 500000.0 and c.rope_original_max_pos == 8192
    assert q["quant_method"] is None and c.act_dtype == L.F16        # config.rs:27: no torch_dtype -> "f16"


def test_huggingface_config_rope_scaling_yarn():
    """YaRN (RopeScalingConfig scaling_type "yarn", config.rs:83-95): the Llama form (default betas / attention factor) and DeepSeek-V2's
    (mscale / mscale_all_dim: cos / sin factor = their mscale ratio, softmax mscale from mscale_all_dim)"""
    base = dict(vocab_size=10, hidden_size=256, num_hidden_layers=2, num_attention_heads=4, intermediate_size=512)
    c, _ = runtime.config_from_hf_json(json.dumps(dict(base, model_type="llama", rope_scaling=dict(rope_type="yarn", factor=4.0, original_max_position_embeddings=2048))))
    assert c.rope_scaling == L.ROPE_YARN and c.rope_factor == 4.0 and c.rope_original_max_pos == 2048
    assert c.rope_beta_fast == 0.0 and c.rope_beta_slow == 0.0 and c.rope_attn_factor == 0.0 and c.mla_softmax_mscale == 0.0      # 0 = defaults
    ds = dict(base, model_type="deepseek_v2", architectures=["DeepseekV2ForCausalLM"], kv_lora_rank=64, qk_nope_head_dim=32, qk_rope_head_dim=16, v_head_dim=32,
              n_routed_experts=4, num_experts_per_tok=2, moe_intermediate_size=64,
              rope_scaling=dict(type="yarn", factor=40, original_max_position_embeddings=4096, beta_fast=32, beta_slow=1, mscale=0.707, mscale_all_dim=0.707))
    c, _ = runtime.config_from_hf_json(json.dumps(ds))
    want = 0.1 * 0.707 * np.log(40.0) + 1.0
    assert c.rope_scaling == L.ROPE_YARN and c.rope_factor == 40.0 and c.rope_beta_fast == 32.0 and c.rope_beta_slow == 1.0
    assert abs(c.rope_attn_factor - 1.0) < 1e-6 and abs(c.mla_softmax_mscale - want) < 1e-6


def test_hf_config_defaults_dtype_and_quantization():
    # model/config.rs:119-145 defaults; config.rs:15-28 torch_dtype; detect_arch.rs:79-90 quantization_config; awq.rs:69-71 forces f16
    c, q = runtime.config_from_hf_json(json.dumps(dict(model_type="llama", vocab_size=10, hidden_size=256, num_hidden_layers=2, num_attention_heads=4,
                                                       intermediate_size=512, torch_dtype="bfloat16")))
    assert abs(c.rms_eps - 1e-5) < 1e-12 and c.rope_theta == 10000.0 and c.max_seq_len == 4096 and c.n_kv_heads == 4 and c.head_dim == 64
    assert c.act_dtype == L.BF16 and c.rope_scaling == L.ROPE_NONE and q["torch_dtype"] == L.BF16
    c, q = runtime.config_from_hf_json(json.dumps(dict(model_type="llama", vocab_size=10, hidden_size=256, num_hidden_layers=2, num_attention_heads=4,
                                                       intermediate_size=512, torch_dtype="bfloat16", quantization_config=dict(quant_method="AWQ", group_size=64, bits=4))))
    assert q == dict(quant_method="awq", group_size=64, torch_dtype=L.BF16) and c.act_dtype == L.F16
    with pytest.raises(L.BlazrHipError):
        runtime.config_from_hf_json("{not json")
    with pytest.raises(L.BlazrHipError):
        runtime.config_from_hf_json(json.dumps(dict(model_type="llama", hidden_size=8, rope_scaling=dict(rope_type="longrope", factor=40))))


@pytest.mark.parametrize("preset", ["tiny-bf16", "tiny-awq", "tiny-mamba2", "tiny-dsv2", "deepseek-v2-lite", "llama3.2-1b-bf16"])
def test_hf_config_round_trip_equals_synth_config(preset):
    if preset in synth.MAMBA_PRESETS:
        cfg = synth.make_mamba_config(preset)
    elif preset in synth.DSV2_PRESETS:
        cfg = synth.make_dsv2_config(preset)
    else:
        cfg = synth.make_config(preset)
    got, _ = runtime.config_from_hf_json(json.dumps(W.hf_config(cfg)))
    want = runtime.make_config(cfg)
    common = {"abi_version", "arch", "hidden", "n_layers", "vocab", "max_seq_len", "rms_eps", "act_dtype", "tie_embeddings"}
    attn = {"n_heads", "inter", "rope_theta", "rope_interleaved", "rope_scaling", "rope_factor", "rope_low_freq_factor", "rope_high_freq_factor", "rope_original_max_pos"}
    arch = cfg.get("arch", "llama")
    if arch == "mamba2":
        fields = common | {n for n, _ in L.ModelConfig._fields_ if n.startswith("ssm_")}
    elif arch == "deepseek2":
        fields = common | attn | {n for n, _ in L.ModelConfig._fields_ if n.startswith(("mla_", "moe_"))}   # n_kv_heads / head_dim: overridden by the latent-cache shape
    else:
        fields = common | attn | {"n_kv_heads", "head_dim"}
    for name, _ty in L.ModelConfig._fields_:
        if name not in fields:
            continue
        a, b = getattr(got, name), getattr(want, name)
        assert a == b or (isinstance(a, float) and abs(a - b) <= 1e-6 * max(abs(b), 1e-30)), (preset, name, a, b)


# ---- SafeTensors container --------------------------------------------------------------------------------------------------------------------
def test_safetensors_describe_single_and_sharded(tmp_path):
    model = synth.make_llama("tiny-awq")
    W.write_hf_checkpoint(str(tmp_path / "one"), model)
    W.write_hf_checkpoint(str(tmp_path / "three"), model, shards=3)
    one = runtime.safetensors_describe(str(tmp_path / "one"))
    three = runtime.safetensors_describe(str(tmp_path / "three" / "model-00001-of-00003.safetensors"))
    assert one["num_shards"] == 1 and three["num_shards"] == 3 and one["tensors"] == three["tensors"]
    t = one["tensors"]
    assert t["model.layers.0.self_attn.q_proj.qweight"] == dict(dtype="I32", shape=[256, 32], bytes=256 * 32 * 4)     # awq.rs:3-6 [K, N/8]
    assert t["model.layers.0.self_attn.q_proj.scales"]["dtype"] == "F16" and t["model.layers.0.self_attn.q_proj.qzeros"]["shape"] == [2, 32]
    assert t["model.embed_tokens.weight"] == dict(dtype="F16", shape=[1024, 256], bytes=1024 * 256 * 2)


def test_safetensors_malformed_files_fail_loudly(tmp_path):
    f = tmp_path / "model.safetensors"
    f.write_bytes(b"abc")
    with pytest.raises(L.BlazrHipError):
        runtime.safetensors_describe(str(f))
    f.write_bytes(struct.pack("<Q", 1 << 40) + b"{}")                # header length beyond the file
    with pytest.raises(L.BlazrHipError):
        runtime.safetensors_describe(str(f))
    hdr = json.dumps({"x": {"dtype": "F32", "shape": [4], "data_offsets": [0, 64]}}).encode()
    f.write_bytes(struct.pack("<Q", len(hdr)) + hdr + b"\0" * 16)   # data_offsets beyond the file
    with pytest.raises(L.BlazrHipError):
        runtime.safetensors_describe(str(f))


def test_safetensors_lying_headers_are_rejected(tmp_path):
    """sizes in the header are not trusted -- byte range != dtype x shape, absurd / negative / fractional dimensions, overflowing
    products (the safetensors crate the reference uses rejects these files: TensorInvalidInfo / ValidationOverflow)"""
    f = tmp_path / "model.safetensors"

    def write(entry, data=b"\0" * 64):
        hdr = json.dumps({"x": entry}).encode()
        f.write_bytes(struct.pack("<Q", len(hdr)) + hdr + data)

    write({"dtype": "F32", "shape": [4], "data_offsets": [0, 16]})
    assert runtime.safetensors_describe(str(f))["tensors"]["x"]["bytes"] == 16      # the honest file passes
    for entry in ({"dtype": "F32", "shape": [4096, 4096], "data_offsets": [0, 16]},             # 16 bytes claimed to be a 64 MiB matrix
                  {"dtype": "F16", "shape": [4], "data_offsets": [0, 16]},                      # twice the bytes the shape needs
                  {"dtype": "F32", "shape": [-4], "data_offsets": [0, 16]},
                  {"dtype": "F32", "shape": [2.5], "data_offsets": [0, 16]},
                  {"dtype": "F32", "shape": [1 << 40, 1 << 40], "data_offsets": [0, 16]},       # product overflows 64 bits
                  {"dtype": "F32", "shape": ["4"], "data_offsets": [0, 16]},
                  {"dtype": "F32", "shape": [4], "data_offsets": [-16, 0]},
                  {"dtype": "F32", "shape": [4], "data_offsets": [16, 0]}):
        write(entry)
        with pytest.raises(L.BlazrHipError):
            runtime.safetensors_describe(str(f))
    # a number that runs to the very end of the mapped header (no terminating NUL behind it) parses from a bounded copy
    hdr = b'{"x": {"dtype": "F32", "shape": [4], "data_offsets": [0, 16]}}'
    f.write_bytes(struct.pack("<Q", len(hdr)) + hdr + b"\0" * 16)
    assert runtime.safetensors_describe(str(f))["tensors"]["x"]["shape"] == [4]
    # names with quotes / backslashes / control characters come back as valid JSON
    hdr = json.dumps({'we"ird\\na\tme': {"dtype": "F32", "shape": [4], "data_offsets": [0, 16]}}).encode()
    f.write_bytes(struct.pack("<Q", len(hdr)) + hdr + b"\0" * 16)
    assert list(runtime.safetensors_describe(str(f))["tensors"]) == ['we"ird\\na\tme']


def test_gguf_lying_headers_are_rejected(tmp_path):
    model = synth.make_llama("tiny-q8_0")
    path = tmp_path / "m.gguf"
    W.write_gguf(str(path), model)
    cfg, info = runtime.config_from_gguf(str(path))
    assert info["n_tensors"] > 0
    raw = path.read_bytes()
    bad = tmp_path / "bad.gguf"
    bad.write_bytes(raw[:200])                                              # truncated inside the metadata
    with pytest.raises(L.BlazrHipError):
        runtime.config_from_gguf(str(bad))
    W.write_gguf(str(bad), model, extra_kv=[("general.alignment", W.GG_U32, 48)])   # not a power of two
    with pytest.raises(L.BlazrHipError):
        runtime.config_from_gguf(str(bad))
    W.write_gguf(str(bad), model, extra_kv=[("general.alignment", W.GG_U32, 1 << 20)])
    with pytest.raises(L.BlazrHipError):
        runtime.config_from_gguf(str(bad))


# ---- GGUF container + metadata (loader/gguf.rs:101-306) -----------------------------------------------------------------------------------------
def test_gguf_metadata_to_config(tmp_path):
    model = synth.make_llama("tiny-q4km")
    cfg = model["config"]
    p = str(tmp_path / "m.gguf")
    W.write_gguf(p, model)
    c, info = runtime.config_from_gguf(p)
    assert (c.vocab, c.hidden, c.n_layers, c.n_heads, c.n_kv_heads, c.head_dim, c.inter, c.max_seq_len) == \
        (cfg["vocab"], cfg["hidden"], cfg["n_layers"], cfg["n_heads"], cfg["n_kv_heads"], cfg["head_dim"], cfg["inter"], cfg["max_seq_len"])
    assert c.act_dtype == L.F32 and c.rope_interleaved == 1 and c.arch == L.ARCH_LLAMA          # gguf.rs:305
    assert info["architecture"] == "llama" and info["version"] == 3 and info["n_tensors"] == 3 + 9 * cfg["n_layers"]
    assert info["dominant_ggml_type"] == synth.GGML_Q4_K and not info["is_moe"] and not info["is_ssm"] and info["file_size_bytes"] == os.path.getsize(p)
    # vocab from the tokenizer array when general.vocab_size is absent (gguf.rs:110-120); head_dim from hidden / heads when key_length is absent
    W.write_gguf(p, model, with_vocab_array=True)
    assert runtime.config_from_gguf(p)[0].vocab == cfg["vocab"]


def test_gguf_metadata_ssm_moe_mla_and_defaults(tmp_path):
    def raw(kvs):
        b = b"GGUF" + struct.pack("<IQQ", 3, 0, len(kvs))
        for k, ty, v in kvs:
            b += W._gs(k) + struct.pack("<I", ty) + (W._gs(v) if ty == W.GG_STR else struct.pack("<I" if ty == W.GG_U32 else "<f", v))
        return b
    p = tmp_path / "x.gguf"
    p.write_bytes(raw([("general.architecture", W.GG_STR, "mamba2"), ("mamba2.embedding_length", W.GG_U32, 2560), ("mamba2.block_count", W.GG_U32, 64),
                       ("mamba2.ssm.state_size", W.GG_U32, 128), ("mamba2.ssm.inner_size", W.GG_U32, 5120)]))
    c, info = runtime.config_from_gguf(str(p))
    # gguf.rs:219-262: head_dim default 64, heads = inner / head_dim, conv kernel 4, one group
    assert c.arch == L.ARCH_MAMBA2 and (c.ssm_d_state, c.ssm_d_inner, c.ssm_head_dim, c.ssm_n_heads, c.ssm_conv_kernel, c.ssm_n_groups) == (128, 5120, 64, 80, 4, 1)
    assert info["is_ssm"] and c.vocab == 32000 and c.max_seq_len == 4096
    p.write_bytes(raw([("general.architecture", W.GG_STR, "deepseek2"), ("deepseek2.embedding_length", W.GG_U32, 2048), ("deepseek2.block_count", W.GG_U32, 27),
                       ("deepseek2.attention.head_count", W.GG_U32, 16), ("deepseek2.attention.kv_lora_rank", W.GG_U32, 512),
                       ("deepseek2.attention.rope_dimension_count", W.GG_U32, 64), ("deepseek2.expert_count", W.GG_U32, 64), ("deepseek2.expert_used_count", W.GG_U32, 6)]))
    c, info = runtime.config_from_gguf(str(p))
    assert info["is_mla"] and info["is_moe"] and (c.mla_kv_lora_rank, c.mla_rope_dim, c.moe_n_experts, c.moe_top_k) == (512, 64, 64, 6)
    p.write_bytes(raw([("general.architecture", W.GG_STR, "llama")]))
    with pytest.raises(L.BlazrHipError):                               # "GGUF missing llama.embedding_length"
        runtime.config_from_gguf(str(p))
    p.write_bytes(b"GGUX" + b"\0" * 40)
    with pytest.raises(L.BlazrHipError):
        runtime.config_from_gguf(str(p))
