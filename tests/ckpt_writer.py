"""Writers for synthetic checkpoints in the on-disk formats the reference loads (test infrastructure): SafeTensors (single file or sharded with
model.safetensors.index.json) + HF config.json, and GGUF v3.  Inputs are blazr_amd.synth model dicts; nothing here is used by the product."""
import json
import os
import struct

import numpy as np

ST_DT = {np.dtype(np.float32): "F32", np.dtype(np.float16): "F16", np.dtype(np.uint16): "BF16", np.dtype(np.int32): "I32", np.dtype(np.int64): "I64"}


def _act_store(x, act):
    """f32 array of act-representable values -> storage array in act dtype (bf16 as uint16 bits)"""
    x = np.asarray(x, dtype=np.float32)
    if act == "f16":
        return x.astype(np.float16)
    if act == "bf16":
        return (x.view(np.uint32) >> 16).astype(np.uint16)
    return x


def write_safetensors(path, tensors):
    header, off, blobs = {}, 0, []
    for name, a in tensors.items():
        a = np.ascontiguousarray(a)
        b = a.tobytes()
        header[name] = {"dtype": ST_DT[a.dtype], "shape": list(a.shape), "data_offsets": [off, off + len(b)]}
        off += len(b)
        blobs.append(b)
    header["__metadata__"] = {"format": "pt"}
    h = json.dumps(header).encode()
    h += b" " * ((8 - len(h) % 8) % 8)
    with open(path, "wb") as f:
        f.write(struct.pack("<Q", len(h)))
        f.write(h)
        for b in blobs:
            f.write(b)


def _linear_tensors(base, spec, out):
    k = spec["kind"]
    if k == "dense":
        out[base + ".weight"] = spec["weight"]
    elif k == "awq":
        out[base + ".qweight"] = spec["qweight"].view(np.int32)
        out[base + ".qzeros"] = spec["qzeros"].view(np.int32)
        out[base + ".scales"] = spec["scales"].astype(np.float16)
    elif k == "gptq":
        out[base + ".qweight"] = spec["qweight"].view(np.int32)
        out[base + ".qzeros"] = spec["qzeros"].view(np.int32)
        out[base + ".scales"] = spec["scales"].astype(np.float16)
        if spec.get("g_idx") is not None:
            out[base + ".g_idx"] = spec["g_idx"].astype(np.int32)
        if spec.get("bias") is not None:
            out[base + ".bias"] = spec["bias"].astype(np.float16)
    else:
        raise ValueError(k)


def hf_tensors(model):
    """HF-named tensor dict of a synth model (llama / mamba2 / deepseek2)"""
    cfg = model["config"]
    act = cfg["act_dtype"]
    arch = cfg.get("arch", "llama")
    t = {}
    if arch == "mamba2":
        t["backbone.embeddings.weight"] = model["embed"]
        t["backbone.norm_f.weight"] = _act_store(model["final_norm"], act)
        for i, lay in enumerate(model["layers"]):
            p = "backbone.layers.%d." % i
            t[p + "norm.weight"] = _act_store(lay["norm"], act)
            _linear_tensors(p + "mixer.in_proj", lay["in_proj"], t)
            cw = np.asarray(lay["conv_w"], np.float32)
            t[p + "mixer.conv1d.weight"] = _act_store(cw.reshape(cw.shape[0], 1, cw.shape[1]), act)
            t[p + "mixer.conv1d.bias"] = _act_store(lay["conv_b"], act)
            for n, key in (("dt_bias", "dt_bias"), ("A_log", "A_log"), ("D", "D")):
                t[p + "mixer." + n] = np.asarray(lay[key], np.float32)
            t[p + "mixer.norm.weight"] = _act_store(lay["gnorm"], act)
            _linear_tensors(p + "mixer.out_proj", lay["out_proj"], t)
        if not cfg.get("tie_embeddings"):
            _linear_tensors("lm_head", model["lm_head"], t)
        return t
    t["model.embed_tokens.weight"] = model["embed"]
    t["model.norm.weight"] = _act_store(model["final_norm"], act)
    if not cfg.get("tie_embeddings"):
        _linear_tensors("lm_head", model["lm_head"], t)
    for i, lay in enumerate(model["layers"]):
        p = "model.layers.%d." % i
        t[p + "input_layernorm.weight"] = _act_store(lay["attn_norm"], act)
        t[p + "post_attention_layernorm.weight"] = _act_store(lay["ffn_norm"], act)
        if arch == "deepseek2":
            t[p + "self_attn.kv_a_layernorm.weight"] = _act_store(lay["kv_norm"], act)
            for short, hf in (("q_proj", "self_attn.q_proj"), ("kv_a", "self_attn.kv_a_proj_with_mqa"), ("kv_b", "self_attn.kv_b_proj"), ("o", "self_attn.o_proj")):
                _linear_tensors(p + hf, lay[short], t)
            if not lay["is_moe"]:
                for n in ("gate", "up", "down"):
                    _linear_tensors(p + "mlp.%s_proj" % n, lay[n], t)
            else:
                _linear_tensors(p + "mlp.gate", lay["router"], t)
                for e, ex in enumerate(lay["experts"]):
                    for n in ("gate", "up", "down"):
                        _linear_tensors(p + "mlp.experts.%d.%s_proj" % (e, n), ex[n], t)
                if "shared" in lay:
                    for n in ("gate", "up", "down"):
                        _linear_tensors(p + "mlp.shared_experts.%s_proj" % n, lay["shared"][n], t)
        else:
            for short, hf in (("q", "self_attn.q_proj"), ("k", "self_attn.k_proj"), ("v", "self_attn.v_proj"), ("o", "self_attn.o_proj"),
                              ("gate", "mlp.gate_proj"), ("up", "mlp.up_proj"), ("down", "mlp.down_proj")):
                _linear_tensors(p + hf, lay[short], t)
    return t


def hf_config(cfg):
    """HF config.json dict for a synth config"""
    td = {"f16": "float16", "bf16": "bfloat16", "f32": "float32"}[cfg["act_dtype"]]
    arch = cfg.get("arch", "llama")
    if arch == "mamba2":
        return dict(model_type="mamba2", architectures=["Mamba2ForCausalLM"], hidden_size=cfg["hidden"], num_hidden_layers=cfg["n_layers"],
                    vocab_size=cfg["vocab"], state_size=cfg["d_state"], num_heads=cfg["n_heads"], head_dim=cfg["head_dim"], n_groups=cfg["n_groups"],
                    conv_kernel=cfg["conv_kernel"], expand=cfg["d_inner"] // cfg["hidden"], layer_norm_epsilon=cfg["rms_eps"],
                    tie_word_embeddings=bool(cfg.get("tie_embeddings")), torch_dtype=td, max_position_embeddings=cfg["max_seq_len"])
    j = dict(architectures=["LlamaForCausalLM"], model_type="llama", vocab_size=cfg["vocab"], hidden_size=cfg["hidden"], num_hidden_layers=cfg["n_layers"],
             num_attention_heads=cfg["n_heads"], intermediate_size=cfg["inter"], max_position_embeddings=cfg["max_seq_len"], rms_norm_eps=cfg["rms_eps"],
             rope_theta=cfg["rope_theta"], tie_word_embeddings=bool(cfg.get("tie_embeddings")), torch_dtype=td)
    if arch == "deepseek2":
        j.update(model_type="deepseek_v2", architectures=["DeepseekV2ForCausalLM"], kv_lora_rank=cfg["kv_lora_rank"], q_lora_rank=None,
                 qk_nope_head_dim=cfg["nope_dim"], qk_rope_head_dim=cfg["rope_dim"], v_head_dim=cfg["v_dim"], n_routed_experts=cfg["n_experts"],
                 n_shared_experts=cfg["n_shared"], num_experts_per_tok=cfg["top_k"], moe_intermediate_size=cfg["moe_inter"],
                 first_k_dense_replace=cfg["first_dense"], routed_scaling_factor=cfg["routed_scale"], norm_topk_prob=bool(cfg["norm_topk"]))
    else:
        j.update(num_key_value_heads=cfg["n_kv_heads"], head_dim=cfg["head_dim"])
        if cfg.get("rope_scaling"):
            rs = cfg["rope_scaling"]
            j["rope_scaling"] = dict(rope_type=rs["type"], factor=rs["factor"], low_freq_factor=rs["low_freq_factor"], high_freq_factor=rs["high_freq_factor"],
                                     original_max_position_embeddings=rs["original_max_position_embeddings"])
        if cfg.get("quant") in ("awq", "gptq"):
            j["quantization_config"] = dict(quant_method=cfg["quant"], bits=4, group_size=cfg["group_size"])
    return j


def write_hf_checkpoint(dirpath, model, shards=1, with_config=True):
    os.makedirs(dirpath, exist_ok=True)
    t = hf_tensors(model)
    if with_config:
        json.dump(hf_config(model["config"]), open(os.path.join(dirpath, "config.json"), "w"))
    if shards == 1:
        write_safetensors(os.path.join(dirpath, "model.safetensors"), t)
        return
    names = list(t)
    per = (len(names) + shards - 1) // shards
    weight_map = {}
    for s in range(shards):
        fn = "model-%05d-of-%05d.safetensors" % (s + 1, shards)
        part = {n: t[n] for n in names[s * per:(s + 1) * per]}
        write_safetensors(os.path.join(dirpath, fn), part)
        weight_map.update({n: fn for n in part})
    json.dump({"metadata": {}, "weight_map": weight_map}, open(os.path.join(dirpath, "model.safetensors.index.json"), "w"))


# ---- GGUF v3 --------------------------------------------------------------------------------------------------------------------------------
GG_U32, GG_F32, GG_STR, GG_ARR = 4, 6, 8, 9


def _gs(s):
    b = s.encode()
    return struct.pack("<Q", len(b)) + b


def write_gguf(path, model, arch="llama", extra_kv=None, with_vocab_array=False):
    cfg = model["config"]
    kv = [("general.architecture", GG_STR, arch), ("general.alignment", GG_U32, 32), (arch + ".embedding_length", GG_U32, cfg["hidden"]),
          (arch + ".block_count", GG_U32, cfg["n_layers"]), (arch + ".context_length", GG_U32, cfg["max_seq_len"]),
          (arch + ".feed_forward_length", GG_U32, cfg["inter"]), (arch + ".attention.head_count", GG_U32, cfg["n_heads"]),
          (arch + ".attention.head_count_kv", GG_U32, cfg["n_kv_heads"]), (arch + ".attention.key_length", GG_U32, cfg["head_dim"]),
          (arch + ".attention.layer_norm_rms_epsilon", GG_F32, cfg["rms_eps"]), (arch + ".rope.freq_base", GG_F32, cfg["rope_theta"])]
    if with_vocab_array:
        kv.append(("tokenizer.ggml.tokens", GG_ARR, ["t%d" % i for i in range(cfg["vocab"])]))
    else:
        kv.append(("general.vocab_size", GG_U32, cfg["vocab"]))
    kv += list(extra_kv or [])
    tensors = []   # (name, ne (innermost first), ggml type, bytes)

    def add_linear(name, spec):
        if spec["kind"] == "gguf":
            tensors.append((name, [spec["K"], spec["N"]], spec["ggml_type"], np.ascontiguousarray(spec["blocks"]).tobytes()))
        else:
            w = np.ascontiguousarray(spec["weight"])
            ty = {np.dtype(np.float32): 0, np.dtype(np.float16): 1, np.dtype(np.uint16): 30}[w.dtype]
            tensors.append((name, [spec["K"], spec["N"]], ty, w.tobytes()))

    emb = np.ascontiguousarray(model["embed"])
    tensors.append(("token_embd.weight", [cfg["hidden"], cfg["vocab"]], {np.dtype(np.float32): 0, np.dtype(np.float16): 1, np.dtype(np.uint16): 30}[emb.dtype], emb.tobytes()))
    tensors.append(("output_norm.weight", [cfg["hidden"]], 0, np.asarray(model["final_norm"], np.float32).tobytes()))
    if not cfg.get("tie_embeddings"):
        add_linear("output.weight", model["lm_head"])
    for i, lay in enumerate(model["layers"]):
        p = "blk.%d." % i
        tensors.append((p + "attn_norm.weight", [cfg["hidden"]], 0, np.asarray(lay["attn_norm"], np.float32).tobytes()))
        tensors.append((p + "ffn_norm.weight", [cfg["hidden"]], 0, np.asarray(lay["ffn_norm"], np.float32).tobytes()))
        for short, gg in (("q", "attn_q"), ("k", "attn_k"), ("v", "attn_v"), ("o", "attn_output"), ("gate", "ffn_gate"), ("up", "ffn_up"), ("down", "ffn_down")):
            add_linear(p + gg + ".weight", lay[short])
    with open(path, "wb") as f:
        f.write(b"GGUF" + struct.pack("<IQQ", 3, len(tensors), len(kv)))
        for k, ty, v in kv:
            f.write(_gs(k) + struct.pack("<I", ty))
            if ty == GG_U32:
                f.write(struct.pack("<I", v))
            elif ty == GG_F32:
                f.write(struct.pack("<f", v))
            elif ty == GG_STR:
                f.write(_gs(v))
            elif ty == GG_ARR:
                f.write(struct.pack("<IQ", GG_STR, len(v)))
                for s in v:
                    f.write(_gs(s))
        off = 0
        offs = []
        for name, ne, ty, b in tensors:
            offs.append(off)
            off = (off + len(b) + 31) // 32 * 32
        for (name, ne, ty, b), o in zip(tensors, offs):
            f.write(_gs(name) + struct.pack("<I", len(ne)) + b"".join(struct.pack("<Q", d) for d in ne) + struct.pack("<IQ", ty, o))
        pos = f.tell()
        f.write(b"\0" * ((32 - pos % 32) % 32))
        base = f.tell()
        for (name, ne, ty, b), o in zip(tensors, offs):
            f.seek(base + o)
            f.write(b)
        end = f.tell()
        f.write(b"\0" * ((32 - end % 32) % 32))


def write_gguf_mamba2(path, model):
    """A Mamba2 model in llama.cpp's GGUF conventions (arch "mamba2"): ssm_in / ssm_conv1d / ssm_dt.bias / ssm_a (= -exp(A_log)) / ssm_d / ssm_norm / ssm_out,
    unquantised tensors (F32 / F16 / BF16 as the synth model stores them), the per-head vectors with llama.cpp's unit axis."""
    cfg = model["config"]
    arch = "mamba2"
    kv = [("general.architecture", GG_STR, arch), ("general.alignment", GG_U32, 32), (arch + ".embedding_length", GG_U32, cfg["hidden"]),
          (arch + ".block_count", GG_U32, cfg["n_layers"]), (arch + ".context_length", GG_U32, 1 << 20),
          (arch + ".ssm.state_size", GG_U32, cfg["d_state"]), (arch + ".ssm.conv_kernel", GG_U32, cfg["conv_kernel"]), (arch + ".ssm.inner_size", GG_U32, cfg["d_inner"]),
          (arch + ".ssm.head_dim", GG_U32, cfg["head_dim"]), (arch + ".ssm.group_count", GG_U32, cfg["n_groups"]),
          (arch + ".attention.layer_norm_rms_epsilon", GG_F32, cfg["rms_eps"]), ("general.vocab_size", GG_U32, cfg["vocab"])]
    tensors = []
    ty_of = {np.dtype(np.float32): 0, np.dtype(np.float16): 1, np.dtype(np.uint16): 30}

    def add(name, ne, arr):
        a = np.ascontiguousarray(arr)
        tensors.append((name, ne, ty_of[a.dtype], a.tobytes()))

    D, DI, NH, G = cfg["hidden"], cfg["d_inner"], cfg["n_heads"], cfg["n_groups"]
    add("token_embd.weight", [D, cfg["vocab"]], model["embed"])
    add("output_norm.weight", [D], np.asarray(model["final_norm"], np.float32))
    if not cfg.get("tie_embeddings"):
        add("output.weight", [D, cfg["vocab"]], model["lm_head"]["weight"])
    for i, lay in enumerate(model["layers"]):
        p = "blk.%d." % i
        conv_dim = lay["conv_w"].shape[0]
        add(p + "attn_norm.weight", [D], np.asarray(lay["norm"], np.float32))
        add(p + "ssm_in.weight", [lay["in_proj"]["K"], lay["in_proj"]["N"]], lay["in_proj"]["weight"])
        add(p + "ssm_conv1d.weight", [cfg["conv_kernel"], conv_dim], np.asarray(lay["conv_w"], np.float32))
        add(p + "ssm_conv1d.bias", [conv_dim], np.asarray(lay["conv_b"], np.float32))
        add(p + "ssm_dt.bias", [NH], np.asarray(lay["dt_bias"], np.float32))
        add(p + "ssm_a", [1, NH], (-np.exp(np.asarray(lay["A_log"], np.float64))).astype(np.float32))
        add(p + "ssm_d", [1, NH], np.asarray(lay["D"], np.float32))
        add(p + "ssm_norm.weight", [DI // G, G], np.asarray(lay["gnorm"], np.float32))
        add(p + "ssm_out.weight", [lay["out_proj"]["K"], lay["out_proj"]["N"]], lay["out_proj"]["weight"])
    with open(path, "wb") as f:
        f.write(b"GGUF" + struct.pack("<IQQ", 3, len(tensors), len(kv)))
        for k, ty, v in kv:
            f.write(_gs(k) + struct.pack("<I", ty))
            f.write(struct.pack("<I", v) if ty == GG_U32 else (struct.pack("<f", v) if ty == GG_F32 else _gs(v)))
        off, offs = 0, []
        for name, ne, ty, b in tensors:
            offs.append(off)
            off = (off + len(b) + 31) // 32 * 32
        for (name, ne, ty, b), o in zip(tensors, offs):
            f.write(_gs(name) + struct.pack("<I", len(ne)) + b"".join(struct.pack("<Q", d) for d in ne) + struct.pack("<IQ", ty, o))
        pos = f.tell()
        f.write(b"\0" * ((32 - pos % 32) % 32))
        base = f.tell()
        for (name, ne, ty, b), o in zip(tensors, offs):
            f.seek(base + o)
            f.write(b)
        end = f.tell()
        f.write(b"\0" * ((32 - end % 32) % 32))
