"""Seeded synthetic checkpoints in the tensor formats blazr's loaders hand to the forward path.

There is no network and no real checkpoint in the build or bench environment, so every model is
random-init at the reference's shapes.  Tensors are produced exactly in the form the
reference's loaders produce them, so they can be fed to `LoadedModel.add_*` unchanged:

* AWQ  triplet  qweight u32[K, N/8] (AWQ nibble order), qzeros u32[G, N/8], scales f16[G, N]
  (/root/reference/src/loader/safetensors/awq.rs:3-6)
* GPTQ 5-tuple  qweight u32[K/8, N], qzeros u32[G, N/8], scales f16[G, N], g_idx i32[K]?, bias f16[N]?
  (/root/reference/src/loader/safetensors/gptq.rs:3-8)
* GGUF          raw ggml blocks per row (/root/reference/src/loader/gguf.rs:33 VarMap::from_gguf)
* dense         [N, K] f16 / bf16(as uint16) / f32 (/root/reference/src/loader/safetensors/regular.rs:89-117)
"""
import zlib

import numpy as np

PRESETS = {
    # BASELINE.json configs[1]: Llama-3-8B AWQ INT4 gs=128
    "llama3-8b-awq": dict(arch="llama", hidden=4096, n_layers=32, n_heads=32, n_kv_heads=8, head_dim=128, inter=14336,
                          vocab=128256, rms_eps=1e-5, rope_theta=500000.0, act_dtype="f16", quant="awq", group_size=128,
                          max_seq_len=2048, tie_embeddings=False),
    # same layer shapes, 2 layers, small vocab: full-width kernels at test cost
    "llama3-8b-awq-2l": dict(arch="llama", hidden=4096, n_layers=2, n_heads=32, n_kv_heads=8, head_dim=128, inter=14336,
                             vocab=8192, rms_eps=1e-5, rope_theta=500000.0, act_dtype="f16", quant="awq", group_size=128,
                             max_seq_len=512, tie_embeddings=False),
    "tiny-awq": dict(arch="llama", hidden=256, n_layers=2, n_heads=4, n_kv_heads=2, head_dim=64, inter=512, vocab=1024,
                     rms_eps=1e-5, rope_theta=10000.0, act_dtype="f16", quant="awq", group_size=128, max_seq_len=256,
                     tie_embeddings=False),
    "tiny-gptq": dict(arch="llama", hidden=256, n_layers=2, n_heads=4, n_kv_heads=2, head_dim=64, inter=512, vocab=1024,
                      rms_eps=1e-5, rope_theta=10000.0, act_dtype="f16", quant="gptq", group_size=128, max_seq_len=256,
                      tie_embeddings=False),
    # BASELINE.json configs[0]: Llama-3.2-1B bf16 (tied embeddings)
    "llama3.2-1b-bf16": dict(arch="llama", hidden=2048, n_layers=16, n_heads=32, n_kv_heads=8, head_dim=64, inter=8192,
                             vocab=128256, rms_eps=1e-5, rope_theta=500000.0, act_dtype="bf16", quant="none",
                             max_seq_len=2048, tie_embeddings=True,
                             rope_scaling=dict(type="llama3", factor=32.0, low_freq_factor=1.0, high_freq_factor=4.0,
                                               original_max_position_embeddings=8192)),
    "tiny-bf16": dict(arch="llama", hidden=256, n_layers=2, n_heads=4, n_kv_heads=2, head_dim=64, inter=512, vocab=1024,
                      rms_eps=1e-5, rope_theta=10000.0, act_dtype="bf16", quant="none", max_seq_len=256,
                      tie_embeddings=True,
                      rope_scaling=dict(type="llama3", factor=8.0, low_freq_factor=1.0, high_freq_factor=4.0,
                                        original_max_position_embeddings=64)),
    # BASELINE.json configs[2]: Mistral-7B GGUF Q4_K_M (f32 activations, gguf.rs:305)
    "mistral-7b-q4km": dict(arch="llama", hidden=4096, n_layers=32, n_heads=32, n_kv_heads=8, head_dim=128, inter=14336,
                            vocab=32000, rms_eps=1e-5, rope_theta=10000.0, act_dtype="f32", quant="q4_k_m",
                            max_seq_len=2048, tie_embeddings=False, rope_interleaved=1),
    "tiny-q4km": dict(arch="llama", hidden=256, n_layers=8, n_heads=4, n_kv_heads=2, head_dim=64, inter=512, vocab=1024,
                      rms_eps=1e-5, rope_theta=10000.0, act_dtype="f32", quant="q4_k_m", max_seq_len=256,
                      tie_embeddings=False, rope_interleaved=1),
    "tiny-q8_0": dict(arch="llama", hidden=256, n_layers=2, n_heads=4, n_kv_heads=2, head_dim=64, inter=512, vocab=1024,
                      rms_eps=1e-5, rope_theta=10000.0, act_dtype="f32", quant="q8_0", max_seq_len=256,
                      tie_embeddings=False, rope_interleaved=1),
}

MAMBA_PRESETS = {
    # BASELINE.json configs[3]: Mamba2-2.7B (SURVEY.md 8d cfg 4)
    "mamba2-2.7b": dict(arch="mamba2", hidden=2560, n_layers=64, vocab=50288, d_inner=5120, n_heads=80, head_dim=64, d_state=128,
                        n_groups=1, conv_kernel=4, rms_eps=1e-5, act_dtype="bf16", tie_embeddings=True, max_seq_len=1 << 20),
    "tiny-mamba2": dict(arch="mamba2", hidden=256, n_layers=3, vocab=1024, d_inner=512, n_heads=8, head_dim=64, d_state=128,
                        n_groups=1, conv_kernel=4, rms_eps=1e-5, act_dtype="bf16", tie_embeddings=True, max_seq_len=1 << 20),
    "tiny-mamba2-g2": dict(arch="mamba2", hidden=256, n_layers=2, vocab=1024, d_inner=512, n_heads=8, head_dim=64, d_state=64,
                           n_groups=2, conv_kernel=4, rms_eps=1e-5, act_dtype="f32", tie_embeddings=False, max_seq_len=1 << 20),
}

DSV2_PRESETS = {
    # BASELINE.json configs[4]: DeepSeek-V2-Lite (SURVEY.md 8d cfg 5): MLA (kv_lora 512, nope 128 + rope 64, v 128), first layer dense,
    # 26 MoE layers with 64 routed top-6 + 2 shared experts
    "deepseek-v2-lite": dict(arch="deepseek2", hidden=2048, n_layers=27, n_heads=16, vocab=102400, max_seq_len=4096, kv_lora_rank=512,
                             q_lora_rank=0, nope_dim=128, rope_dim=64, v_dim=128, inter=10944, n_experts=64, top_k=6, n_shared=2,
                             moe_inter=1408, first_dense=1, routed_scale=1.0, norm_topk=False, rms_eps=1e-6, act_dtype="bf16",
                             rope_theta=10000.0, tie_embeddings=False),
    "tiny-dsv2": dict(arch="deepseek2", hidden=256, n_layers=3, n_heads=4, vocab=1024, max_seq_len=256, kv_lora_rank=128, q_lora_rank=0,
                      nope_dim=64, rope_dim=32, v_dim=64, inter=512, n_experts=8, top_k=3, n_shared=2, moe_inter=128, first_dense=1,
                      routed_scale=1.0, norm_topk=False, rms_eps=1e-6, act_dtype="bf16", rope_theta=10000.0, tie_embeddings=False),
    "tiny-dsv2-f32": dict(arch="deepseek2", hidden=256, n_layers=2, n_heads=4, vocab=1024, max_seq_len=256, kv_lora_rank=64, q_lora_rank=0,
                          nope_dim=64, rope_dim=32, v_dim=32, inter=512, n_experts=8, top_k=2, n_shared=1, moe_inter=128, first_dense=0,
                          routed_scale=2.0, norm_topk=True, rms_eps=1e-6, act_dtype="f32", rope_theta=10000.0, tie_embeddings=True),
}

GGML_Q8_0, GGML_Q4_K, GGML_Q6_K = 8, 12, 14
BASE_SEED = 0xB1A2


def _rng(name, seed=BASE_SEED):
    return np.random.Generator(np.random.PCG64([seed, zlib.crc32(name.encode())]))


def f32_to_bf16_bits(x):
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)


def bf16_bits_to_f32(b):
    return (np.ascontiguousarray(b, dtype=np.uint16).astype(np.uint32) << 16).view(np.float32)


def _normal(rng, shape, std, chunk=1 << 24):
    n = int(np.prod(shape))
    out = np.empty(n, dtype=np.float32)
    for i in range(0, n, chunk):
        m = min(chunk, n - i)
        out[i:i + m] = rng.standard_normal(m, dtype=np.float32) * std
    return out.reshape(shape)


def _store(x, dtype):
    """f32 array -> storage array for dtype name ('f16' -> float16, 'bf16' -> uint16 bits, 'f32')."""
    if dtype == "f16":
        return x.astype(np.float16)
    if dtype == "bf16":
        return f32_to_bf16_bits(x)
    return x.astype(np.float32)


def _repr(x, dtype):
    """f32 array rounded to values representable in dtype."""
    if dtype == "f16":
        return x.astype(np.float16).astype(np.float32)
    if dtype == "bf16":
        return bf16_bits_to_f32(f32_to_bf16_bits(x))
    return x.astype(np.float32)


def awq_linear(name, N, K, gs, seed=BASE_SEED):
    r = _rng(name, seed)
    G = K // gs
    qweight = r.integers(0, 1 << 32, size=(K, N // 8), dtype=np.uint32)
    qzeros = r.integers(0, 1 << 32, size=(G, N // 8), dtype=np.uint32)
    scales = (np.abs(r.standard_normal((G, N), dtype=np.float32)) * (0.02 / 8) + 1e-4).astype(np.float16)
    return dict(kind="awq", N=N, K=K, group_size=gs, qweight=qweight, qzeros=qzeros, scales=scales)


def gptq_linear(name, N, K, gs, act_order=False, bias=False, seed=BASE_SEED):
    r = _rng(name, seed)
    G = K // gs
    qweight = r.integers(0, 1 << 32, size=(K // 8, N), dtype=np.uint32)
    qzeros = r.integers(0, 1 << 32, size=(G, N // 8), dtype=np.uint32)
    scales = (np.abs(r.standard_normal((G, N), dtype=np.float32)) * (0.02 / 8) + 1e-4).astype(np.float16)
    d = dict(kind="gptq", N=N, K=K, group_size=gs, qweight=qweight, qzeros=qzeros, scales=scales, g_idx=None, bias=None)
    if act_order:
        d["g_idx"] = r.permutation(np.arange(K, dtype=np.int32) // gs).astype(np.int32)
    if bias:
        d["bias"] = (r.standard_normal(N, dtype=np.float32) * 0.01).astype(np.float16)
    return d


def dense_linear(name, N, K, dtype, std=0.02, seed=BASE_SEED):
    return dict(kind="dense", N=N, K=K, weight=_store(_normal(_rng(name, seed), (N, K), std), dtype))


def gguf_blocks(name, ggml_type, N, K, seed=BASE_SEED):
    """Random but well-conditioned ggml blocks: rows of K weights, N rows -> uint8 [N, row_bytes]."""
    r = _rng(name, seed)
    if ggml_type == GGML_Q8_0:
        nb = K // 32
        blk = np.zeros((N, nb, 34), dtype=np.uint8)
        d = r.uniform(1e-4, 3e-4, size=(N, nb)).astype(np.float16)
        blk[:, :, 0:2] = d.view(np.uint8).reshape(N, nb, 2)
        blk[:, :, 2:] = r.integers(0, 256, size=(N, nb, 32), dtype=np.uint8)
    elif ggml_type == GGML_Q4_K:
        nb = K // 256
        blk = np.zeros((N, nb, 144), dtype=np.uint8)
        d = r.uniform(0.6e-4, 1.2e-4, size=(N, nb)).astype(np.float16)  # d * sc(<=63) * q(<=15)
        dmin = r.uniform(1e-4, 2e-4, size=(N, nb)).astype(np.float16)
        blk[:, :, 0:2] = d.view(np.uint8).reshape(N, nb, 2)
        blk[:, :, 2:4] = dmin.view(np.uint8).reshape(N, nb, 2)
        blk[:, :, 4:] = r.integers(0, 256, size=(N, nb, 140), dtype=np.uint8)
    elif ggml_type == GGML_Q6_K:
        nb = K // 256
        blk = np.zeros((N, nb, 210), dtype=np.uint8)
        blk[:, :, :208] = r.integers(0, 256, size=(N, nb, 208), dtype=np.uint8)
        d = r.uniform(1e-5, 2e-5, size=(N, nb)).astype(np.float16)      # d * sc(int8) * q(-32..31)
        blk[:, :, 208:210] = d.view(np.uint8).reshape(N, nb, 2)
    else:
        raise ValueError("unsupported ggml type %r" % ggml_type)
    return dict(kind="gguf", N=N, K=K, ggml_type=ggml_type, blocks=blk.reshape(N, -1))


def q4km_uses_q6k(layer, n_layers):
    """llama.cpp Q4_K_M `use_more_bits` layer subset (SURVEY.md 8d cfg 3) for attn_v / ffn_down."""
    return layer < n_layers // 8 or layer >= 7 * n_layers // 8 or (layer - n_layers // 8) % 3 == 2


def _linear(cfg, name, N, K, layer=None, role=None, seed=BASE_SEED):
    q = cfg["quant"]
    if q == "awq":
        return awq_linear(name, N, K, cfg["group_size"], seed)
    if q == "gptq":
        return gptq_linear(name, N, K, cfg["group_size"], act_order=cfg.get("act_order", False),
                           bias=cfg.get("bias", False), seed=seed)
    if q == "q8_0":
        return gguf_blocks(name, GGML_Q8_0, N, K, seed)
    if q == "q4_k_m":
        six = role in ("v", "down") and q4km_uses_q6k(layer, cfg["n_layers"])
        return gguf_blocks(name, GGML_Q6_K if six else GGML_Q4_K, N, K, seed)
    return dense_linear(name, N, K, cfg["act_dtype"], seed=seed)


def llama_layer(cfg, i, seed=BASE_SEED):
    H, I = cfg["hidden"], cfg["inter"]
    nq, nkv, hd = cfg["n_heads"], cfg["n_kv_heads"], cfg["head_dim"]
    act = cfg["act_dtype"]
    p = "model.layers.%d." % i
    lay = {}
    lay["attn_norm"] = _repr(1.0 + _normal(_rng(p + "input_layernorm.weight", seed), (H,), 0.02), act)
    lay["ffn_norm"] = _repr(1.0 + _normal(_rng(p + "post_attention_layernorm.weight", seed), (H,), 0.02), act)
    lay["q"] = _linear(cfg, p + "self_attn.q_proj", nq * hd, H, i, "q", seed)
    lay["k"] = _linear(cfg, p + "self_attn.k_proj", nkv * hd, H, i, "k", seed)
    lay["v"] = _linear(cfg, p + "self_attn.v_proj", nkv * hd, H, i, "v", seed)
    lay["o"] = _linear(cfg, p + "self_attn.o_proj", H, nq * hd, i, "o", seed)
    lay["gate"] = _linear(cfg, p + "mlp.gate_proj", I, H, i, "gate", seed)
    lay["up"] = _linear(cfg, p + "mlp.up_proj", I, H, i, "up", seed)
    lay["down"] = _linear(cfg, p + "mlp.down_proj", H, I, i, "down", seed)
    return lay


def make_config(preset, **over):
    cfg = dict(PRESETS[preset]) if isinstance(preset, str) else dict(preset)
    cfg.update(over)
    cfg.setdefault("rope_interleaved", 0)
    cfg.setdefault("rope_scaling", None)
    return cfg


def llama_head(cfg, seed=BASE_SEED):
    """(embed storage array, final_norm f32, lm_head spec)."""
    H, V = cfg["hidden"], cfg["vocab"]
    act = cfg["act_dtype"]
    q = cfg["quant"]
    emb_dt = act if act != "f32" else "f32"
    embed = _store(_normal(_rng("model.embed_tokens.weight", seed), (V, H), 0.02), emb_dt)
    final_norm = _repr(1.0 + _normal(_rng("model.norm.weight", seed), (H,), 0.02), act)
    if cfg.get("tie_embeddings"):
        lm = dict(kind="dense", N=V, K=H, weight=embed)
    elif q in ("q4_k_m",):
        lm = gguf_blocks("lm_head", GGML_Q6_K, V, H, seed)       # llama.cpp: output.weight is Q6_K in Q4_K_M
    elif q == "q8_0":
        lm = gguf_blocks("lm_head", GGML_Q8_0, V, H, seed)
    else:
        lm = dense_linear("lm_head", V, H, emb_dt, seed=seed)
    return embed, final_norm, lm


def make_llama(preset, seed=BASE_SEED, **over):
    """Whole model as a dict of numpy arrays (host memory). Use the layer-wise functions for big models
    when only the device copy is needed."""
    cfg = make_config(preset, **over)
    embed, final_norm, lm = llama_head(cfg, seed)
    return dict(config=cfg, embed=embed, final_norm=final_norm, lm_head=lm,
                layers=[llama_layer(cfg, i, seed) for i in range(cfg["n_layers"])])


def prompt_tokens(n, vocab, seed=7):
    return np.random.Generator(np.random.PCG64(seed)).integers(0, vocab, size=n, dtype=np.int64)


def algorithmic_bytes_per_token(cfg):
    """minimal on-disk bytes one decoded token must stream (batch 1, short context)."""
    H, I, V, L = cfg["hidden"], cfg["inter"], cfg["vocab"], cfg["n_layers"]
    nq, nkv, hd = cfg["n_heads"], cfg["n_kv_heads"], cfg["head_dim"]
    lin = L * (H * nq * hd + 2 * H * nkv * hd + nq * hd * H + 3 * H * I)
    q = cfg["quant"]
    act_b = {"f16": 2, "bf16": 2, "f32": 4}[cfg["act_dtype"]]
    if q in ("awq", "gptq"):
        gs = cfg["group_size"]
        wb = lin // 2 + lin // gs * 2 + lin // gs // 2
        head = V * H * 2
    elif q == "none":
        wb = lin * act_b
        head = V * H * act_b
    elif q == "q8_0":
        wb = lin // 32 * 34
        head = V * H // 32 * 34
    elif q == "q4_k_m":
        six = sum(q4km_uses_q6k(i, L) for i in range(L))
        per_q6 = H * nkv * hd + H * I
        n6 = six * per_q6
        wb = (lin - n6) // 256 * 144 + n6 // 256 * 210
        head = V * H // 256 * 210
    else:
        raise ValueError(q)
    norms = (2 * L + 1) * H * act_b
    emb_row = H * act_b
    return wb + head + norms + emb_row


# ---------------------------------------------------------------------------------------------------------
# Mamba2 (HF Mamba2 tensor names; BASELINE.json configs[3])
# ---------------------------------------------------------------------------------------------------------
def mamba2_layer(cfg, i, seed=BASE_SEED):
    D, DI, NH, NS, G, KC = cfg["hidden"], cfg["d_inner"], cfg["n_heads"], cfg["d_state"], cfg["n_groups"], cfg["conv_kernel"]
    act = cfg["act_dtype"]
    conv_dim = DI + 2 * G * NS
    d_in = 2 * DI + 2 * G * NS + NH
    p = "backbone.layers.%d." % i
    r = _rng(p + "mixer.misc", seed)
    lay = {}
    lay["norm"] = _repr(1.0 + _normal(_rng(p + "norm.weight", seed), (D,), 0.02), act)
    lay["in_proj"] = dense_linear(p + "mixer.in_proj", d_in, D, act, seed=seed)
    lay["conv_w"] = _repr(_normal(_rng(p + "mixer.conv1d.weight", seed), (conv_dim, KC), 0.3), act)
    lay["conv_b"] = _repr(_normal(_rng(p + "mixer.conv1d.bias", seed), (conv_dim,), 0.05), act)
    lay["dt_bias"] = _repr(r.uniform(-4.0, -1.0, NH).astype(np.float32), act)       # softplus -> dt in ~[0.02, 0.3]
    lay["A_log"] = _repr(np.log(r.uniform(1.0, 16.0, NH)).astype(np.float32), act)
    lay["D"] = _repr(r.uniform(0.5, 1.5, NH).astype(np.float32), act)
    lay["gnorm"] = _repr(1.0 + _normal(_rng(p + "mixer.norm.weight", seed), (DI,), 0.02), act)
    lay["out_proj"] = dense_linear(p + "mixer.out_proj", D, DI, act, seed=seed)
    return lay


def mamba2_head(cfg, seed=BASE_SEED):
    D, V, act = cfg["hidden"], cfg["vocab"], cfg["act_dtype"]
    embed = _store(_normal(_rng("backbone.embeddings.weight", seed), (V, D), 0.02), act)
    final_norm = _repr(1.0 + _normal(_rng("backbone.norm_f.weight", seed), (D,), 0.02), act)
    lm = dict(kind="dense", N=V, K=D, weight=embed) if cfg.get("tie_embeddings") else dense_linear("lm_head", V, D, act, seed=seed)
    return embed, final_norm, lm


def make_mamba_config(preset, **over):
    cfg = dict(MAMBA_PRESETS[preset]) if isinstance(preset, str) else dict(preset)
    cfg.update(over)
    return cfg


def make_mamba2(preset, seed=BASE_SEED, **over):
    cfg = make_mamba_config(preset, **over)
    embed, final_norm, lm = mamba2_head(cfg, seed)
    return dict(config=cfg, embed=embed, final_norm=final_norm, lm_head=lm, layers=[mamba2_layer(cfg, i, seed) for i in range(cfg["n_layers"])])


def mamba2_bytes_per_token(cfg):
    """weights streamed + recurrent state read and written per decoded token (SURVEY.md 8d cfg 4)"""
    D, DI, NH, HD, NS, G, KC, V, L = (cfg[k] for k in ("hidden", "d_inner", "n_heads", "head_dim", "d_state", "n_groups", "conv_kernel", "vocab", "n_layers"))
    b = {"f16": 2, "bf16": 2, "f32": 4}[cfg["act_dtype"]]
    conv_dim, d_in = DI + 2 * G * NS, 2 * DI + 2 * G * NS + NH
    per_layer = (d_in * D + D * DI) * b + (conv_dim * (KC + 1) + 3 * NH + DI + D) * b
    weights = L * per_layer + V * D * b + D * b + D * b
    state = L * (NH * HD * NS * b * 2 + conv_dim * (KC - 1) * 4 * 2)
    return weights, state


# ---- DeepSeek-V2 family (MLA + MoE) ---------------------------------------------------------------------------------
def make_dsv2_config(preset, **over):
    cfg = dict(DSV2_PRESETS[preset]) if isinstance(preset, str) else dict(preset)
    cfg.update(over)
    return cfg


def _dsv2_mlp(prefix, H, I, act, seed):
    return dict(gate=dense_linear(prefix + "gate_proj", I, H, act, seed=seed), up=dense_linear(prefix + "up_proj", I, H, act, seed=seed),
                down=dense_linear(prefix + "down_proj", H, I, act, seed=seed))


def dsv2_layer(cfg, i, seed=BASE_SEED):
    """HF DeepSeek-V2 tensor names: self_attn.{q_proj, kv_a_proj_with_mqa, kv_a_layernorm, kv_b_proj, o_proj}, mlp.{gate, experts.N, shared_experts}"""
    H, NH, R, DN, DR, DV, act = cfg["hidden"], cfg["n_heads"], cfg["kv_lora_rank"], cfg["nope_dim"], cfg["rope_dim"], cfg["v_dim"], cfg["act_dtype"]
    QL = int(cfg.get("q_lora_rank") or 0)
    p = "model.layers.%d." % i
    lay = dict(attn_norm=_repr(1.0 + _normal(_rng(p + "input_layernorm.weight", seed), (H,), 0.02), act),
               ffn_norm=_repr(1.0 + _normal(_rng(p + "post_attention_layernorm.weight", seed), (H,), 0.02), act),
               kv_norm=_repr(1.0 + _normal(_rng(p + "self_attn.kv_a_layernorm.weight", seed), (R,), 0.02), act),
               # q_lora_rank > 0 (DeepSeek-V2 full, gguf.rs:188-196): q = q_b_proj(q_a_layernorm(q_a_proj(x))); "q_proj" then holds q_a_proj
               q_proj=dense_linear(p + ("self_attn.q_a_proj" if QL else "self_attn.q_proj"), QL if QL else NH * (DN + DR), H, act, seed=seed),
               kv_a=dense_linear(p + "self_attn.kv_a_proj_with_mqa", R + DR, H, act, seed=seed),
               kv_b=dense_linear(p + "self_attn.kv_b_proj", NH * (DN + DV), R, act, std=0.05, seed=seed),
               o=dense_linear(p + "self_attn.o_proj", H, NH * DV, act, seed=seed))
    if QL:
        lay["q_norm"] = _repr(1.0 + _normal(_rng(p + "self_attn.q_a_layernorm.weight", seed), (QL,), 0.02), act)
        lay["q_b"] = dense_linear(p + "self_attn.q_b_proj", NH * (DN + DR), QL, act, std=0.05, seed=seed)
    lay["is_moe"] = i >= cfg["first_dense"] and cfg["n_experts"] > 0
    if not lay["is_moe"]:
        lay.update(_dsv2_mlp(p + "mlp.", H, cfg["inter"], act, seed))
    else:
        lay["router"] = dense_linear(p + "mlp.gate", cfg["n_experts"], H, act, std=0.05, seed=seed)
        lay["experts"] = [_dsv2_mlp(p + "mlp.experts.%d." % e, H, cfg["moe_inter"], act, seed) for e in range(cfg["n_experts"])]
        if cfg["n_shared"] > 0:
            lay["shared"] = _dsv2_mlp(p + "mlp.shared_experts.", H, cfg["n_shared"] * cfg["moe_inter"], act, seed)
    return lay


def dsv2_head(cfg, seed=BASE_SEED):
    H, V, act = cfg["hidden"], cfg["vocab"], cfg["act_dtype"]
    embed = _store(_normal(_rng("model.embed_tokens.weight", seed), (V, H), 0.02), act)
    final_norm = _repr(1.0 + _normal(_rng("model.norm.weight", seed), (H,), 0.02), act)
    lm = dict(kind="dense", N=V, K=H, weight=embed) if cfg.get("tie_embeddings") else dense_linear("lm_head", V, H, act, seed=seed)
    return embed, final_norm, lm


def make_dsv2(preset, seed=BASE_SEED, **over):
    cfg = make_dsv2_config(preset, **over)
    embed, final_norm, lm = dsv2_head(cfg, seed)
    return dict(config=cfg, embed=embed, final_norm=final_norm, lm_head=lm, layers=[dsv2_layer(cfg, i, seed) for i in range(cfg["n_layers"])])


def dsv2_bytes_per_token(cfg):
    """active weight bytes one decoded token streams (SURVEY.md 8d cfg 5: 4.90 GB for DeepSeek-V2-Lite); latent cache excluded"""
    H, NH, R, DN, DR, DV, V, L = (cfg[k] for k in ("hidden", "n_heads", "kv_lora_rank", "nope_dim", "rope_dim", "v_dim", "vocab", "n_layers"))
    b = {"f16": 2, "bf16": 2, "f32": 4}[cfg["act_dtype"]]
    QL = int(cfg.get("q_lora_rank") or 0)
    qp = (QL * H + NH * (DN + DR) * QL) if QL else NH * (DN + DR) * H
    attn = qp + (R + DR) * H + NH * (DN + DV) * R + H * NH * DV
    dense = 3 * H * cfg["inter"]
    moe = cfg["n_experts"] * H + (cfg["top_k"] + cfg["n_shared"]) * 3 * H * cfg["moe_inter"]
    n_moe = sum(1 for i in range(L) if i >= cfg["first_dense"] and cfg["n_experts"] > 0)
    params = L * attn + (L - n_moe) * dense + n_moe * moe + V * H
    norms = L * (2 * H + R + QL) + H
    return (params + norms + H) * b
