"""Host-side mirror of the boostr surface blazr's engine calls, over the C-ABI.

Names and argument meaning follow the reference so that parity tests read like the reference's call sites:

* ``LoadedModel.forward_with_kv_cache(input, kv, position)``      /root/reference/src/engine/executor_generate.rs:357,372
* ``LoadedModel.forward_with_paged_kv_cache(input, cache, slot_mapping, block_table, seq_len_k, start_pos)``  :259-262
* ``LoadedModel.forward_embed / forward_layers_range / forward_head``  /root/reference/src/cli/swarm_forward.rs:205,239-263
* ``LayeredKvCache.new_positional(...)``, ``LayeredPagedKvCache(...)``  executor_generate.rs:350-353, :208-210
* ``logits_to_token(...)``  /root/reference/src/engine/sampling.rs:445-460
* ``Executor.generate``  executor_generate.rs:341-410 (the loop itself runs in C++: ``bz_generate``)

All compute happens in libblazr_hip.so on the GPU; numpy arrays are only the host view of inputs/outputs.
"""
import ctypes as C
import os

import numpy as np

from . import _lib as L

_NP2BZ = {np.dtype(np.float32): L.F32, np.dtype(np.float16): L.F16, np.dtype(np.int64): L.I64, np.dtype(np.int32): L.I32,
          np.dtype(np.uint32): L.U32, np.dtype(np.uint8): L.U8}
_BZ2NP = {L.F32: np.float32, L.F16: np.float16, L.BF16: np.uint16, L.I64: np.int64, L.I32: np.int32, L.U32: np.uint32, L.U8: np.uint8}
_DT = {"f32": L.F32, "f16": L.F16, "bf16": L.BF16}


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Device:
    """boostr::CudaDevice::new(id) + CudaClient (cli/run.rs:70-81)."""

    def __init__(self, device_id=0):
        h = C.c_void_p()
        L.check(L.lib().bz_device_open(device_id, C.byref(h)))
        self.h = h

    def close(self):
        if self.h:
            L.lib().bz_device_close(self.h)
            self.h = None

    def synchronize(self):
        L.check(L.lib().bz_device_synchronize(self.h))

    def memory_info(self):
        f, t = C.c_size_t(), C.c_size_t()
        L.check(L.lib().bz_device_memory_info(self.h, C.byref(f), C.byref(t)))
        return f.value, t.value

    def name(self):
        b = C.create_string_buffer(256)
        L.check(L.lib().bz_device_name(self.h, b, 256))
        return b.value.decode()

    def stream(self):
        return L.lib().bz_device_stream(self.h)

    # Tensor::from_slice / zeros
    def tensor(self, array, dtype=None):
        a = np.ascontiguousarray(array)
        dt = dtype if dtype is not None else _NP2BZ[a.dtype]
        shape = (C.c_int64 * max(a.ndim, 1))(*(a.shape if a.ndim else (1,)))
        h = C.c_void_p()
        L.check(L.lib().bz_tensor_from_host(self.h, dt, shape, max(a.ndim, 1), _ptr(a), C.byref(h)))
        return Tensor(self, h, dt, tuple(a.shape) if a.ndim else (1,))

    def zeros(self, shape, dtype=L.F32):
        shape = tuple(int(s) for s in shape)
        sh = (C.c_int64 * len(shape))(*shape)
        h = C.c_void_p()
        L.check(L.lib().bz_tensor_zeros(self.h, dtype, sh, len(shape), C.byref(h)))
        return Tensor(self, h, dtype, shape)

    def record_event(self):
        ev = C.c_uint64()
        L.check(L.lib().bz_event_record(self.h, C.byref(ev)))
        return ev.value


class Tensor:
    def __init__(self, dev, h, dtype, shape):
        self.dev, self.h, self.dtype, self.shape = dev, h, dtype, shape

    def __del__(self):
        try:
            if self.h and L.alive:
                L.lib().bz_tensor_free(self.h)
        except Exception:
            pass

    def to_numpy(self):
        out = np.empty(self.shape, dtype=_BZ2NP[self.dtype])
        L.check(L.lib().bz_tensor_to_host(self.h, _ptr(out), out.nbytes))
        return out

    to_vec = to_numpy

    def copy_from(self, array):
        a = np.ascontiguousarray(array, dtype=_BZ2NP[self.dtype])
        L.check(L.lib().bz_tensor_copy_from_host(self.h, _ptr(a), a.nbytes))

    def to_vec_pipelined(self, event):
        out = np.empty(self.shape, dtype=_BZ2NP[self.dtype])
        L.check(L.lib().bz_tensor_to_host_pipelined(self.h, event, _ptr(out), out.nbytes))
        return out


def make_config(cfg):
    """synth/HF-style config dict -> bz_model_config POD."""
    c = L.ModelConfig()
    c.abi_version = L.ABI_VERSION
    if cfg.get("arch") == "mamba2":
        # SsmConfig (loader/gguf.rs:219-262)
        c.arch = L.ARCH_MAMBA2
        for k in ("hidden", "n_layers", "vocab", "max_seq_len"):
            setattr(c, k, int(cfg[k]))
        for k in ("d_inner", "n_heads", "head_dim", "d_state", "n_groups", "conv_kernel"):
            setattr(c, "ssm_" + k, int(cfg[k]))
        c.rms_eps = cfg["rms_eps"]
        c.act_dtype = _DT[cfg["act_dtype"]]
        c.tie_embeddings = int(bool(cfg.get("tie_embeddings")))
        return c
    if cfg.get("arch") == "deepseek2":
        # AttentionConfig kv_latent_dim / q_latent_dim / d_rope (loader/gguf.rs:188-196), MoeConfig (loader/gguf.rs:271-283)
        c.arch = L.ARCH_DEEPSEEK2
        for k in ("hidden", "n_layers", "n_heads", "vocab", "max_seq_len", "inter"):
            setattr(c, k, int(cfg[k]))
        for k in ("kv_lora_rank", "q_lora_rank", "nope_dim", "rope_dim", "v_dim"):
            setattr(c, "mla_" + k, int(cfg[k]))
        c.moe_n_experts, c.moe_top_k, c.moe_n_shared = int(cfg["n_experts"]), int(cfg["top_k"]), int(cfg["n_shared"])
        c.moe_inter, c.moe_first_dense, c.moe_norm_topk = int(cfg["moe_inter"]), int(cfg["first_dense"]), int(bool(cfg["norm_topk"]))
        c.moe_routed_scale = float(cfg["routed_scale"])
        c.rms_eps = cfg["rms_eps"]
        c.act_dtype = _DT[cfg["act_dtype"]]
        c.tie_embeddings = int(bool(cfg.get("tie_embeddings")))
        c.rope_theta = cfg["rope_theta"]
        c.rope_interleaved = 1
        _rope_scaling_to_config(cfg, c)
        return c
    c.arch = L.ARCH_LLAMA
    for k in ("hidden", "n_layers", "n_heads", "n_kv_heads", "head_dim", "inter", "vocab", "max_seq_len"):
        setattr(c, k, int(cfg[k]))
    c.rms_eps = cfg["rms_eps"]
    c.act_dtype = _DT[cfg["act_dtype"]]
    c.tie_embeddings = int(bool(cfg.get("tie_embeddings")))
    c.rope_theta = cfg["rope_theta"]
    c.rope_interleaved = int(cfg.get("rope_interleaved", 0))
    _rope_scaling_to_config(cfg, c)
    return c


def yarn_mscale(factor, mscale=1.0):
    """HF yarn_get_mscale"""
    import math
    return 1.0 if factor <= 1.0 else 0.1 * mscale * math.log(factor) + 1.0


def _rope_scaling_to_config(cfg, c):
    """rope_scaling dict (HF config.json names) -> the bz_model_config fields (loader/safetensors/config.rs:83-95)"""
    rs = cfg.get("rope_scaling") or {}
    c.rope_scaling = {"none": L.ROPE_NONE, "linear": L.ROPE_LINEAR, "llama3": L.ROPE_LLAMA3, "yarn": L.ROPE_YARN}[rs.get("type", "none")]
    c.rope_factor = rs.get("factor", 1.0)
    c.rope_low_freq_factor = rs.get("low_freq_factor", 1.0)
    c.rope_high_freq_factor = rs.get("high_freq_factor", 4.0)
    c.rope_original_max_pos = rs.get("original_max_position_embeddings", 8192)
    c.rope_beta_fast, c.rope_beta_slow = rs.get("beta_fast", 0.0), rs.get("beta_slow", 0.0)
    c.rope_attn_factor, c.mla_softmax_mscale = 0.0, 0.0
    if rs.get("type") == "yarn":
        if "attention_factor" in rs:
            c.rope_attn_factor = rs["attention_factor"]
        elif "mscale" in rs and "mscale_all_dim" in rs:      # DeepSeek-V2: cos / sin by mscale / mscale_all_dim, softmax scale by mscale_all_dim^2
            c.rope_attn_factor = yarn_mscale(c.rope_factor, rs["mscale"]) / yarn_mscale(c.rope_factor, rs["mscale_all_dim"])
        if cfg.get("arch") == "deepseek2" and rs.get("mscale_all_dim"):
            c.mla_softmax_mscale = yarn_mscale(c.rope_factor, rs["mscale_all_dim"])


_AWQ_SHIFTS = np.array([0, 16, 4, 20, 8, 24, 12, 28], dtype=np.uint32)  # awq.rs:32


def unpack_awq_zeros(qzeros, N):
    """awq.rs:239-263: packed [G, N/8] -> f32 [G, N] (what DecomposedQuantTensor carries)."""
    qz = np.ascontiguousarray(qzeros, dtype=np.uint32)
    return ((qz[:, :, None] >> _AWQ_SHIFTS[None, None, :]) & 0xF).reshape(qz.shape[0], N).astype(np.float32)


class LoadedModel:
    """boostr::model::LoadedModel<R> behind the C-ABI."""

    def __init__(self, dev, cfg):
        self.dev = dev
        self.cfg = dict(cfg)
        self.c = make_config(cfg)
        h = C.c_void_p()
        L.check(L.lib().bz_model_create(dev.h, C.byref(self.c), C.byref(h)))
        self.h = h
        self.finalized = False
        self._shapes = {}

    def __del__(self):
        try:
            if self.h and L.alive:
                L.lib().bz_model_free(self.h)
        except Exception:
            pass

    # --- VarMap::insert / insert_decomposed_quant ---------------------------------------------------------
    def add_dense(self, name, array):
        a = np.ascontiguousarray(array)
        dt = L.BF16 if a.dtype == np.uint16 else _NP2BZ[a.dtype]
        shape = (C.c_int64 * a.ndim)(*a.shape)
        L.check(L.lib().bz_model_add_dense(self.h, name.encode(), dt, shape, a.ndim, _ptr(a)))

    def add_linear(self, base, spec):
        """`spec` in the loader formats of blazr_amd.synth (== what awq.rs / gptq.rs / regular.rs produce)."""
        name = (base + ".weight").encode()
        kind = spec["kind"]
        self._shapes[base + ".weight"] = (int(spec["N"]), int(spec["K"]))
        if kind == "dense":
            self.add_dense(base + ".weight", spec["weight"])
        elif kind == "awq":
            qw = np.ascontiguousarray(spec["qweight"], dtype=np.uint32)
            sc = np.ascontiguousarray(spec["scales"]).astype(np.float32)            # awq.rs:202-206 f16 -> f32
            zf = unpack_awq_zeros(spec["qzeros"], spec["N"])                        # awq.rs:208-213
            L.check(L.lib().bz_model_add_awq(self.h, name, spec["N"], spec["K"], _ptr(qw), _ptr(sc), _ptr(zf), spec["group_size"]))
        elif kind == "gptq":
            qw = np.ascontiguousarray(spec["qweight"], dtype=np.uint32)
            sc = np.ascontiguousarray(spec["scales"]).astype(np.float32)
            qz = np.ascontiguousarray(spec["qzeros"], dtype=np.uint32)              # kept packed (gptq.rs:209-219)
            gi = None if spec.get("g_idx") is None else np.ascontiguousarray(spec["g_idx"], dtype=np.int32)
            bi = None if spec.get("bias") is None else np.ascontiguousarray(spec["bias"]).astype(np.float32)
            L.check(L.lib().bz_model_add_gptq(self.h, name, spec["N"], spec["K"], _ptr(qw), _ptr(sc), _ptr(qz),
                                              None if gi is None else _ptr(gi), None if bi is None else _ptr(bi), spec["group_size"]))
        elif kind == "gguf":
            b = np.ascontiguousarray(spec["blocks"], dtype=np.uint8)
            L.check(L.lib().bz_model_add_gguf(self.h, name, spec["ggml_type"], spec["N"], spec["K"], _ptr(b)))
        else:
            raise ValueError(kind)

    def add_llama_layer(self, i, lay):
        p = "model.layers.%d." % i
        self.add_dense(p + "input_layernorm.weight", np.asarray(lay["attn_norm"], dtype=np.float32))
        self.add_dense(p + "post_attention_layernorm.weight", np.asarray(lay["ffn_norm"], dtype=np.float32))
        for short, hf in (("q", "self_attn.q_proj"), ("k", "self_attn.k_proj"), ("v", "self_attn.v_proj"), ("o", "self_attn.o_proj"),
                          ("gate", "mlp.gate_proj"), ("up", "mlp.up_proj"), ("down", "mlp.down_proj")):
            self.add_linear(p + hf, lay[short])

    def add_llama_head(self, embed, final_norm, lm_head):
        self.add_dense("model.embed_tokens.weight", embed)
        self.add_dense("model.norm.weight", np.asarray(final_norm, dtype=np.float32))
        if not self.cfg.get("tie_embeddings"):
            self.add_linear("lm_head", lm_head)

    def add_dsv2_layer(self, i, lay):
        """HF DeepSeek-V2 tensor names"""
        p = "model.layers.%d." % i
        f32 = lambda a: np.asarray(a, dtype=np.float32)
        self.add_dense(p + "input_layernorm.weight", f32(lay["attn_norm"]))
        self.add_dense(p + "post_attention_layernorm.weight", f32(lay["ffn_norm"]))
        self.add_dense(p + "self_attn.kv_a_layernorm.weight", f32(lay["kv_norm"]))
        ql = "q_b" in lay     # q_lora_rank > 0: q_a_proj -> q_a_layernorm -> q_b_proj
        for short, hf in (("q_proj", "self_attn.q_a_proj" if ql else "self_attn.q_proj"), ("kv_a", "self_attn.kv_a_proj_with_mqa"), ("kv_b", "self_attn.kv_b_proj"),
                          ("o", "self_attn.o_proj")):
            self.add_linear(p + hf, lay[short])
        if ql:
            self.add_dense(p + "self_attn.q_a_layernorm.weight", f32(lay["q_norm"]))
            self.add_linear(p + "self_attn.q_b_proj", lay["q_b"])
        if not lay["is_moe"]:
            for n in ("gate", "up", "down"):
                self.add_linear(p + "mlp.%s_proj" % n, lay[n])
        else:
            self.add_linear(p + "mlp.gate", lay["router"])
            for e, ex in enumerate(lay["experts"]):
                for n in ("gate", "up", "down"):
                    self.add_linear(p + "mlp.experts.%d.%s_proj" % (e, n), ex[n])
            if "shared" in lay:
                for n in ("gate", "up", "down"):
                    self.add_linear(p + "mlp.shared_experts.%s_proj" % n, lay["shared"][n])

    def add_mamba2_layer(self, i, lay):
        """HF Mamba2 tensor names (backbone.layers.{i}.mixer.*)"""
        p = "backbone.layers.%d." % i
        f32 = lambda a: np.asarray(a, dtype=np.float32)
        self.add_dense(p + "norm.weight", f32(lay["norm"]))
        self.add_linear(p + "mixer.in_proj", lay["in_proj"])
        cw = f32(lay["conv_w"])
        self.add_dense(p + "mixer.conv1d.weight", cw.reshape(cw.shape[0], 1, cw.shape[1]))
        self.add_dense(p + "mixer.conv1d.bias", f32(lay["conv_b"]))
        self.add_dense(p + "mixer.dt_bias", f32(lay["dt_bias"]))
        self.add_dense(p + "mixer.A_log", f32(lay["A_log"]))
        self.add_dense(p + "mixer.D", f32(lay["D"]))
        self.add_dense(p + "mixer.norm.weight", f32(lay["gnorm"]))
        self.add_linear(p + "mixer.out_proj", lay["out_proj"])

    def add_mamba2_head(self, embed, final_norm, lm_head):
        self.add_dense("backbone.embeddings.weight", embed)
        self.add_dense("backbone.norm_f.weight", np.asarray(final_norm, dtype=np.float32))
        if not self.cfg.get("tie_embeddings"):
            self.add_linear("lm_head", lm_head)

    @classmethod
    def from_synth(cls, dev, model):
        """Whole in-memory model dict (blazr_amd.synth.make_llama / make_mamba2)."""
        m = cls(dev, model["config"])
        if model["config"].get("arch") == "deepseek2":
            for i, lay in enumerate(model["layers"]):
                m.add_dsv2_layer(i, lay)
            m.add_llama_head(model["embed"], model["final_norm"], model["lm_head"])
            m.finalize()
            return m
        if model["config"].get("arch") == "mamba2":
            for i, lay in enumerate(model["layers"]):
                m.add_mamba2_layer(i, lay)
            m.add_mamba2_head(model["embed"], model["final_norm"], model["lm_head"])
            m.finalize()
            return m
        for i, lay in enumerate(model["layers"]):
            m.add_llama_layer(i, lay)
        m.add_llama_head(model["embed"], model["final_norm"], model["lm_head"])
        m.finalize()
        return m

    @classmethod
    def from_synth_streamed(cls, dev, cfg, seed=None):
        """Generate + upload layer by layer (bounded host memory) -- used for the 8B bench shape."""
        from . import synth
        kw = {} if seed is None else {"seed": seed}
        m = cls(dev, cfg)
        if cfg.get("arch") == "deepseek2":
            for i in range(cfg["n_layers"]):
                m.add_dsv2_layer(i, synth.dsv2_layer(cfg, i, **kw))
            emb, fn, lm = synth.dsv2_head(cfg, **kw)
            m.add_llama_head(emb, fn, lm)
            m.finalize()
            return m
        if cfg.get("arch") == "mamba2":
            for i in range(cfg["n_layers"]):
                m.add_mamba2_layer(i, synth.mamba2_layer(cfg, i, **kw))
            emb, fn, lm = synth.mamba2_head(cfg, **kw)
            m.add_mamba2_head(emb, fn, lm)
            m.finalize()
            return m
        for i in range(cfg["n_layers"]):
            m.add_llama_layer(i, synth.llama_layer(cfg, i, **kw))
        emb, fn, lm = synth.llama_head(cfg, **kw)
        m.add_llama_head(emb, fn, lm)
        m.finalize()
        return m

    def finalize(self):
        L.check(L.lib().bz_model_finalize(self.h))
        self.finalized = True

    # --- accessors (LoadedModel::num_layers etc.) ---------------------------------------------------------
    def num_layers(self):
        return self.c.n_layers

    def num_kv_heads(self):
        return 1 if self.c.arch == L.ARCH_DEEPSEEK2 else self.c.n_kv_heads

    def head_dim(self):
        return self.c.mla_kv_lora_rank + self.c.mla_rope_dim if self.c.arch == L.ARCH_DEEPSEEK2 else self.c.head_dim

    def hidden_size(self):
        return self.c.hidden

    def vocab_size(self):
        return self.c.vocab

    def needs_kv_cache(self):
        return self.c.arch != L.ARCH_MAMBA2

    def needs_ssm_state(self):
        return self.c.arch == L.ARCH_MAMBA2

    def moe_config(self):
        if self.c.arch != L.ARCH_DEEPSEEK2 or self.c.moe_n_experts == 0:
            return None
        return dict(num_experts=self.c.moe_n_experts, experts_per_tok=self.c.moe_top_k, shared_experts=self.c.moe_n_shared)

    def new_kv_cache(self, capacity, max_seq_len=None):
        """LayeredKvCache::new_positional with this model's cache shape (for MLA: one 'head' of kv_lora_rank + rope_dim latents)"""
        dt = self.c.act_dtype
        return LayeredKvCache(self.dev, self.c.n_layers, 1, self.num_kv_heads(), capacity, max_seq_len or self.c.max_seq_len, self.head_dim(), dt)

    def mamba_config(self):
        if self.c.arch != L.ARCH_MAMBA2:
            return None
        return {k: getattr(self.c, "ssm_" + k) for k in ("d_inner", "n_heads", "head_dim", "d_state", "n_groups", "conv_kernel")}

    def weight_bytes(self):
        a, b = C.c_size_t(), C.c_size_t()
        L.check(L.lib().bz_model_weight_bytes(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def rope_caches(self):
        n = self.c.max_seq_len * (self.c.head_dim // 2)
        c, s = np.empty(n, np.float32), np.empty(n, np.float32)
        L.check(L.lib().bz_rope_caches(self.h, _ptr(c), _ptr(s)))
        return c.reshape(self.c.max_seq_len, -1), s.reshape(self.c.max_seq_len, -1)

    # --- forward --------------------------------------------------------------------------------------------
    def _tokens(self, tokens):
        if isinstance(tokens, Tensor):
            return tokens, int(np.prod(tokens.shape))
        t = np.ascontiguousarray(tokens, dtype=np.int64).reshape(-1)
        return self.dev.tensor(t), len(t)

    def forward_with_kv_cache(self, tokens, kv, position, all_logits=False):
        t, S = self._tokens(tokens)
        out = self.dev.zeros((S if all_logits else 1, self.c.vocab), L.F32)
        L.check(L.lib().bz_forward_kv(self.h, t.h, S, kv.h, position, out.h, L.FWD_ALL_LOGITS if all_logits else 0))
        return out

    def forward_with_paged_kv_cache(self, tokens, cache, slot_mapping, block_table, seq_len_k, start_pos, all_logits=False):
        t, S = self._tokens(tokens)
        sm = slot_mapping if isinstance(slot_mapping, Tensor) else self.dev.tensor(np.asarray(slot_mapping, dtype=np.int32))
        bt = block_table if isinstance(block_table, Tensor) else self.dev.tensor(np.asarray(block_table, dtype=np.int32).reshape(-1))
        out = self.dev.zeros((S if all_logits else 1, self.c.vocab), L.F32)
        L.check(L.lib().bz_forward_paged(self.h, t.h, S, cache.h, sm.h, bt.h, int(np.prod(bt.shape)), seq_len_k, start_pos, out.h,
                                         L.FWD_ALL_LOGITS if all_logits else 0))
        return out

    def forward_with_ssm_state(self, tokens, ssm, all_logits=False):
        """LoadedModel::forward_with_ssm_state (executor_generate.rs:137,148)"""
        t, S = self._tokens(tokens)
        out = self.dev.zeros((S if all_logits else 1, self.c.vocab), L.F32)
        L.check(L.lib().bz_forward_ssm(self.h, t.h, S, ssm.h, out.h, L.FWD_ALL_LOGITS if all_logits else 0))
        return out

    def forward_paged_batch(self, tokens, cache, slot_mapping, block_tables, seq_lens):
        """process_decode_batch (batch_decode.rs:35-150): tokens [N], slots [N], block_tables [N][<= max_blocks] (ragged ok), seq_lens [N]"""
        t, n = self._tokens(tokens)
        nb = max(len(b) for b in block_tables)
        bt = np.zeros((n, nb), dtype=np.int32)
        for i, b in enumerate(block_tables):
            bt[i, :len(b)] = b                                    # shorter tables padded with 0 (batch_decode.rs:118-126)
        sm, btt = self.dev.tensor(np.asarray(slot_mapping, dtype=np.int32)), self.dev.tensor(bt)
        sl = np.asarray(seq_lens, dtype=np.int32)
        out = self.dev.zeros((n, self.c.vocab), L.F32)
        L.check(L.lib().bz_forward_paged_batch(self.h, t.h, n, cache.h, sm.h, btt.h, nb, sl.ctypes.data_as(C.c_void_p), out.h))
        return out

    def forward_embed(self, tokens):
        t, S = self._tokens(tokens)
        out = self.dev.zeros((S, self.c.hidden), L.F32)
        L.check(L.lib().bz_forward_embed(self.h, t.h, S, out.h))
        return out

    def forward_layers_range(self, hidden, prev_mlp, kv, start, end, position):
        S = hidden.shape[0]
        has = C.c_int(0 if prev_mlp is None else 1)
        pm = prev_mlp if prev_mlp is not None else self.dev.zeros((S, self.c.hidden), L.F32)
        L.check(L.lib().bz_forward_layers_range(self.h, hidden.h, pm.h, C.byref(has), S, kv.h, start, end, position))
        return hidden, (pm if has.value else None)

    def forward_head(self, hidden, prev_mlp, all_logits=False):
        S = hidden.shape[0]
        out = self.dev.zeros((S if all_logits else 1, self.c.vocab), L.F32)
        L.check(L.lib().bz_forward_head(self.h, hidden.h, None if prev_mlp is None else prev_mlp.h, int(prev_mlp is not None), S, out.h,
                                        L.FWD_ALL_LOGITS if all_logits else 0))
        return out

    def profile_step(self, kv, token, position, iters=4):
        """per-kernel dispatch times of `iters` real decode steps (bz_profile_step)"""
        buf = (L.KernelTime * 32)()
        n = C.c_int()
        L.check(L.lib().bz_profile_step(self.h, kv.h, int(token), int(position), iters, buf, 32, C.byref(n)))
        return [dict(name=buf[i].name.decode(), launches=buf[i].launches, total_ms=buf[i].total_ms, algo_bytes=buf[i].algo_bytes)
                for i in range(n.value)]

    def profile_step_ssm(self, ssm, token, iters=4):
        buf = (L.KernelTime * 32)()
        n = C.c_int()
        L.check(L.lib().bz_profile_step_ssm(self.h, ssm.h, int(token), iters, buf, 32, C.byref(n)))
        return [dict(name=buf[i].name.decode(), launches=buf[i].launches, total_ms=buf[i].total_ms, algo_bytes=buf[i].algo_bytes)
                for i in range(n.value)]

    # --- op-level ----------------------------------------------------------------------------------------------
    def quant_matmul(self, name, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        S = 1 if x.ndim == 1 else x.shape[0]
        xt = self.dev.tensor(x.reshape(S, -1))
        N = self.linear_shape(name)[0]
        y = self.dev.zeros((S, N), L.F32)
        L.check(L.lib().bz_quant_matmul(self.h, name.encode(), xt.h, S, y.h))
        return y.to_numpy()

    def linear_shape(self, name):
        return self._shapes[name]

    def conv1d_step(self, layer, state, zx):
        """ConvOps: the conv1d step (+ SiLU) of Mamba2 layer `layer` on the in_proj row zx; returns xbc [conv_dim]; the layer's conv window advances"""
        c = self.mamba_config()
        zt = self.dev.tensor(np.ascontiguousarray(zx, dtype=np.float32))
        out = self.dev.zeros((c["d_inner"] + 2 * c["n_groups"] * c["d_state"],), L.F32)
        L.check(L.lib().bz_conv1d_step(self.h, layer, state.h, zt.h, out.h))
        return out.to_numpy()

    def ssm_step(self, layer, state, zx):
        """the mixer's fused step (conv1d + SSM recurrence + gate) on the in_proj row zx; returns the gated y [d_inner]; conv window and SSM state advance"""
        c = self.mamba_config()
        zt = self.dev.tensor(np.ascontiguousarray(zx, dtype=np.float32))
        out = self.dev.zeros((c["d_inner"],), L.F32)
        L.check(L.lib().bz_ssm_step(self.h, layer, state.h, zt.h, out.h))
        return out.to_numpy()

    def moe_route(self, layer, hidden):
        """router of DeepSeek MoE layer `layer` on the residual-stream row (norm inside): (sel int32 [top_k + n_shared], w float32 [...], xn float32 [H])"""
        n = self.c.moe_top_k + self.c.moe_n_shared
        ht = self.dev.tensor(np.ascontiguousarray(hidden, dtype=np.float32))
        sel, w, xn = self.dev.zeros((n,), L.I32), self.dev.zeros((n,), L.F32), self.dev.zeros((self.c.hidden,), L.F32)
        L.check(L.lib().bz_moe_route(self.h, layer, ht.h, sel.h, w.h, xn.h))
        return sel.to_numpy(), w.to_numpy(), xn.to_numpy()

    def moe_grouped_gemv(self, layer, which, sel, x):
        """grouped expert GEMV: which 0 -> gate|up of experts sel[] on the shared row x [H]; which 1 -> down (SiLU*up prologue) on x [n_slots, 2 moe_inter]"""
        sel = np.ascontiguousarray(sel, dtype=np.int32)
        st = self.dev.tensor(sel, L.I32)
        xt = self.dev.tensor(np.ascontiguousarray(x, dtype=np.float32))
        N = 2 * self.c.moe_inter if which == 0 else self.c.hidden
        y = self.dev.zeros((len(sel), N), L.F32)
        L.check(L.lib().bz_moe_grouped_gemv(self.h, layer, which, st.h, len(sel), xt.h, y.h))
        return y.to_numpy()

    def dequant(self, name):
        N, K = self._shapes[name]
        out = np.empty((N, K), dtype=np.float32)
        L.check(L.lib().bz_dequant(self.h, name.encode(), _ptr(out)))
        return out


def detect_model_source(path):
    """loader/detect.rs:34-150 -> dict(format 'safetensors' | 'gguf', weights_path, config_path or None)"""
    s = L.ModelSource()
    L.check(L.lib().bz_detect_model_source(os.fsencode(path), C.byref(s)))
    return dict(format="gguf" if s.format == 1 else "safetensors", weights_path=os.fsdecode(s.weights_path),
                config_path=os.fsdecode(s.config_path) if s.has_config else None)


LAYER_TYPES = ["StandardTransformer", "Mamba2", "Mamba3", "MlaWithMoe", "MlaWithMlp"]


def detect_architecture_from_names(names):
    """boostr::model::detection::detect_architecture_from_names (tests: loader/safetensors/detect_arch.rs:200-315)"""
    arr = (C.c_char_p * len(names))(*[n.encode() for n in names])
    d = L.DetectedArch()
    L.check(L.lib().bz_detect_architecture_from_names(arr, len(names), C.byref(d)))
    return dict(format="HuggingFace" if d.format == 0 else "Oxidizr", num_layers=d.num_layers, tie_word_embeddings=bool(d.tie_word_embeddings),
                layer_types=[LAYER_TYPES[d.layer_types[i]] for i in range(d.num_layers)])


def config_from_hf_json(text):
    """HF config.json text -> (bz_model_config POD, dict(quant_method, group_size, torch_dtype))"""
    c, q = L.ModelConfig(), L.QuantInfo()
    L.check(L.lib().bz_config_from_hf_json(text.encode(), C.byref(c), C.byref(q)))
    return c, dict(quant_method={0: None, 1: "awq", 2: "gptq"}[q.quant_method], group_size=q.group_size, torch_dtype=q.torch_dtype)


def config_from_gguf(path):
    c, g = L.ModelConfig(), L.GgufInfo()
    L.check(L.lib().bz_config_from_gguf(os.fsencode(path), C.byref(c), C.byref(g)))
    return c, dict(architecture=g.architecture.decode(), n_tensors=g.n_tensors, version=g.version, dominant_ggml_type=g.dominant_ggml_type,
                   is_mla=bool(g.is_mla), is_moe=bool(g.is_moe), is_ssm=bool(g.is_ssm), file_size_bytes=g.file_size_bytes)


def safetensors_describe(path):
    import json
    n = C.c_size_t()
    L.check(L.lib().bz_safetensors_describe(os.fsencode(path), None, 0, C.byref(n)))
    buf = C.create_string_buffer(n.value)
    L.check(L.lib().bz_safetensors_describe(os.fsencode(path), buf, n.value, None))
    return json.loads(buf.value.decode())


def load_model(dev, path):
    """loaders.rs load_model: checkpoint directory / .safetensors / .gguf -> finalized LoadedModel"""
    h, c = C.c_void_p(), L.ModelConfig()
    L.check(L.lib().bz_load_model(dev.h, os.fsencode(path), C.byref(h), C.byref(c)))
    m = LoadedModel.__new__(LoadedModel)
    m.dev, m.cfg, m.c, m.h, m.finalized, m._shapes = dev, {}, c, h, True, {}
    return m


class LayeredKvCache:
    """boostr::inference::LayeredKvCache::new_positional (executor_generate.rs:350-353)."""

    def __init__(self, dev, layers, batch, n_kv_heads, initial_capacity, max_seq_len, head_dim, dtype):
        h = C.c_void_p()
        L.check(L.lib().bz_kv_create(dev.h, layers, batch, n_kv_heads, initial_capacity, max_seq_len, head_dim, dtype, C.byref(h)))
        self.h, self.dev, self.head_dim = h, dev, head_dim

    new_positional = classmethod(lambda cls, *a: cls(*a))

    def __del__(self):
        try:
            if self.h and L.alive:
                L.lib().bz_kv_free(self.h)
        except Exception:
            pass

    def seq_len(self):
        return L.lib().bz_kv_seq_len(self.h)

    def reset(self):
        L.check(L.lib().bz_kv_reset(self.h))

    def read(self, layer, kv_head, which, length):
        out = np.empty((length, self.head_dim), dtype=np.float32)
        L.check(L.lib().bz_kv_read(self.h, layer, kv_head, which, length, _ptr(out)))
        return out


class LayeredSsmState:
    """boostr::inference::LayeredSsmState::new(layers, batch, mamba_config, dtype, device) (executor_generate.rs:131-133):
    ssm [B, n_heads, head_dim, d_state] + conv [B, conv_dim, k-1] per layer (docs/architecture.md:52-54)."""

    def __init__(self, model, batch=1, dtype=None):
        h = C.c_void_p()
        L.check(L.lib().bz_ssm_state_create(model.h, batch, model.c.act_dtype if dtype is None else dtype, C.byref(h)))
        self.h, self.dev = h, model.dev

    def __del__(self):
        try:
            if self.h and L.alive:
                L.lib().bz_ssm_state_free(self.h)
        except Exception:
            pass

    def reset(self):
        L.check(L.lib().bz_ssm_state_reset(self.h))

    def read(self, layer, which, n):
        """one layer of the state as float32: which 0 = SSM state (n_heads * head_dim * d_state values), 1 = conv window (conv_dim * (k-1))"""
        out = np.empty(n, dtype=np.float32)
        L.check(L.lib().bz_ssm_state_read(self.h, layer, which, _ptr(out), n))
        return out


class LayeredPagedKvCache:
    """boostr::inference::kv_cache::LayeredPagedKvCache::new (executor_generate.rs:208-210) + a private block list
    (CpuBlockAllocator hands out blocks in order)."""

    def __init__(self, dev, layers, num_blocks, block_size, n_kv_heads, head_dim, dtype):
        h = C.c_void_p()
        L.check(L.lib().bz_paged_kv_create(dev.h, layers, num_blocks, block_size, n_kv_heads, head_dim, dtype, C.byref(h)))
        self.h, self.dev = h, dev
        self.block_size, self.num_blocks = block_size, num_blocks
        self.blocks = []

    def __del__(self):
        try:
            if self.h and L.alive:
                L.lib().bz_paged_kv_free(self.h)
        except Exception:
            pass

    def set_blocks(self, blocks):
        self.blocks = list(blocks)

    def seq_len(self):
        return L.lib().bz_paged_kv_seq_len(self.h)

    def set_seq_len(self, n):
        L.check(L.lib().bz_paged_kv_set_seq_len(self.h, n))

    def compute_slot_mapping(self, start, length):
        """slot = block_id * block_size + offset (batch_decode.rs:81-88)."""
        bs = self.block_size
        return [self.blocks[p // bs] * bs + p % bs for p in range(start, start + length)]

    def block_table_device_format(self):
        return list(self.blocks)


def penalty_window(recent_tokens, repeat_last_n):
    """sampling.rs:169-191: unique ids + counts over the last `repeat_last_n` tokens."""
    w = recent_tokens[-repeat_last_n:] if 0 < repeat_last_n < len(recent_tokens) else recent_tokens
    ids, cnts = [], []
    for t in w:
        if t in ids:
            cnts[ids.index(t)] += 1
        else:
            ids.append(int(t))
            cnts.append(1)
    return np.asarray(ids, dtype=np.int64), np.asarray(cnts, dtype=np.int32)


def logits_to_token(dev, logits, ids, cnts, repeat_penalty=1.0, frequency_penalty=0.0, presence_penalty=0.0, temperature=0.0,
                    top_k=0, top_p=1.0, min_p=0.0, seed=0):
    """SamplingOps::logits_to_token (sampling.rs:445-460) -> I64[1] device tensor."""
    rows, vocab = logits.shape
    n = len(ids)
    tid = dev.tensor(np.asarray(ids, dtype=np.int64)) if n else None
    tcn = dev.tensor(np.asarray(cnts, dtype=np.int32)) if n else None
    out = dev.zeros((1,), L.I64)
    L.check(L.lib().bz_logits_to_token(dev.h, logits.h, rows, vocab, tid.h if n else None, tcn.h if n else None, n, repeat_penalty,
                                       frequency_penalty, presence_penalty, temperature, top_k, top_p, min_p, seed, out.h))
    return out


class BatchDecodeGraph:
    """Executor::capture_batched_graph / replay_batched_graph + BatchedGraphState (cuda_graphs_batched.rs:43-257): one hipGraph per decode step of N
    sequences over a shared paged cache; tokens, positions and slots live on the device between replays."""

    def __init__(self, model, cache, n, max_blocks):
        h = C.c_void_p()
        L.check(L.lib().bz_decode_batch_graph_capture(model.h, cache.h, int(n), int(max_blocks), C.byref(h)))
        self.h, self.model, self.cache, self.n, self.max_blocks = h, model, cache, int(n), int(max_blocks)

    def __del__(self):
        try:
            if self.h and L.alive:
                L.lib().bz_decode_batch_graph_free(self.h)
        except Exception:
            pass

    def _table(self, block_tables):
        bt = np.zeros((self.n, self.max_blocks), dtype=np.int32)
        for i, row in enumerate(block_tables):
            bt[i, :len(row)] = row
        return bt

    def seed(self, tokens, seq_lens, block_tables):
        t = np.ascontiguousarray(tokens, dtype=np.int64)
        sl = np.ascontiguousarray(seq_lens, dtype=np.int32)
        bt = self._table(block_tables)
        L.check(L.lib().bz_decode_batch_graph_seed(self.h, _ptr(t), _ptr(sl), _ptr(bt)))

    def set_block_table(self, block_tables):
        bt = self._table(block_tables)
        L.check(L.lib().bz_decode_batch_graph_set_block_table(self.h, _ptr(bt)))

    def replay(self):
        L.check(L.lib().bz_decode_batch_graph_replay(self.h))

    def read_tokens(self, step):
        out = np.empty(self.n, dtype=np.int64)
        L.check(L.lib().bz_decode_batch_graph_read_tokens(self.h, int(step), _ptr(out)))
        return out

    def read_logits(self):
        t = C.c_void_p()
        L.check(L.lib().bz_decode_batch_graph_logits(self.h, C.byref(t)))
        out = np.empty((self.n, self.model.c.vocab), dtype=np.float32)
        L.check(L.lib().bz_tensor_to_host(t, _ptr(out), out.nbytes))
        return out


class DecodeGraph:
    """inference::decode_graph::DecodeGraph (cuda_graphs.rs:97-189) as a hipGraph."""

    def __init__(self, model, kv, max_blocks=0):
        h = C.c_void_p()
        if isinstance(kv, LayeredSsmState):
            L.check(L.lib().bz_decode_graph_capture_ssm(model.h, kv.h, C.byref(h)))
        elif isinstance(kv, LayeredPagedKvCache):
            L.check(L.lib().bz_decode_graph_capture_paged(model.h, kv.h, max_blocks or kv.num_blocks, C.byref(h)))
        else:
            L.check(L.lib().bz_decode_graph_capture(model.h, kv.h, C.byref(h)))
        self.h, self.model, self.kv = h, model, kv

    def __del__(self):
        try:
            if self.h and L.alive:
                L.lib().bz_decode_graph_free(self.h)
        except Exception:
            pass

    def seed_next_token(self, token, position):
        L.check(L.lib().bz_decode_graph_seed(self.h, int(token), int(position)))

    def set_block_table(self, blocks):
        b = np.asarray(blocks, dtype=np.int32)
        L.check(L.lib().bz_decode_graph_set_block_table(self.h, _ptr(b), len(b)))

    def replay(self):
        L.check(L.lib().bz_decode_graph_replay(self.h))

    def read_token(self, step):
        t = C.c_int64()
        L.check(L.lib().bz_decode_graph_read_token(self.h, step, C.byref(t)))
        return t.value

    def read_logits(self):
        out = np.empty(self.model.c.vocab, dtype=np.float32)
        L.check(L.lib().bz_decode_graph_read_logits(self.h, _ptr(out), len(out)))
        return out


class Executor:
    """engine::Executor<R> restricted to token ids in -> token ids out (tokenizer is out of scope)."""

    def __init__(self, model):
        self.model = model

    def generate(self, prompt_tokens, max_tokens, temperature=0.0, repeat_penalty=1.0, repeat_last_n=64, frequency_penalty=0.0,
                 presence_penalty=0.0, eos_id=-1, use_graph=False, paged=False, block_size=16, seed=0, top_k=0, top_p=1.0, min_p=0.0,
                 dry_multiplier=0.0, dry_base=2, dry_allowed_length=0, typical_p=0.0, dynatemp_range=0.0, dynatemp_exponent=1.0, mirostat_mode=0,
                 mirostat_tau=5.0, mirostat_eta=0.1, logit_bias=None):
        g = L.GenConfig()
        g.max_tokens, g.temperature, g.repeat_penalty, g.repeat_last_n = max_tokens, temperature, repeat_penalty, repeat_last_n
        g.frequency_penalty, g.presence_penalty = frequency_penalty, presence_penalty
        g.top_k, g.top_p, g.min_p, g.seed = top_k, top_p, min_p, seed
        g.eos_id, g.use_graph, g.paged, g.block_size = eos_id, int(use_graph), int(paged), block_size
        g.dry_multiplier, g.dry_base, g.dry_allowed_length, g.typical_p = dry_multiplier, dry_base, dry_allowed_length, typical_p
        g.dynatemp_range, g.dynatemp_exponent = dynatemp_range, dynatemp_exponent
        g.mirostat_mode, g.mirostat_tau, g.mirostat_eta = mirostat_mode, mirostat_tau, mirostat_eta
        if logit_bias:
            b_ids = np.asarray(list(logit_bias.keys()), dtype=np.uint32)
            b_vals = np.asarray(list(logit_bias.values()), dtype=np.float32)
            g.n_logit_bias, g.logit_bias_ids, g.logit_bias_vals = len(b_ids), b_ids.ctypes.data, b_vals.ctypes.data
        p = np.ascontiguousarray(prompt_tokens, dtype=np.int64)
        out = np.zeros(max(max_tokens, 1), dtype=np.int64)
        st = L.GenStats()
        L.check(L.lib().bz_generate(self.model.h, _ptr(p), len(p), C.byref(g), _ptr(out), C.byref(st)))
        self.last_stats = dict(prefill_ms=st.prefill_ms, decode_ms=st.decode_ms, n_generated=st.n_generated, finish_reason=st.finish_reason,
                               ttft_ms=st.ttft_ms, total_ms=st.total_ms, itl_p50_ms=st.itl_p50_ms, itl_p99_ms=st.itl_p99_ms, itl_max_ms=st.itl_max_ms,
                               decode_tok_per_s=st.decode_tok_per_s)
        return out[:st.n_generated]
