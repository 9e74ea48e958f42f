"""ctypes view of oracle/liborc.so -- TEST INFRASTRUCTURE ONLY (see oracle/orc.h).

Only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() may import this module.
The product package (blazr_amd/) never does.  PARITY UNPINNED: the reference holds no golden
vectors for this path; this module restates the algorithm on the CPU.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

F32, F16, BF16 = 0, 1, 2
LIN_DENSE, LIN_AWQ, LIN_GPTQ, LIN_GGUF = 0, 1, 2, 3
GGML_F32, GGML_F16, GGML_Q8_0, GGML_Q4_K, GGML_Q6_K, GGML_BF16 = 0, 1, 8, 12, 14, 30


def build():
    subprocess.run(["make", "-s", "-C", _HERE], check=True)


class Linear(C.Structure):
    _fields_ = [("kind", C.c_int), ("N", C.c_int), ("K", C.c_int), ("group_size", C.c_int), ("w_dtype", C.c_int),
                ("ggml_type", C.c_int), ("w", C.c_void_p), ("scales", C.c_void_p), ("zeros_f", C.c_void_p),
                ("qzeros", C.c_void_p), ("g_idx", C.c_void_p), ("bias", C.c_void_p)]


class RopeCfg(C.Structure):
    _fields_ = [("head_dim", C.c_int), ("max_pos", C.c_int), ("theta", C.c_float), ("scaling_type", C.c_int),
                ("factor", C.c_float), ("low_freq_factor", C.c_float), ("high_freq_factor", C.c_float),
                ("original_max_pos", C.c_int), ("beta_fast", C.c_float), ("beta_slow", C.c_float), ("attn_factor", C.c_float)]


class LlamaCfg(C.Structure):
    _fields_ = [("hidden", C.c_int), ("n_layers", C.c_int), ("n_heads", C.c_int), ("n_kv_heads", C.c_int),
                ("head_dim", C.c_int), ("inter", C.c_int), ("vocab", C.c_int), ("rms_eps", C.c_float),
                ("act_dtype", C.c_int), ("rope_interleaved", C.c_int), ("max_seq_len", C.c_int), ("rope", RopeCfg)]


class LlamaLayer(C.Structure):
    _fields_ = [("attn_norm", C.c_void_p), ("ffn_norm", C.c_void_p), ("q", Linear), ("k", Linear), ("v", Linear),
                ("o", Linear), ("gate", Linear), ("up", Linear), ("down", Linear)]


class Llama(C.Structure):
    _fields_ = [("cfg", LlamaCfg), ("embed", C.c_void_p), ("embed_dtype", C.c_int), ("final_norm", C.c_void_p),
                ("lm_head", Linear), ("layers", C.POINTER(LlamaLayer)), ("cos_t", C.c_void_p), ("sin_t", C.c_void_p)]


class Mamba2Cfg(C.Structure):
    _fields_ = [("hidden", C.c_int), ("n_layers", C.c_int), ("vocab", C.c_int), ("d_inner", C.c_int), ("n_heads", C.c_int),
                ("head_dim", C.c_int), ("d_state", C.c_int), ("n_groups", C.c_int), ("conv_kernel", C.c_int), ("rms_eps", C.c_float),
                ("act_dtype", C.c_int)]


class Mamba2Layer(C.Structure):
    _fields_ = [("norm", C.c_void_p), ("in_proj", Linear), ("conv_w", C.c_void_p), ("conv_b", C.c_void_p), ("dt_bias", C.c_void_p),
                ("A_log", C.c_void_p), ("D", C.c_void_p), ("gnorm", C.c_void_p), ("out_proj", Linear)]


class Mamba2(C.Structure):
    _fields_ = [("cfg", Mamba2Cfg), ("embed", C.c_void_p), ("embed_dtype", C.c_int), ("final_norm", C.c_void_p), ("lm_head", Linear),
                ("layers", C.POINTER(Mamba2Layer))]


class Dsv2Cfg(C.Structure):
    _fields_ = [("hidden", C.c_int), ("n_layers", C.c_int), ("n_heads", C.c_int), ("vocab", C.c_int), ("max_seq_len", C.c_int),
                ("kv_lora_rank", C.c_int), ("q_lora_rank", C.c_int), ("nope_dim", C.c_int), ("rope_dim", C.c_int), ("v_dim", C.c_int),
                ("inter", C.c_int), ("n_experts", C.c_int), ("top_k", C.c_int), ("n_shared", C.c_int), ("moe_inter", C.c_int),
                ("first_dense", C.c_int), ("routed_scale", C.c_float), ("norm_topk", C.c_int), ("rms_eps", C.c_float),
                ("act_dtype", C.c_int), ("rope", RopeCfg), ("softmax_mscale", C.c_float)]


class Dsv2Layer(C.Structure):
    _fields_ = [("attn_norm", C.c_void_p), ("ffn_norm", C.c_void_p), ("kv_norm", C.c_void_p), ("q_norm", C.c_void_p),
                ("q_proj", Linear), ("q_b", Linear), ("kv_a", Linear), ("kv_b", Linear), ("o", Linear), ("is_moe", C.c_int),
                ("gate", Linear), ("up", Linear), ("down", Linear), ("router", Linear),
                ("e_gate", C.POINTER(Linear)), ("e_up", C.POINTER(Linear)), ("e_down", C.POINTER(Linear)),
                ("s_gate", Linear), ("s_up", Linear), ("s_down", Linear), ("kv_b_f32", C.c_void_p)]


class Dsv2(C.Structure):
    _fields_ = [("cfg", Dsv2Cfg), ("embed", C.c_void_p), ("embed_dtype", C.c_int), ("final_norm", C.c_void_p), ("lm_head", Linear),
                ("layers", C.POINTER(Dsv2Layer)), ("cos_t", C.c_void_p), ("sin_t", C.c_void_p)]


class MlaCache(C.Structure):
    _fields_ = [("n_layers", C.c_int), ("width", C.c_int), ("capacity", C.c_int), ("seq_len", C.c_int), ("lat", C.c_void_p)]


class SsmState(C.Structure):
    _fields_ = [("ssm", C.c_void_p), ("conv", C.c_void_p)]


class Kv(C.Structure):
    _fields_ = [("n_layers", C.c_int), ("n_kv_heads", C.c_int), ("head_dim", C.c_int), ("capacity", C.c_int),
                ("seq_len", C.c_int), ("k", C.c_void_p), ("v", C.c_void_p)]


class PagedKv(C.Structure):
    _fields_ = [("n_layers", C.c_int), ("n_kv_heads", C.c_int), ("head_dim", C.c_int), ("num_blocks", C.c_int),
                ("block_size", C.c_int), ("seq_len", C.c_int), ("k", C.c_void_p), ("v", C.c_void_p)]


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liborc.so")
        if not os.path.exists(path):
            build()
        # the GPU box exposes many more cores than its CPU share (16 for one GPU): cap the OpenMP team and never spin
        ncpu = max(1, min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)))
        os.environ.setdefault("OMP_NUM_THREADS", str(ncpu))
        os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
        L = C.CDLL(path)
        L.orc_set_num_threads.argtypes = [C.c_int]
        L.orc_set_num_threads(int(os.environ["OMP_NUM_THREADS"]))
        L.orc_f16_to_f32.restype = C.c_float
        L.orc_f16_to_f32.argtypes = [C.c_uint16]
        L.orc_f32_to_f16.restype = C.c_uint16
        L.orc_f32_to_f16.argtypes = [C.c_float]
        L.orc_f32_to_bf16.restype = C.c_uint16
        L.orc_f32_to_bf16.argtypes = [C.c_float]
        L.orc_round.restype = C.c_float
        L.orc_round.argtypes = [C.c_float, C.c_int]
        L.orc_ggml_row_bytes.restype = C.c_size_t
        L.orc_ggml_row_bytes.argtypes = [C.c_int, C.c_size_t]
        L.orc_ggml_dequant.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
        L.orc_linear_forward.argtypes = [C.POINTER(Linear), C.c_void_p, C.c_int, C.c_void_p]
        L.orc_linear_dequant.argtypes = [C.POINTER(Linear), C.c_void_p]
        L.orc_awq_unpack_zeros.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.orc_rms_norm.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_void_p]
        L.orc_rope_tables.argtypes = [C.POINTER(RopeCfg), C.c_void_p, C.c_void_p]
        L.orc_rope_apply.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_attn_decode.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int,
                                      C.c_float, C.c_void_p]
        L.orc_silu.restype = C.c_float
        L.orc_silu.argtypes = [C.c_float]
        L.orc_expf.restype = C.c_float
        L.orc_expf.argtypes = [C.c_float]
        L.orc_argmax.restype = C.c_int64
        L.orc_argmax.argtypes = [C.c_void_p, C.c_int64]
        L.orc_logits_to_token.restype = C.c_int64
        L.orc_logits_to_token.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_float,
                                          C.c_float, C.c_float, C.c_int, C.c_float, C.c_float, C.c_uint64]
        L.orc_penalty_window.restype = C.c_int
        L.orc_penalty_window.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_llama_new.restype = C.POINTER(Llama)
        L.orc_llama_new.argtypes = [C.POINTER(LlamaCfg)]
        L.orc_llama_free.argtypes = [C.POINTER(Llama)]
        L.orc_kv_new.restype = C.POINTER(Kv)
        L.orc_kv_new.argtypes = [C.c_int] * 4
        L.orc_kv_free.argtypes = [C.POINTER(Kv)]
        L.orc_paged_kv_new.restype = C.POINTER(PagedKv)
        L.orc_paged_kv_new.argtypes = [C.c_int] * 5
        L.orc_paged_kv_free.argtypes = [C.POINTER(PagedKv)]
        L.orc_llama_forward_kv.restype = C.c_int
        L.orc_llama_forward_kv.argtypes = [C.POINTER(Llama), C.c_void_p, C.c_int, C.POINTER(Kv), C.c_int, C.c_void_p,
                                           C.c_int]
        L.orc_llama_forward_paged.restype = C.c_int
        L.orc_llama_forward_paged.argtypes = [C.POINTER(Llama), C.c_void_p, C.c_int, C.POINTER(PagedKv), C.c_void_p,
                                              C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.orc_llama_embed.argtypes = [C.POINTER(Llama), C.c_void_p, C.c_int, C.c_void_p]
        L.orc_llama_layers_range.restype = C.c_int
        L.orc_llama_layers_range.argtypes = [C.POINTER(Llama), C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.c_int,
                                             C.POINTER(Kv), C.c_int, C.c_int, C.c_int]
        L.orc_llama_head.argtypes = [C.POINTER(Llama), C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.orc_llama_generate.restype = C.c_int
        L.orc_llama_generate.argtypes = [C.POINTER(Llama), C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int64,
                                         C.c_void_p, C.c_void_p]
        L.orc_mamba2_new.restype = C.POINTER(Mamba2)
        L.orc_mamba2_new.argtypes = [C.POINTER(Mamba2Cfg)]
        L.orc_mamba2_free.argtypes = [C.POINTER(Mamba2)]
        L.orc_ssm_state_new.restype = C.POINTER(SsmState)
        L.orc_ssm_state_new.argtypes = [C.POINTER(Mamba2Cfg)]
        L.orc_ssm_state_free.argtypes = [C.POINTER(SsmState)]
        L.orc_mamba2_forward.restype = C.c_int
        L.orc_mamba2_forward.argtypes = [C.POINTER(Mamba2), C.c_void_p, C.c_int, C.POINTER(SsmState), C.c_void_p, C.c_int]
        L.orc_mamba2_conv1d_step.argtypes = [C.POINTER(Mamba2Cfg), C.POINTER(Mamba2Layer), C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_mamba2_ssm_step.argtypes = [C.POINTER(Mamba2Cfg), C.POINTER(Mamba2Layer), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_mamba2_generate.restype = C.c_int
        L.orc_mamba2_generate.argtypes = [C.POINTER(Mamba2), C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_void_p]
        L.orc_dsv2_new.restype = C.POINTER(Dsv2)
        L.orc_dsv2_new.argtypes = [C.POINTER(Dsv2Cfg)]
        L.orc_dsv2_prepare.argtypes = [C.POINTER(Dsv2)]
        L.orc_dsv2_free.argtypes = [C.POINTER(Dsv2)]
        L.orc_mla_cache_new.restype = C.POINTER(MlaCache)
        L.orc_mla_cache_new.argtypes = [C.POINTER(Dsv2Cfg), C.c_int]
        L.orc_mla_cache_free.argtypes = [C.POINTER(MlaCache)]
        L.orc_moe_route.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_dsv2_forward.restype = C.c_int
        L.orc_dsv2_forward.argtypes = [C.POINTER(Dsv2), C.c_void_p, C.c_int, C.POINTER(MlaCache), C.c_int, C.c_void_p, C.c_int]
        L.orc_dsv2_generate.restype = C.c_int
        L.orc_dsv2_generate.argtypes = [C.POINTER(Dsv2), C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_void_p]
        L.orc_num_threads.restype = C.c_int
        _LIB = L
    return _LIB


def set_threads(n):
    """OpenMP team size of the oracle's GEMVs (bench.py --cpu-threads); the default team is min(16, CPU share) -- see lib()."""
    lib().orc_set_num_threads(int(n))


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


_DT = {"f32": F32, "f16": F16, "bf16": BF16}


def round_act(x, act):
    """Round a float32 array to values representable in `act` ('f32' | 'f16' | 'bf16')."""
    x = np.asarray(x, dtype=np.float32)
    if act == "f16":
        return x.astype(np.float16).astype(np.float32)
    if act == "bf16":
        u = x.view(np.uint32).astype(np.uint64)
        r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint32) << 16
        return r.astype(np.uint32).view(np.float32).reshape(x.shape)
    return x.copy()


class OrcLinear:
    """Keeps numpy buffers alive next to the C struct. `spec` is a blazr_amd.synth-style dict."""

    def __init__(self, spec):
        self.keep = []
        s = Linear()
        kind = spec["kind"]
        s.N, s.K = int(spec["N"]), int(spec["K"])
        s.group_size = int(spec.get("group_size", 0))
        if kind == "dense":
            s.kind = LIN_DENSE
            w = np.ascontiguousarray(spec["weight"])
            s.w_dtype = {np.dtype(np.float32): F32, np.dtype(np.float16): F16, np.dtype(np.uint16): BF16}[w.dtype]
            self.keep.append(w)
            s.w = _p(w)
        elif kind == "awq":
            s.kind = LIN_AWQ
            qw = np.ascontiguousarray(spec["qweight"], dtype=np.uint32)
            sc = np.ascontiguousarray(spec["scales"]).astype(np.float32)     # awq.rs:202-206
            G = sc.shape[0]
            zf = np.empty((G, s.N), dtype=np.float32)
            qz = np.ascontiguousarray(spec["qzeros"], dtype=np.uint32)
            lib().orc_awq_unpack_zeros(_p(qz), G, s.N, _p(zf))               # awq.rs:208-213
            self.keep += [qw, sc, zf]
            s.w, s.scales, s.zeros_f = _p(qw), _p(sc), _p(zf)
        elif kind == "gptq":
            s.kind = LIN_GPTQ
            qw = np.ascontiguousarray(spec["qweight"], dtype=np.uint32)
            sc = np.ascontiguousarray(spec["scales"]).astype(np.float32)
            qz = np.ascontiguousarray(spec["qzeros"], dtype=np.uint32)
            self.keep += [qw, sc, qz]
            s.w, s.scales, s.qzeros = _p(qw), _p(sc), _p(qz)
            if spec.get("g_idx") is not None:
                gi = np.ascontiguousarray(spec["g_idx"], dtype=np.int32)
                self.keep.append(gi)
                s.g_idx = _p(gi)
        elif kind == "gguf":
            s.kind = LIN_GGUF
            s.ggml_type = int(spec["ggml_type"])
            w = np.ascontiguousarray(spec["blocks"], dtype=np.uint8)
            self.keep.append(w)
            s.w = _p(w)
        else:
            raise ValueError(kind)
        if spec.get("bias") is not None:
            b = np.ascontiguousarray(spec["bias"]).astype(np.float32)
            self.keep.append(b)
            s.bias = _p(b)
        self.c = s

    def forward(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1, self.c.K)
        y = np.empty((x.shape[0], self.c.N), dtype=np.float32)
        lib().orc_linear_forward(C.byref(self.c), _p(x), x.shape[0], _p(y))
        return y

    def dequant(self):
        out = np.empty((self.c.N, self.c.K), dtype=np.float32)
        lib().orc_linear_dequant(C.byref(self.c), _p(out))
        return out


def ggml_dequant(ggml_type, blocks, n):
    blocks = np.ascontiguousarray(blocks, dtype=np.uint8)
    out = np.empty(n, dtype=np.float32)
    lib().orc_ggml_dequant(ggml_type, _p(blocks), n, _p(out))
    return out


class OrcLlama:
    """Oracle Llama-family model built from a blazr_amd.synth model dict."""

    def __init__(self, model):
        cfg = model["config"]
        self.model = model
        c = LlamaCfg()
        c.hidden, c.n_layers, c.n_heads, c.n_kv_heads = cfg["hidden"], cfg["n_layers"], cfg["n_heads"], cfg["n_kv_heads"]
        c.head_dim, c.inter, c.vocab = cfg["head_dim"], cfg["inter"], cfg["vocab"]
        c.rms_eps = cfg["rms_eps"]
        c.act_dtype = _DT[cfg["act_dtype"]]
        c.rope_interleaved = int(cfg.get("rope_interleaved", 0))
        c.max_seq_len = cfg["max_seq_len"]
        _rope_cfg(cfg, c.rope)
        self.cfg = c
        self.h = lib().orc_llama_new(C.byref(c))
        self.keep = []
        m = self.h.contents
        emb = np.ascontiguousarray(model["embed"])
        m.embed = _p(emb)
        m.embed_dtype = {np.dtype(np.float32): F32, np.dtype(np.float16): F16, np.dtype(np.uint16): BF16}[emb.dtype]
        fn = np.ascontiguousarray(model["final_norm"], dtype=np.float32)
        m.final_norm = _p(fn)
        self.keep += [emb, fn]
        lm = OrcLinear(model["lm_head"])
        self.keep.append(lm)
        m.lm_head = lm.c
        for i, lay in enumerate(model["layers"]):
            L = m.layers[i]
            an = np.ascontiguousarray(lay["attn_norm"], dtype=np.float32)
            fnn = np.ascontiguousarray(lay["ffn_norm"], dtype=np.float32)
            L.attn_norm, L.ffn_norm = _p(an), _p(fnn)
            self.keep += [an, fnn]
            for name in ("q", "k", "v", "o", "gate", "up", "down"):
                ol = OrcLinear(lay[name])
                self.keep.append(ol)
                setattr(L, name, ol.c)

    def __del__(self):
        try:
            lib().orc_llama_free(self.h)
        except Exception:
            pass

    def new_kv(self, capacity):
        c = self.cfg
        return lib().orc_kv_new(c.n_layers, c.n_kv_heads, c.head_dim, capacity)

    def new_paged_kv(self, num_blocks, block_size):
        c = self.cfg
        return lib().orc_paged_kv_new(c.n_layers, num_blocks, block_size, c.n_kv_heads, c.head_dim)

    def forward_kv(self, tokens, kv, position, all_logits=False):
        t = np.ascontiguousarray(tokens, dtype=np.int64)
        out = np.empty((len(t) if all_logits else 1, self.cfg.vocab), dtype=np.float32)
        rc = lib().orc_llama_forward_kv(self.h, _p(t), len(t), kv, position, _p(out), int(all_logits))
        if rc != 0:
            raise RuntimeError("orc_llama_forward_kv rc=%d" % rc)
        return out

    def forward_paged(self, tokens, pkv, slot_mapping, block_table, seq_len_k, start_pos, all_logits=False):
        t = np.ascontiguousarray(tokens, dtype=np.int64)
        sm = np.ascontiguousarray(slot_mapping, dtype=np.int32)
        bt = np.ascontiguousarray(block_table, dtype=np.int32)
        out = np.empty((len(t) if all_logits else 1, self.cfg.vocab), dtype=np.float32)
        rc = lib().orc_llama_forward_paged(self.h, _p(t), len(t), pkv, _p(sm), _p(bt), len(bt), seq_len_k, start_pos,
                                           _p(out), int(all_logits))
        if rc != 0:
            raise RuntimeError("orc_llama_forward_paged rc=%d" % rc)
        return out

    def embed(self, tokens):
        t = np.ascontiguousarray(tokens, dtype=np.int64)
        h = np.empty((len(t), self.cfg.hidden), dtype=np.float32)
        lib().orc_llama_embed(self.h, _p(t), len(t), _p(h))
        return h

    def layers_range(self, hidden, prev_mlp, kv, start, end, position):
        hidden = np.ascontiguousarray(hidden, dtype=np.float32).copy()
        S = hidden.shape[0]
        has = C.c_int(0 if prev_mlp is None else 1)
        pm = np.zeros_like(hidden) if prev_mlp is None else np.ascontiguousarray(prev_mlp, dtype=np.float32).copy()
        rc = lib().orc_llama_layers_range(self.h, _p(hidden), _p(pm), C.byref(has), S, kv, start, end, position)
        if rc != 0:
            raise RuntimeError("orc_llama_layers_range rc=%d" % rc)
        return hidden, (pm if has.value else None)

    def head(self, hidden, prev_mlp, all_logits=False):
        hidden = np.ascontiguousarray(hidden, dtype=np.float32)
        S = hidden.shape[0]
        out = np.empty((S if all_logits else 1, self.cfg.vocab), dtype=np.float32)
        pm = None if prev_mlp is None else np.ascontiguousarray(prev_mlp, dtype=np.float32)
        lib().orc_llama_head(self.h, _p(hidden), _p(pm), int(pm is not None), S, _p(out), int(all_logits))
        return out

    def generate(self, prompt, max_tokens, repeat_penalty=1.0, repeat_last_n=64, eos_id=-1, trace=False):
        p = np.ascontiguousarray(prompt, dtype=np.int64)
        out = np.zeros(max_tokens, dtype=np.int64)
        tr = np.zeros((max_tokens, self.cfg.vocab), dtype=np.float32) if trace else None
        n = lib().orc_llama_generate(self.h, _p(p), len(p), max_tokens, repeat_penalty, repeat_last_n, eos_id, _p(out),
                                     _p(tr))
        if n < 0:
            raise RuntimeError("orc_llama_generate failed")
        return (out[:n], tr[:n]) if trace else out[:n]


class OrcMamba2:
    """Oracle Mamba2 model built from a blazr_amd.synth.make_mamba2 dict."""

    def __init__(self, model):
        cfg = model["config"]
        c = Mamba2Cfg()
        for k in ("hidden", "n_layers", "vocab", "d_inner", "n_heads", "head_dim", "d_state", "n_groups", "conv_kernel"):
            setattr(c, k, cfg[k])
        c.rms_eps = cfg["rms_eps"]
        c.act_dtype = _DT[cfg["act_dtype"]]
        self.cfg = c
        self.h = lib().orc_mamba2_new(C.byref(c))
        self.keep = []
        m = self.h.contents
        emb = np.ascontiguousarray(model["embed"])
        m.embed = _p(emb)
        m.embed_dtype = {np.dtype(np.float32): F32, np.dtype(np.float16): F16, np.dtype(np.uint16): BF16}[emb.dtype]
        fn = np.ascontiguousarray(model["final_norm"], dtype=np.float32)
        m.final_norm = _p(fn)
        lm = OrcLinear(model["lm_head"])
        m.lm_head = lm.c
        self.keep += [emb, fn, lm]
        for i, lay in enumerate(model["layers"]):
            Lr = m.layers[i]
            for name in ("norm", "conv_w", "conv_b", "dt_bias", "A_log", "D", "gnorm"):
                a = np.ascontiguousarray(lay[name], dtype=np.float32)
                self.keep.append(a)
                setattr(Lr, name, _p(a))
            for name in ("in_proj", "out_proj"):
                ol = OrcLinear(lay[name])
                self.keep.append(ol)
                setattr(Lr, name, ol.c)

    def __del__(self):
        try:
            lib().orc_mamba2_free(self.h)
        except Exception:
            pass

    def new_state(self):
        return lib().orc_ssm_state_new(C.byref(self.cfg))

    def forward(self, tokens, state, all_logits=False):
        t = np.ascontiguousarray(tokens, dtype=np.int64)
        out = np.empty((len(t) if all_logits else 1, self.cfg.vocab), dtype=np.float32)
        lib().orc_mamba2_forward(self.h, _p(t), len(t), state, _p(out), int(all_logits))
        return out

    def conv1d_step(self, layer, zx, conv_state):
        """conv_state: float32 [conv_dim, k-1] of ONE layer, updated in place; returns xbc [conv_dim]"""
        c = self.cfg
        xbc = np.empty(c.d_inner + 2 * c.n_groups * c.d_state, dtype=np.float32)
        lib().orc_mamba2_conv1d_step(C.byref(c), C.byref(self.h.contents.layers[layer]), _p(np.ascontiguousarray(zx, dtype=np.float32)), _p(conv_state), _p(xbc))
        return xbc

    def ssm_step(self, layer, zx, xbc, ssm):
        """ssm: float32 [n_heads, head_dim, d_state] of ONE layer, updated in place; returns the gated y [d_inner]"""
        c = self.cfg
        y = np.empty(c.d_inner, dtype=np.float32)
        lib().orc_mamba2_ssm_step(C.byref(c), C.byref(self.h.contents.layers[layer]), _p(np.ascontiguousarray(zx, dtype=np.float32)), _p(np.ascontiguousarray(xbc, dtype=np.float32)),
                                  _p(ssm), _p(y))
        return y

    def generate(self, prompt, max_tokens, eos_id=-1, trace=False):
        p = np.ascontiguousarray(prompt, dtype=np.int64)
        out = np.zeros(max_tokens, dtype=np.int64)
        tr = np.zeros((max_tokens, self.cfg.vocab), dtype=np.float32) if trace else None
        n = lib().orc_mamba2_generate(self.h, _p(p), len(p), max_tokens, eos_id, _p(out), _p(tr))
        return (out[:n], tr[:n]) if trace else out[:n]


def _yarn_mscale(factor, mscale=1.0):
    return 1.0 if factor <= 1.0 else 0.1 * mscale * float(np.log(factor)) + 1.0


def _rope_cfg(cfg, r):
    """rope_scaling dict (HF names) -> orc_rope_cfg; returns the MLA softmax mscale (DeepSeek-V2 YaRN: mscale_all_dim), 0 = none"""
    rs = cfg.get("rope_scaling") or {}
    r.theta = cfg["rope_theta"]
    r.scaling_type = {"none": 0, "linear": 1, "llama3": 2, "yarn": 3}[rs.get("type", "none")]
    r.factor = rs.get("factor", 1.0)
    r.low_freq_factor = rs.get("low_freq_factor", 1.0)
    r.high_freq_factor = rs.get("high_freq_factor", 4.0)
    r.original_max_pos = rs.get("original_max_position_embeddings", 8192)
    r.beta_fast, r.beta_slow, r.attn_factor = rs.get("beta_fast", 0.0), rs.get("beta_slow", 0.0), 0.0
    sm = 0.0
    if rs.get("type") == "yarn":
        if "attention_factor" in rs:
            r.attn_factor = rs["attention_factor"]
        elif "mscale" in rs and "mscale_all_dim" in rs:
            r.attn_factor = _yarn_mscale(r.factor, rs["mscale"]) / _yarn_mscale(r.factor, rs["mscale_all_dim"])
        if cfg.get("arch") == "deepseek2" and rs.get("mscale_all_dim"):
            sm = _yarn_mscale(r.factor, rs["mscale_all_dim"])
    return sm


class OrcDsv2:
    """Oracle DeepSeek-V2 (MLA + MoE) model built from a blazr_amd.synth.make_dsv2 dict."""

    def __init__(self, model):
        cfg = model["config"]
        c = Dsv2Cfg()
        for k in ("hidden", "n_layers", "n_heads", "vocab", "max_seq_len", "kv_lora_rank", "q_lora_rank", "nope_dim", "rope_dim", "v_dim",
                  "inter", "n_experts", "top_k", "n_shared", "moe_inter", "first_dense"):
            setattr(c, k, int(cfg[k]))
        c.routed_scale, c.norm_topk = float(cfg["routed_scale"]), int(bool(cfg["norm_topk"]))
        c.rms_eps, c.act_dtype = cfg["rms_eps"], _DT[cfg["act_dtype"]]
        c.softmax_mscale = _rope_cfg(cfg, c.rope)
        self.cfg = c
        self.h = lib().orc_dsv2_new(C.byref(c))
        self.keep = []
        m = self.h.contents
        emb = np.ascontiguousarray(model["embed"])
        m.embed = _p(emb)
        m.embed_dtype = {np.dtype(np.float32): F32, np.dtype(np.float16): F16, np.dtype(np.uint16): BF16}[emb.dtype]
        fn = np.ascontiguousarray(model["final_norm"], dtype=np.float32)
        m.final_norm = _p(fn)
        lm = OrcLinear(model["lm_head"])
        m.lm_head = lm.c
        self.keep += [emb, fn, lm]

        def lin(spec):
            ol = OrcLinear(spec)
            self.keep.append(ol)
            return ol.c

        for i, lay in enumerate(model["layers"]):
            Lr = m.layers[i]
            for name in ("attn_norm", "ffn_norm", "kv_norm"):
                a = np.ascontiguousarray(lay[name], dtype=np.float32)
                self.keep.append(a)
                setattr(Lr, name, _p(a))
            for name in ("q_proj", "kv_a", "kv_b", "o"):
                setattr(Lr, name, lin(lay[name]))
            if "q_b" in lay:                      # q_lora_rank > 0: q_proj holds q_a_proj
                a = np.ascontiguousarray(lay["q_norm"], dtype=np.float32)
                self.keep.append(a)
                Lr.q_norm = _p(a)
                Lr.q_b = lin(lay["q_b"])
            Lr.is_moe = int(lay["is_moe"])
            if not lay["is_moe"]:
                for name in ("gate", "up", "down"):
                    setattr(Lr, name, lin(lay[name]))
            else:
                Lr.router = lin(lay["router"])
                E = len(lay["experts"])
                for name in ("gate", "up", "down"):
                    arr = (Linear * E)(*[lin(ex[name]) for ex in lay["experts"]])
                    self.keep.append(arr)
                    setattr(Lr, "e_" + name, C.cast(arr, C.POINTER(Linear)))
                if "shared" in lay:
                    for name in ("gate", "up", "down"):
                        setattr(Lr, "s_" + name, lin(lay["shared"][name]))
        lib().orc_dsv2_prepare(self.h)

    def __del__(self):
        try:
            lib().orc_dsv2_free(self.h)
        except Exception:
            pass

    def new_cache(self, capacity):
        return lib().orc_mla_cache_new(C.byref(self.cfg), capacity)

    def forward(self, tokens, cache, position, all_logits=False):
        t = np.ascontiguousarray(tokens, dtype=np.int64)
        out = np.empty((len(t) if all_logits else 1, self.cfg.vocab), dtype=np.float32)
        rc = lib().orc_dsv2_forward(self.h, _p(t), len(t), cache, position, _p(out), int(all_logits))
        assert rc == 0, "orc_dsv2_forward failed (cache capacity / max_seq_len)"
        return out

    def generate(self, prompt, max_tokens, eos_id=-1, trace=False):
        p = np.ascontiguousarray(prompt, dtype=np.int64)
        out = np.zeros(max(max_tokens, 1), dtype=np.int64)
        tr = np.zeros((max(max_tokens, 1), self.cfg.vocab), dtype=np.float32) if trace else None
        n = lib().orc_dsv2_generate(self.h, _p(p), len(p), max_tokens, eos_id, _p(out), _p(tr))
        return (out[:n], tr[:n]) if trace else out[:n]
