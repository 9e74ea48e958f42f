"""ctypes loader for libblazr_hip.so (the C-ABI in include/blazr_hip.h).

The product path has NO fallback: if the shared library is missing or a compute call is made without a
gfx950 device, it raises.  Nothing here imports oracle/.
"""
import atexit
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BZ_LIB_PATH") or os.path.join(_HERE, "libblazr_hip.so")   # BZ_LIB_PATH: A/B builds of the library (tuning only)
_LIB = None
alive = True   # cleared at interpreter exit: handle destructors must not call into a torn-down HIP runtime


def _at_exit():
    global alive
    alive = False


atexit.register(_at_exit)

OK, E_INVALID, E_NODEVICE, E_HIP, E_UNSUPPORTED, E_NOTFOUND, E_OOM = 0, -1, -2, -3, -4, -5, -6
F32, F16, BF16, I64, I32, U32, U8 = range(7)
ABI_VERSION = 4
FWD_ALL_LOGITS = 1
ROPE_NONE, ROPE_LINEAR, ROPE_LLAMA3, ROPE_YARN = 0, 1, 2, 3
ARCH_LLAMA = 0
ARCH_MAMBA2 = 1
ARCH_DEEPSEEK2 = 2


class BlazrHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libblazr_hip error %d: %s" % (code, msg))
        self.code = code


class ModelConfig(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("arch", C.c_int32), ("hidden", C.c_int32), ("n_layers", C.c_int32),
                ("n_heads", C.c_int32), ("n_kv_heads", C.c_int32), ("head_dim", C.c_int32), ("inter", C.c_int32),
                ("vocab", C.c_int32), ("max_seq_len", C.c_int32), ("rms_eps", C.c_float), ("act_dtype", C.c_int32),
                ("tie_embeddings", C.c_int32), ("rope_theta", C.c_float), ("rope_interleaved", C.c_int32),
                ("rope_scaling", C.c_int32), ("rope_factor", C.c_float), ("rope_low_freq_factor", C.c_float),
                ("rope_high_freq_factor", C.c_float), ("rope_original_max_pos", C.c_int32),
                ("ssm_d_inner", C.c_int32), ("ssm_n_heads", C.c_int32), ("ssm_head_dim", C.c_int32), ("ssm_d_state", C.c_int32),
                ("ssm_n_groups", C.c_int32), ("ssm_conv_kernel", C.c_int32),
                ("mla_kv_lora_rank", C.c_int32), ("mla_q_lora_rank", C.c_int32), ("mla_nope_dim", C.c_int32), ("mla_rope_dim", C.c_int32),
                ("mla_v_dim", C.c_int32), ("moe_n_experts", C.c_int32), ("moe_top_k", C.c_int32), ("moe_n_shared", C.c_int32),
                ("moe_inter", C.c_int32), ("moe_first_dense", C.c_int32), ("moe_norm_topk", C.c_int32), ("moe_routed_scale", C.c_float),
                ("rope_beta_fast", C.c_float), ("rope_beta_slow", C.c_float), ("rope_attn_factor", C.c_float), ("mla_softmax_mscale", C.c_float),
                ("reserved", C.c_int32 * 4)]


class GenConfig(C.Structure):
    _fields_ = [("max_tokens", C.c_int32), ("temperature", C.c_float), ("repeat_penalty", C.c_float),
                ("repeat_last_n", C.c_int32), ("frequency_penalty", C.c_float), ("presence_penalty", C.c_float),
                ("top_k", C.c_int32), ("top_p", C.c_float), ("min_p", C.c_float), ("seed", C.c_uint64),
                ("eos_id", C.c_int64), ("use_graph", C.c_int32), ("paged", C.c_int32), ("block_size", C.c_int32),
                ("dry_multiplier", C.c_float), ("dry_base", C.c_int32), ("dry_allowed_length", C.c_int32), ("typical_p", C.c_float),
                ("dynatemp_range", C.c_float), ("dynatemp_exponent", C.c_float), ("mirostat_mode", C.c_int32), ("mirostat_tau", C.c_float),
                ("mirostat_eta", C.c_float), ("n_logit_bias", C.c_int32), ("logit_bias_ids", C.c_void_p), ("logit_bias_vals", C.c_void_p),
                ("reserved", C.c_int32 * 4)]


class ModelSource(C.Structure):
    _fields_ = [("format", C.c_int32), ("has_config", C.c_int32), ("weights_path", C.c_char * 1024), ("config_path", C.c_char * 1024)]


class DetectedArch(C.Structure):
    _fields_ = [("format", C.c_int32), ("num_layers", C.c_int32), ("tie_word_embeddings", C.c_int32), ("layer_types", C.c_uint8 * 512)]


class QuantInfo(C.Structure):
    _fields_ = [("quant_method", C.c_int32), ("group_size", C.c_int32), ("torch_dtype", C.c_int32)]


class GgufInfo(C.Structure):
    _fields_ = [("architecture", C.c_char * 64), ("n_tensors", C.c_int32), ("version", C.c_int32), ("dominant_ggml_type", C.c_int32),
                ("is_mla", C.c_int32), ("is_moe", C.c_int32), ("is_ssm", C.c_int32), ("file_size_bytes", C.c_uint64)]


class KernelTime(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_int32), ("total_ms", C.c_double), ("algo_bytes", C.c_double)]


class GenStats(C.Structure):
    _fields_ = [("prefill_ms", C.c_double), ("decode_ms", C.c_double), ("n_generated", C.c_int32),
                ("finish_reason", C.c_int32), ("ttft_ms", C.c_double), ("total_ms", C.c_double), ("itl_p50_ms", C.c_double),
                ("itl_p99_ms", C.c_double), ("itl_max_ms", C.c_double), ("decode_tok_per_s", C.c_double)]


# name -> (restype, argtypes); every symbol include/blazr_hip.h declares
P = C.c_void_p
SYMBOLS = {
    "bz_last_error": (C.c_char_p, []),
    "bz_abi_version": (C.c_int, []),
    "bz_device_open": (C.c_int, [C.c_int, C.POINTER(P)]),
    "bz_device_close": (C.c_int, [P]),
    "bz_device_synchronize": (C.c_int, [P]),
    "bz_device_memory_info": (C.c_int, [P, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "bz_device_name": (C.c_int, [P, C.c_char_p, C.c_size_t]),
    "bz_device_stream": (P, [P]),
    "bz_tensor_from_host": (C.c_int, [P, C.c_int, C.POINTER(C.c_int64), C.c_int, P, C.POINTER(P)]),
    "bz_tensor_zeros": (C.c_int, [P, C.c_int, C.POINTER(C.c_int64), C.c_int, C.POINTER(P)]),
    "bz_tensor_free": (C.c_int, [P]),
    "bz_tensor_to_host": (C.c_int, [P, P, C.c_size_t]),
    "bz_tensor_nbytes": (C.c_int, [P, C.POINTER(C.c_size_t)]),
    "bz_tensor_copy_from_host": (C.c_int, [P, P, C.c_size_t]),
    "bz_event_record": (C.c_int, [P, C.POINTER(C.c_uint64)]),
    "bz_event_sync": (C.c_int, [P, C.c_uint64]),
    "bz_tensor_to_host_pipelined": (C.c_int, [P, C.c_uint64, P, C.c_size_t]),
    "bz_model_create": (C.c_int, [P, C.POINTER(ModelConfig), C.POINTER(P)]),
    "bz_model_free": (C.c_int, [P]),
    "bz_model_add_dense": (C.c_int, [P, C.c_char_p, C.c_int, C.POINTER(C.c_int64), C.c_int, P]),
    "bz_model_add_awq": (C.c_int, [P, C.c_char_p, C.c_int64, C.c_int64, P, P, P, C.c_int]),
    "bz_model_add_gptq": (C.c_int, [P, C.c_char_p, C.c_int64, C.c_int64, P, P, P, P, P, C.c_int]),
    "bz_model_add_gguf": (C.c_int, [P, C.c_char_p, C.c_int, C.c_int64, C.c_int64, P]),
    "bz_model_finalize": (C.c_int, [P]),
    "bz_model_get_config": (C.c_int, [P, C.POINTER(ModelConfig)]),
    "bz_model_weight_bytes": (C.c_int, [P, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "bz_kv_create": (C.c_int, [P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(P)]),
    "bz_kv_free": (C.c_int, [P]),
    "bz_kv_reset": (C.c_int, [P]),
    "bz_kv_seq_len": (C.c_int, [P]),
    "bz_kv_read": (C.c_int, [P, C.c_int, C.c_int, C.c_int, C.c_int, P]),
    "bz_paged_kv_create": (C.c_int, [P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(P)]),
    "bz_paged_kv_free": (C.c_int, [P]),
    "bz_paged_kv_set_seq_len": (C.c_int, [P, C.c_int]),
    "bz_paged_kv_seq_len": (C.c_int, [P]),
    "bz_forward_kv": (C.c_int, [P, P, C.c_int, P, C.c_int, P, C.c_uint32]),
    "bz_forward_paged": (C.c_int, [P, P, C.c_int, P, P, P, C.c_int, C.c_int, C.c_int, P, C.c_uint32]),
    "bz_forward_embed": (C.c_int, [P, P, C.c_int, P]),
    "bz_forward_layers_range": (C.c_int, [P, P, P, C.POINTER(C.c_int), C.c_int, P, C.c_int, C.c_int, C.c_int]),
    "bz_forward_head": (C.c_int, [P, P, P, C.c_int, C.c_int, P, C.c_uint32]),
    "bz_logits_to_token": (C.c_int, [P, P, C.c_int64, C.c_int64, P, P, C.c_int, C.c_float, C.c_float, C.c_float,
                                     C.c_float, C.c_int, C.c_float, C.c_float, C.c_uint64, P]),
    "bz_argmax_to_buf": (C.c_int, [P, P, C.c_int64, C.c_int64, P]),
    "bz_decode_graph_capture": (C.c_int, [P, P, C.POINTER(P)]),
    "bz_decode_graph_capture_paged": (C.c_int, [P, P, C.c_int, C.POINTER(P)]),
    "bz_decode_graph_seed": (C.c_int, [P, C.c_int64, C.c_int]),
    "bz_decode_graph_set_block_table": (C.c_int, [P, P, C.c_int]),
    "bz_decode_graph_replay": (C.c_int, [P]),
    "bz_decode_graph_read_token": (C.c_int, [P, C.c_int64, C.POINTER(C.c_int64)]),
    "bz_decode_graph_read_logits": (C.c_int, [P, P, C.c_size_t]),
    "bz_decode_graph_free": (C.c_int, [P]),
    "bz_decode_batch_graph_capture": (C.c_int, [P, P, C.c_int, C.c_int, C.POINTER(P)]),
    "bz_decode_batch_graph_seed": (C.c_int, [P, P, P, P]),
    "bz_decode_batch_graph_set_block_table": (C.c_int, [P, P]),
    "bz_decode_batch_graph_replay": (C.c_int, [P]),
    "bz_decode_batch_graph_read_tokens": (C.c_int, [P, C.c_int64, P]),
    "bz_decode_batch_graph_logits": (C.c_int, [P, C.POINTER(P)]),
    "bz_decode_batch_graph_free": (C.c_int, [P]),
    "bz_generate": (C.c_int, [P, P, C.c_int, C.POINTER(GenConfig), P, C.POINTER(GenStats)]),
    "bz_profile_step": (C.c_int, [P, P, C.c_int64, C.c_int, C.c_int, C.POINTER(KernelTime), C.c_int, C.POINTER(C.c_int)]),
    "bz_profile_step_ssm": (C.c_int, [P, P, C.c_int64, C.c_int, C.POINTER(KernelTime), C.c_int, C.POINTER(C.c_int)]),
    "bz_ssm_state_create": (C.c_int, [P, C.c_int, C.c_int, C.POINTER(P)]),
    "bz_ssm_state_free": (C.c_int, [P]),
    "bz_ssm_state_reset": (C.c_int, [P]),
    "bz_forward_paged_batch": (C.c_int, [P, P, C.c_int, P, P, P, C.c_int, P, P]),
    "bz_forward_ssm": (C.c_int, [P, P, C.c_int, P, P, C.c_uint32]),
    "bz_decode_graph_capture_ssm": (C.c_int, [P, P, C.POINTER(P)]),
    "bz_tune_rows": (C.c_int, [P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]),
    "bz_tune_mlp": (C.c_int, [P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_longlong)]),
    "bz_probe_hbm_read": (C.c_int, [P, C.c_size_t, C.c_int, C.POINTER(C.c_double)]),
    "bz_expf_spec": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "bz_conv1d_step": (C.c_int, [P, C.c_int, P, P, P]),
    "bz_ssm_step": (C.c_int, [P, C.c_int, P, P, P]),
    "bz_ssm_state_read": (C.c_int, [P, C.c_int, C.c_int, C.c_void_p, C.c_size_t]),
    "bz_moe_route": (C.c_int, [P, C.c_int, P, P, P, P]),
    "bz_moe_grouped_gemv": (C.c_int, [P, C.c_int, C.c_int, P, C.c_int, P, P]),
    "bz_tune_gemv": (C.c_int, [P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]),
    "bz_compute_dynamic_temperature": (C.c_float, [P, C.c_int64, C.c_float, C.c_float, C.c_float]),
    "bz_apply_dry_penalty": (C.c_int, [P, C.c_int64, P, C.c_int64, C.c_float, C.c_int, C.c_int]),
    "bz_apply_typical_filter": (C.c_int, [P, C.c_int64, C.c_float]),
    "bz_apply_logit_bias": (C.c_int, [P, C.c_int64, P, P, C.c_int]),
    "bz_compute_logprobs": (C.c_int, [P, C.c_int64, C.c_uint32, C.c_int, C.POINTER(C.c_float), P, P, C.POINTER(C.c_int)]),
    "bz_mirostat_create": (C.c_int, [C.c_float, C.c_float, C.c_uint64, C.POINTER(P)]),
    "bz_mirostat_sample": (C.c_int, [P, P, C.c_int64, C.c_float, C.POINTER(C.c_uint32), C.POINTER(C.c_float)]),
    "bz_mirostat_mu": (C.c_float, [P]),
    "bz_mirostat_free": (C.c_int, [P]),
    "bz_detect_model_source": (C.c_int, [C.c_char_p, C.POINTER(ModelSource)]),
    "bz_detect_architecture_from_names": (C.c_int, [C.POINTER(C.c_char_p), C.c_int, C.POINTER(DetectedArch)]),
    "bz_config_from_hf_json": (C.c_int, [C.c_char_p, C.POINTER(ModelConfig), C.POINTER(QuantInfo)]),
    "bz_config_from_gguf": (C.c_int, [C.c_char_p, C.POINTER(ModelConfig), C.POINTER(GgufInfo)]),
    "bz_safetensors_describe": (C.c_int, [C.c_char_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "bz_load_model": (C.c_int, [P, C.c_char_p, C.POINTER(P), C.POINTER(ModelConfig)]),
    "bz_prefill_matmul": (C.c_int, [P, C.c_char_p, P, C.c_int, P]),
    "bz_quant_matmul": (C.c_int, [P, C.c_char_p, P, C.c_int, P]),
    "bz_dequant": (C.c_int, [P, C.c_char_p, P]),
    "bz_rms_norm": (C.c_int, [P, P, P, P, C.c_int, C.c_int, C.c_float, C.c_int, P, P]),
    "bz_rope": (C.c_int, [P, P, C.c_int, C.c_int, C.c_int]),
    "bz_silu_mul": (C.c_int, [P, P, P, C.c_int64, C.c_int, P]),
    "bz_attn_decode": (C.c_int, [P, P, P, C.c_int, C.c_int, P]),
    "bz_paged_attn_decode": (C.c_int, [P, P, P, C.c_int, P, C.c_int, P]),
    "bz_kv_insert": (C.c_int, [P, P, C.c_int, C.c_int, P, P]),
    "bz_rope_caches": (C.c_int, [P, P, P]),
}


def build(force=False):
    """Compile blazr_amd/csrc for gfx950 with hipcc (cross-compiles without a GPU)."""
    args = ["make", "-s", "-C", os.path.join(_HERE, "csrc")]
    if force:
        subprocess.run(args + ["clean"], check=True)
    subprocess.run(args, check=True)


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise BlazrHipError(E_NODEVICE, "%s not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                                            "(there is no CPU fallback)" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)   # AttributeError here == a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        if L.bz_abi_version() != ABI_VERSION:
            raise BlazrHipError(E_INVALID, "ABI version mismatch")
        _LIB = L
    return _LIB


def check(rc):
    if rc != OK:
        raise BlazrHipError(rc, lib().bz_last_error().decode("utf-8", "replace"))
