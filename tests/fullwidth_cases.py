"""Full-width, few-layer, small-vocab variants of every BASELINE.json config: the exact kernels / grids / template instantiations of the real
models at a size the CPU oracle finishes in seconds.  Shared by tests/test_gpu_fullwidth.py and tests/switch_probe.py (one process per
run-time switch).  Data only; nothing here touches the GPU."""
import numpy as np

from blazr_amd import synth

# name -> (family, preset, overrides, what the case is there to reach)
CASES = {
    # BASELINE configs[2]: H 4096, I 14336, 32q/8kv x 128.  Two layers: layer 0 has Q4_K attn_v / ffn_down, layer 1 Q6_K (the Q4_K_M rule,
    # synth.q4km_uses_q6k), Q6_K `output`: k_gemv_gq_slim<Q4_K|Q6_K, PRO_NORM NJ=2 / PRO_SILU>, k_attn2f (f32 cache, head_dim 128)
    "mistral-7b-q4km-2l": ("llama", "mistral-7b-q4km", dict(n_layers=2, vocab=8192, max_seq_len=512)),
    # GGUF at hidden 2048 (NJ = 1) and 8192 (NJ = 4): the other two NORM instantiations of the slim kernel
    "q4km-h2048": ("llama", "mistral-7b-q4km", dict(hidden=2048, n_heads=16, n_kv_heads=4, inter=2048, n_layers=2, vocab=2048, max_seq_len=256)),
    "q4km-h8192": ("llama", "mistral-7b-q4km", dict(hidden=8192, n_heads=16, n_kv_heads=2, inter=2048, n_layers=2, vocab=2048, max_seq_len=256)),
    "q8_0-h4096": ("llama", "tiny-q8_0", dict(hidden=4096, n_heads=32, n_kv_heads=8, head_dim=128, inter=4096, n_layers=1, vocab=2048, max_seq_len=256)),
    # BASELINE configs[0] decode at width: H 2048, I 8192, 32q/8kv x 64, bf16, tied embeddings, llama3 rope scaling
    "llama3.2-1b-bf16-2l": ("llama", "llama3.2-1b-bf16", dict(n_layers=2, vocab=8192, max_seq_len=512)),
    # AWQ at hidden 2048: fused MLP <.., 2, 4, 8>, slim q/k/v at K = 2048; and hidden 8192: slim q/k/v NJ = 4, unfused MLP
    "awq-h2048": ("llama", "llama3-8b-awq-2l", dict(hidden=2048, n_heads=16, n_kv_heads=4, inter=5632, vocab=4096, max_seq_len=256)),
    "awq-h8192": ("llama", "llama3-8b-awq-2l", dict(hidden=8192, n_heads=16, n_kv_heads=2, inter=2048, n_layers=1, vocab=2048, max_seq_len=256)),
    "gptq-h4096": ("llama", "tiny-gptq", dict(hidden=4096, n_heads=32, n_kv_heads=8, head_dim=128, inter=14336, n_layers=1, vocab=2048, max_seq_len=256)),
    # BASELINE configs[1] itself (already in test_gpu_llama.py::test_full_width_layers; here for the run-time switches)
    "llama3-8b-awq-2l": ("llama", "llama3-8b-awq-2l", {}),
    # BASELINE configs[3]: d_model 2560, 80 heads x 64, d_state 128 (split-K k_gemv_rows on out_proj, 1024-thread SSM step)
    "mamba2-2.7b-2l": ("mamba2", "mamba2-2.7b", dict(n_layers=2, vocab=4096)),
    # BASELINE configs[4]: H 2048, 16 heads, kv_lora 512, nope 128 + rope 64, v 128; one dense layer + one MoE layer (64 routed top-6 + 2 shared)
    "deepseek-v2-lite-2l": ("dsv2", "deepseek-v2-lite", dict(n_layers=2, vocab=4096, max_seq_len=1024)),
}


def make(name):
    fam, preset, over = CASES[name]
    if fam == "llama":
        return fam, synth.make_llama(preset, **over)
    if fam == "mamba2":
        return fam, synth.make_mamba2(preset, **over)
    return fam, synth.make_dsv2(preset, **over)


def make_oracle(fam, model):
    from oracle import orc_py
    return {"llama": orc_py.OrcLlama, "mamba2": orc_py.OrcMamba2, "dsv2": orc_py.OrcDsv2}[fam](model)


class GpuRun:
    """prefill (all logits) + n decode steps fed with given ids, on the product path; state object hidden behind one interface"""

    def __init__(self, device, fam, model):
        from blazr_amd import _lib as L
        from blazr_amd import runtime
        self.fam, self.cfg = fam, model["config"]
        self.lm = runtime.LoadedModel.from_synth(device, model)
        if fam == "mamba2":
            self.state = runtime.LayeredSsmState(self.lm)
        elif fam == "dsv2":
            self.state = self.lm.new_kv_cache(64)
        else:
            dt = {"f16": L.F16, "bf16": L.BF16, "f32": L.F32}[self.cfg["act_dtype"]]
            self.state = runtime.LayeredKvCache(device, self.cfg["n_layers"], 1, self.cfg["n_kv_heads"], 16, self.cfg["max_seq_len"], self.cfg["head_dim"], dt)
        self.pos = 0

    def forward(self, toks, all_logits=False):
        toks = [int(t) for t in toks]
        if self.fam == "mamba2":
            out = self.lm.forward_with_ssm_state(toks, self.state, all_logits=all_logits)
        else:
            out = self.lm.forward_with_kv_cache(toks, self.state, self.pos, all_logits=all_logits)
        self.pos += len(toks)
        return out.to_numpy()


class OrcRun:
    def __init__(self, fam, model, cap=64):
        self.fam = fam
        self.om = make_oracle(fam, model)
        self.state = {"llama": lambda: self.om.new_kv(cap), "mamba2": lambda: self.om.new_state(), "dsv2": lambda: self.om.new_cache(cap)}[fam]()
        self.pos = 0

    def forward(self, toks, all_logits=False):
        toks = [int(t) for t in toks]
        if self.fam == "mamba2":
            out = self.om.forward(toks, self.state, all_logits=all_logits)
        elif self.fam == "dsv2":
            out = self.om.forward(toks, self.state, self.pos, all_logits=all_logits)
        else:
            out = self.om.forward_kv(toks, self.state, self.pos, all_logits=all_logits)
        self.pos += len(toks)
        return np.asarray(out)


def teacher_forced(run, prompt, ids):
    """[prefill all-logits rows..., one row per decode step fed with ids[i]] stacked"""
    rows = [run.forward(prompt, all_logits=True)]
    for t in ids:
        rows.append(run.forward([t]).reshape(1, -1))
    return np.concatenate(rows, axis=0)
