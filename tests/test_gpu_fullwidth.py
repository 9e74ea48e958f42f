"""Every BASELINE.json config at its REAL layer widths (2 layers, small vocab) against the CPU oracle, through the C ABI.

VERDICT r01 'configs_untested': Mistral-7B Q4_K_M (the slim Q4_K / Q6_K GEMVs and the f32-cache attention that carry its number),
Mamba2-2.7B (d_model 2560, 80 x 64 heads, split-K row GEMV), DeepSeek-V2-Lite (H 2048, 64 experts top-6 + 2 shared, kv_lora 512), the decode
kernels of Llama-3.2-1B bf16, and the AWQ kernels' other instantiations (hidden 2048 / 8192).  Logits are held to the north-star bar itself
(factor 1.0: relative L2 <= 1e-3 for f16 / f32 activations, 2^-7 for bf16), greedy ids bit-exact on the fair prefix.

The second half runs the same comparisons under each run-time switch (BZ_NO_*: the fallback kernels INTEGRATION.md lists), one fresh
process per switch because the library reads them once.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

from blazr_amd import runtime, synth
from fullwidth_cases import CASES, GpuRun, OrcRun, make, make_oracle, teacher_forced
from test_gpu_llama import _check_logits, _fair_prefix

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def _factor(act):
    """f16 / f32 activations: the north-star bar itself (1e-3 relative L2).  bf16 (8 significant bits): 2^-7 IS the distance between two correct
    bf16 pipelines -- the oracle itself sits 8.6e-3 from an unrounded float64 evaluation (tests/test_gpu_parity_truth.py prints both distances,
    and bounds the HIP path by 1.25x the oracle's own) -- so full-width bf16 rows are held to 1.25 x 2^-7 here and to the truth-relative bound there"""
    return 1.25 if act == "bf16" else 1.0


@pytest.fixture(scope="module", params=[c for c in CASES if c != "llama3-8b-awq-2l"])
def case(request, device):
    fam, model = make(request.param)
    return request.param, fam, model


def test_logits_prefill_and_decode_at_the_bar(case, device):
    """6-token prompt (token-by-token: the decode kernels), 8 teacher-forced decode steps, then a 20-token second chunk (the batched prefill
    path where the family has one) -- every row of logits at factor 1.0"""
    name, fam, model = case
    cfg = model["config"]
    g, o = GpuRun(device, fam, model), OrcRun(fam, model, cap=64)
    p = synth.prompt_tokens(6, cfg["vocab"], seed=2)
    want = o.forward(p, all_logits=True)
    got = g.forward(p, all_logits=True)
    f = _factor(cfg["act_dtype"])
    _check_logits(got, want, cfg["act_dtype"], factor=f)
    tok = int(want[-1].argmax())
    for _ in range(8):
        lo, lg = o.forward([tok]), g.forward([tok])
        _check_logits(lg.reshape(-1), lo.reshape(-1), cfg["act_dtype"], factor=f)
        tok = int(lo.reshape(-1).argmax())
    p2 = synth.prompt_tokens(20, cfg["vocab"], seed=9)
    _check_logits(g.forward(p2, all_logits=True), o.forward(p2, all_logits=True), cfg["act_dtype"], factor=f)
    lo, lg = o.forward([tok]), g.forward([tok])
    _check_logits(lg.reshape(-1), lo.reshape(-1), cfg["act_dtype"], factor=f)


@pytest.mark.parametrize("mode", ["eager", "graph"])
def test_greedy_ids(case, device, mode):
    name, fam, model = case
    cfg = model["config"]
    om = make_oracle(fam, model)
    n = 0
    for seed in range(3, 40):   # first prompt whose oracle run has no near-tie in its first 8 steps (deterministic)
        p = synth.prompt_tokens(10, cfg["vocab"], seed=seed)
        want, trace = om.generate(p, 16, trace=True)
        n = _fair_prefix(trace)
        if n >= 8:
            break
    assert n >= 8, "no prompt seed gives a fair fixture"
    lm = runtime.LoadedModel.from_synth(device, model)
    got = runtime.Executor(lm).generate(p, 16, use_graph=mode == "graph")
    assert got[:n].tolist() == want[:n].tolist(), (name, mode, got.tolist(), want.tolist(), n)


# ---------------------------------------------------------------------------------------------------------
# run-time switches: every fallback path the library keeps must be a correct path (or be deleted)
# ---------------------------------------------------------------------------------------------------------
SWITCHES = [
    # (env, cases whose kernels the switch changes)
    ("BZ_NO_GQ_SLIM=1", ["mistral-7b-q4km-2l", "tiny-q4km"]),            # generic k_gemv_gq instead of the slim one: cross-checks the slim kernels
    ("BZ_NO_ATTN_F32=1", ["mistral-7b-q4km-2l"]),                        # generic attention instead of k_attn2f
    ("BZ_NO_GQ_MIX=1", ["mistral-7b-q4km-2l"]),                          # q/k (Q4_K) and v (Q6_K) as two slim launches instead of the mixed-format one
    ("BZ_NO_SLIM_QKV=1", ["llama3-8b-awq-2l", "tiny-awq"]),              # k_gemv_q4g instead of k_gemv_q4g_slim
    ("BZ_NO_MLP_FUSION=1", ["llama3-8b-awq-2l", "tiny-awq", "llama3.2-1b-bf16-2l"]),   # gate/up and down as two launches
    ("BZ_NO_ATTN_FUSION=1", ["llama3-8b-awq-2l", "tiny-awq", "mistral-7b-q4km-2l", "llama3.2-1b-bf16-2l"]),   # attention and o_proj as two launches
    ("BZ_NO_ATTN_SPLIT=1 BZ_SPLIT_MIN=4", ["tiny-bf16"]),                # long-context split disabled
    ("BZ_SPLIT_MIN=4", ["llama3-8b-awq-2l", "tiny-bf16"]),               # split-KV attention from position 4 on
    ("BZ_NO_ROWS_SPLITK=1", ["mamba2-2.7b-2l", "llama3.2-1b-bf16-2l"]),  # row GEMVs without split-K
    ("BZ_NO_ATTN2_HD64=1", ["llama3.2-1b-bf16-2l", "tiny-bf16"]),         # one-thread-per-position attention instead of k_attn2<head_dim 64>
    ("BZ_NO_ROWS2=1", ["mamba2-2.7b-2l", "llama3.2-1b-bf16-2l", "deepseek-v2-lite-2l"]),   # 16-row workgroup GEMV (k_gemv_rows) instead of the balanced role kernel
    ("BZ_NO_MLA_SPLIT=1", ["deepseek-v2-lite-2l"]),                      # MLA decode: one workgroup per head over the whole context
    ("BZ_NO_MOE_ROUTE_FUSION=1", ["deepseek-v2-lite-2l"]),               # router launch (last-workgroup top-k) + plain grouped gate/up
    ("BZ_NO_MFMA_PREFILL=1", ["mamba2-2.7b-2l", "tiny-bf16"]),           # prompts token by token
    ("BZ_NO_Q4G_MFMA=1", ["llama3-8b-awq-2l"]),                          # int4 prompts through the multi-row dot4 kernel
    ("BZ_Q4G_MFMA_NO_LDS=1", ["llama3-8b-awq-2l", "awq-h2048"]),          # the 20-row prompt chunk on the single-wave k_gemm_q4g_mfma (default from 17 rows: the LDS kernel)
    ("BZ_Q4G_LDS_MIN=9", ["llama3-8b-awq-2l", "awq-h2048", "tiny-awq"]),  # k_gemm_q4g_lds (wide form) from 9 rows instead of 17
    ("BZ_NO_PF_ATTN_MFMA=1", ["llama3.2-1b-bf16-2l", "tiny-bf16"]),      # prompt attention on the scalar kernel (scores in LDS) instead of k_pf_attn_mfma
    ("BZ_GEMM_NT_WAVE_TILES=1", ["llama3.2-1b-bf16-2l", "mamba2-2.7b-2l", "deepseek-v2-lite-2l"]),   # prefill GEMMs on the register-staged wave-tile kernel instead of the LDS-DMA one
    ("BZ_GGUF_MLP_FUSION=1", ["mistral-7b-q4km-2l", "q4km-h2048"]),       # opt-in fused GGUF MLP (Q4_K gate/up + Q4_K / Q6_K down in one launch)
    ("BZ_COLS_QKV=1", ["llama3-8b-awq-2l", "awq-h2048"]),                # opt-in full-K direct-output q/k/v kernel
    ("BZ_PREFILL_MIN=4", ["tiny-bf16", "tiny-awq"]),
    ("BZ_GEMV_TARGET_WGS=96", ["tiny-awq", "tiny-q4km"]),
]


@pytest.mark.parametrize("env,cases", SWITCHES, ids=[s[0].replace(" ", "+") for s in SWITCHES])
def test_switch_paths_match_the_oracle(env, cases, tmp_path):
    e = dict(os.environ)
    for kv in env.split():
        k, v = kv.split("=")
        e[k] = v
    for c in cases:
        out = tmp_path / (c + ".npz")
        r = subprocess.run([sys.executable, os.path.join(HERE, "switch_probe.py"), c, str(out)], env=e, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (env, c, r.stdout[-2000:], r.stderr[-2000:])
        d = np.load(out)
        act = str(d["act"])
        factor = 2.0 if c.startswith("tiny") else _factor(act)
        _check_logits(d["got"], d["want"], act, factor=factor)
        n = int(d["fair"])
        assert d["ids_got"][:n].tolist() == d["ids_want"][:n].tolist(), (env, c, d["ids_got"].tolist(), d["ids_want"].tolist(), n)
