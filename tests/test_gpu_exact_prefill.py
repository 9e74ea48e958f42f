"""GPU parity: the EXACT prompt rows of f16 int4 models (bz_host.hip `prefill_exact`).
Prompts of up to BZ_EXACT_PREFILL_MAX (16) rows -- and every prompt under BZ_EXACT_PREFILL=1 -- run the multi-row form of the decode kernels' integer
arithmetic (8 rows per pass over the weights), the exact scalar attention (double-precision sums, IEEE division) and the decode lm_head, so a prompt row is
the SAME BITS as the decode step's row and sits where the decode rows sit against the oracle.  Longer prompts take the MFMA GEMMs, whose f32 accumulation
order differs (at depth, the f16 roundings amplify any difference to the f16 noise floor).
"""
import os
import subprocess
import sys

import numpy as np
import pytest

from blazr_amd import _lib as L
from blazr_amd import runtime, synth
from oracle import orc_py

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


CASES = [("tiny-awq", {}, 16), ("tiny-gptq", {}, 12), ("llama3-8b-awq-2l", dict(vocab=4096), 16), ("llama3-8b-awq-2l", dict(vocab=4096, n_layers=6), 13)]


@pytest.mark.parametrize("preset,over,S", CASES, ids=["%s-%d" % (c[0], c[2]) for c in CASES])
def test_short_prompt_rows_are_the_decode_rows_bit_for_bit(device, preset, over, S):
    model = synth.make_llama(preset, **over)
    cfg = model["config"]
    lm, om = runtime.LoadedModel.from_synth(device, model), orc_py.OrcLlama(model)
    nl, nkv = cfg["n_layers"], cfg["n_kv_heads"]
    p = [int(t) for t in synth.prompt_tokens(S, cfg["vocab"], seed=31)]
    mk = lambda: runtime.LayeredKvCache(device, nl, 1, nkv, S + 8, cfg["max_seq_len"], cfg["head_dim"], L.F16)
    kv_a, kv_b = mk(), mk()
    got = lm.forward_with_kv_cache(p, kv_a, 0, all_logits=True).to_numpy().reshape(S, -1)                        # one call: the batched (exact) rows
    step = np.stack([lm.forward_with_kv_cache([t], kv_b, i).to_numpy().reshape(-1) for i, t in enumerate(p)])     # one call per token: the decode kernels
    assert np.array_equal(got, step), "%d of %d logits differ" % (int((got != step).sum()), got.size)
    for l in range(nl):
        for which in (0, 1):
            a = np.concatenate([kv_a.read(l, h, which, S).reshape(-1) for h in range(nkv)])
            b = np.concatenate([kv_b.read(l, h, which, S).reshape(-1) for h in range(nkv)])
            assert np.array_equal(a, b), (l, which, int((a != b).sum()))
    okv = om.new_kv(S + 8)
    want = om.forward_kv(p, okv, 0, all_logits=True).reshape(S, -1)
    worst = max(_rel(got[i], want[i]) for i in range(S))
    print("%s: %d prompt rows == decode rows; vs oracle worst row %.2e" % (preset, S, worst))
    assert worst <= 1e-4          # what is left is the dense f16 lm_head's f32 summation order (and rare 2^-44-grid ties), as for the decode rows
    orc_py.lib().orc_kv_free(okv)


_CHILD = r"""
import sys
import numpy as np
from blazr_amd import _lib as L, runtime, synth
model = synth.make_llama("llama3-8b-awq-2l", vocab=2048)
cfg = model["config"]
dev = runtime.Device(0)
lm = runtime.LoadedModel.from_synth(dev, model)
S = 70
p = [int(t) for t in synth.prompt_tokens(S, cfg["vocab"], seed=32)]
mk = lambda: runtime.LayeredKvCache(dev, cfg["n_layers"], 1, cfg["n_kv_heads"], 80, cfg["max_seq_len"], cfg["head_dim"], L.F16)
kv_a, kv_b = mk(), mk()
a = lm.forward_with_kv_cache(p, kv_a, 0, all_logits=True).to_numpy().reshape(S, -1)
b = np.stack([lm.forward_with_kv_cache([t], kv_b, i).to_numpy().reshape(-1) for i, t in enumerate(p)])
print("DIFF", int((a != b).sum()), float(np.linalg.norm(a.astype(np.float64) - b) / np.linalg.norm(b)))
"""


def _child(env):
    e = dict(os.environ)
    e.update(env)
    e["PYTHONPATH"] = os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + os.pathsep + e.get("PYTHONPATH", "")
    r = subprocess.run([sys.executable, "-c", _CHILD], env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("DIFF")][0].split()
    return int(line[1]), float(line[2])


def test_forced_exact_rows_for_a_long_prompt_and_the_mfma_rows_beside_them():
    n_forced, _ = _child({"BZ_EXACT_PREFILL": "1"})
    assert n_forced == 0                                   # 70 rows, 9 passes: bit-identical to the decode rows
    n_mfma, rel_mfma = _child({})
    assert n_mfma > 0 and rel_mfma <= 2e-3                 # the matrix-core rows: same values to the f16 noise floor of a 2-layer model, not the same bits


@pytest.mark.parametrize("wpb", ["", "2", "4", "8"])
def test_exact_gemm_on_the_integer_matrix_cores(wpb):
    """BZ_EXACT_PREFILL=2: every projection of a 70-token prompt through k_gemm_q4g_i8 (four int8 digit planes of the 32-bit activation codes x sign-extended nibbles on
    v_mfma_i32_32x32x32_i8, exact int32 plane sums, the decode kernels' per-group fold in double) + the exact attention: 0 of 70 x vocab logits differ from the decode
    rows, for each workgroup shape (2 / 4 / 8 waves side by side on the same 32 rows)"""
    env = {"BZ_EXACT_PREFILL": "2"}
    if wpb:
        env["BZ_I8_WPB"] = wpb
    n, _ = _child(env)
    assert n == 0


@pytest.mark.parametrize("preset,over", [("llama3.2-1b-bf16", dict(n_layers=4, vocab=4096)), ("tiny-bf16", {})], ids=["llama3.2-1b-widths", "tiny-bf16"])
def test_dense_16bit_llama_is_the_oracle_bit_for_bit(device, preset, over):
    """BASELINE config 0 (dense bf16 SafeTensors): the decode GEMVs carry their sums in double over exact products (bz_kernels.hip piece_dot_d: q/k/v, o_proj, gate / up, down,
    lm_head), which is the oracle's definition of a linear layer; with the exact norm, RoPE, attention and SiLU of the int4 path every logit of a 12-token prompt (short prompts
    stay on the decode kernels) and of 8 decode steps equals the oracle's.  Round 2 held this config to the bf16 noise floor (2e-2 at 16 layers)."""
    model = synth.make_llama(preset, **over)
    cfg = model["config"]
    lm, om = runtime.LoadedModel.from_synth(device, model), orc_py.OrcLlama(model)
    p = [int(t) for t in synth.prompt_tokens(12, cfg["vocab"], seed=51)]
    kv = runtime.LayeredKvCache(device, cfg["n_layers"], 1, cfg["n_kv_heads"], 32, cfg["max_seq_len"], cfg["head_dim"], L.BF16)
    okv = om.new_kv(32)
    got = [lm.forward_with_kv_cache(p, kv, 0, all_logits=True).to_numpy().reshape(12, -1)]
    want = [om.forward_kv(p, okv, 0, all_logits=True).reshape(12, -1)]
    tok = int(want[0][-1].argmax())
    for i in range(8):
        got.append(lm.forward_with_kv_cache([tok], kv, 12 + i).to_numpy().reshape(1, -1))
        want.append(om.forward_kv([tok], okv, 12 + i).reshape(1, -1))
        tok = int(want[-1][0].argmax())
    got, want = np.concatenate(got), np.concatenate(want)
    ndiff = int((got != want).sum())
    print("%s: %d of %d logits differ, rel L2 %.2e" % (preset, ndiff, got.size, _rel(got, want)))
    assert ndiff <= got.size // 10000 and _rel(got, want) <= 1e-5      # (a sum within 1e-16 of a rounding boundary may still land on the other side)
    orc_py.lib().orc_kv_free(okv)


@pytest.mark.parametrize("preset,over", [("tiny-mamba2", {}), ("tiny-mamba2-g2", {}), ("mamba2-2.7b", dict(n_layers=3, vocab=4096))], ids=["tiny-bf16", "tiny-f32-2groups", "2.7b-widths"])
def test_mamba2_steps_are_the_oracle_bit_for_bit(device, preset, over):
    """BASELINE config 3 (Mamba2): with the exact dense GEMVs (in_proj, out_proj, lm_head) and the step kernel's sums defined as the oracle defines them -- conv window and C . h
    readout exactly rounded (double over exact products), h dA + (dt x) B summed in double, softplus through a double log1p, the gated norm's sum of squares exact (per-head
    hi + lo partials) -- a 10-token prompt (short prompts stay on the step kernels) and 8 decode steps give the oracle's logits."""
    model = synth.make_mamba2(preset, **over)
    cfg = model["config"]
    lm, om = runtime.LoadedModel.from_synth(device, model), orc_py.OrcMamba2(model)
    p = [int(t) for t in synth.prompt_tokens(10, cfg["vocab"], seed=61)]
    st, ost = runtime.LayeredSsmState(lm), om.new_state()
    got = [lm.forward_with_ssm_state(p, st, all_logits=True).to_numpy().reshape(10, -1)]
    want = [om.forward(p, ost, all_logits=True).reshape(10, -1)]
    tok = int(want[0][-1].argmax())
    for i in range(8):
        got.append(lm.forward_with_ssm_state([tok], st).to_numpy().reshape(1, -1))
        want.append(om.forward([tok], ost).reshape(1, -1))
        tok = int(want[-1][0].argmax())
    got, want = np.concatenate(got), np.concatenate(want)
    ndiff = int((got != want).sum())
    print("%s: %d of %d logits differ, rel L2 %.2e" % (preset, ndiff, got.size, _rel(got, want)))
    if cfg["act_dtype"] == "f32":      # f32 activations keep the 2^-32 fixed-point grid between launches (range for GGUF-scale values): f32-rounding-level differences
        assert _rel(got, want) <= 1e-6
    else:
        assert ndiff <= got.size // 1000 and _rel(got, want) <= 1e-4
    orc_py.lib().orc_ssm_state_free(ost)


@pytest.mark.parametrize("preset,over", [("tiny-dsv2", {}), ("deepseek-v2-lite", dict(n_layers=3, vocab=4096))], ids=["tiny-bf16", "v2-lite-widths"])
def test_deepseek_v2_steps_against_the_oracle(device, preset, over):
    """BASELINE config 4 (DeepSeek-V2, MLA + MoE): exact GEMVs (router and experts included), the exact decode MLA (k_mla_attn_x: double sums, one maximum over the whole
    context shared by the context slices) and the combine as the oracle's f32 chain -- a 10-token prompt (short prompts stay on the decode kernels) and 8 decode steps"""
    model = synth.make_dsv2(preset, **over)
    cfg = model["config"]
    lm, om = runtime.LoadedModel.from_synth(device, model), orc_py.OrcDsv2(model)
    p = [int(t) for t in synth.prompt_tokens(10, cfg["vocab"], seed=71)]
    kv, okc = lm.new_kv_cache(32), om.new_cache(64)
    got = [lm.forward_with_kv_cache(p, kv, 0, all_logits=True).to_numpy().reshape(10, -1)]
    want = [om.forward(p, okc, 0, all_logits=True).reshape(10, -1)]
    tok = int(want[0][-1].argmax())
    for i in range(8):
        got.append(lm.forward_with_kv_cache([tok], kv, 10 + i).to_numpy().reshape(1, -1))
        want.append(om.forward([tok], okc, 10 + i).reshape(1, -1))
        tok = int(want[-1][0].argmax())
    got, want = np.concatenate(got), np.concatenate(want)
    ndiff = int((got != want).sum())
    print("%s: %d of %d logits differ, rel L2 %.2e" % (preset, ndiff, got.size, _rel(got, want)))
    assert ndiff <= got.size // 1000 and _rel(got, want) <= 1e-4
    orc_py.lib().orc_mla_cache_free(okc)
