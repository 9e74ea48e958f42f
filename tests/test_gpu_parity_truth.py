"""Whose error is it?  Both the CPU oracle and the HIP path round every activation to the activation dtype; a 1e-6 difference in summation
order flips a fraction of those roundings, so two correct implementations differ by ~1e-3 of the logit range at the real widths (the bar in
test_gpu_llama.py is stated on relative L2 for that reason).  This test replaces the argument by a measurement: an UNROUNDED float64
evaluation of the same model (tests/npref.py::NpLlamaTruth, exact dequantised weights, no rounding anywhere) is the thing both approximate,
and the HIP path must be no further from it than the oracle is (factor 1.25, L2 and max-norm).  The measured distances
are printed in the assertion message either way.
"""
import numpy as np
import pytest

from blazr_amd import _lib as L
from blazr_amd import runtime, synth
from oracle import orc_py
import npref

pytestmark = pytest.mark.gpu
FLOOR = 4e-6


@pytest.mark.parametrize("preset,over", [("llama3-8b-awq-2l", {}), ("tiny-awq", {}), ("mistral-7b-q4km", dict(n_layers=2, vocab=8192, max_seq_len=512))],
                         ids=["llama3-8b-awq-2l", "tiny-awq", "mistral-7b-q4km-2l"])
def test_hip_is_as_close_to_the_unrounded_truth_as_the_oracle(device, preset, over):
    model = synth.make_llama(preset, **over)
    cfg = model["config"]
    lm, om, tm = runtime.LoadedModel.from_synth(device, model), orc_py.OrcLlama(model), npref.NpLlamaTruth(model)
    dt = {"f16": L.F16, "bf16": L.BF16, "f32": L.F32}[cfg["act_dtype"]]
    kv = runtime.LayeredKvCache(device, cfg["n_layers"], 1, cfg["n_kv_heads"], 16, cfg["max_seq_len"], cfg["head_dim"], dt)
    okv = om.new_kv(16)
    p = synth.prompt_tokens(10, cfg["vocab"], seed=2)
    G, O, T = [], [], []
    for i, t in enumerate(p):      # token by token: the decode kernels
        G.append(lm.forward_with_kv_cache([int(t)], kv, i).to_numpy().reshape(-1).astype(np.float64))
        O.append(np.asarray(om.forward_kv([int(t)], okv, i)).reshape(-1).astype(np.float64))
        T.append(tm.step(int(t), i))
    orc_py.lib().orc_kv_free(okv)
    G, O, T = np.stack(G), np.stack(O), np.stack(T)
    nT = np.linalg.norm(T)
    g2, o2 = np.linalg.norm(G - T) / nT, np.linalg.norm(O - T) / nT
    gm, omx = np.abs(G - T).max() / np.abs(T).max(), np.abs(O - T).max() / np.abs(T).max()
    go2, gom = np.linalg.norm(G - O) / np.linalg.norm(O), np.abs(G - O).max() / np.abs(O).max()
    msg = ("%s: relative L2 to the f64 truth: hip %.3e, oracle %.3e; max-norm (of the logit range): hip %.3e, oracle %.3e; hip vs oracle: L2 %.3e, max %.3e"
           % (preset, g2, o2, gm, omx, go2, gom))
    print(msg)
    # f32-activation models (GGUF) have no activation rounding: both distances are summation-order noise of f32 dot products (~1e-6), where
    # a ratio says nothing -- there the absolute floor applies (three orders of magnitude under the 1e-3 bar)
    assert g2 <= max(1.25 * o2, FLOOR), msg
    assert gm <= max(1.25 * omx, 3 * FLOOR), msg


def test_dsv2_full_width_against_the_unrounded_truth(device):
    """DeepSeek-V2-Lite widths (bf16): token-by-token decode rows AND the batched-prefill rows of a second chunk, each held to 1.25x the oracle's own
    distance from the float64 truth (naive HF form, no weight absorption, no rounding)"""
    from fullwidth_cases import GpuRun, OrcRun, make
    fam, model = make("deepseek-v2-lite-2l")
    cfg = model["config"]
    g, o, tm = GpuRun(device, fam, model), OrcRun(fam, model, cap=64), npref.NpDsv2(model, truth=True)
    p = synth.prompt_tokens(6, cfg["vocab"], seed=2)
    p2 = synth.prompt_tokens(20, cfg["vocab"], seed=9)
    G = np.concatenate([g.forward(p, all_logits=True), g.forward(p2, all_logits=True)]).astype(np.float64)     # 6 decode-path rows + 20 prefill-path rows
    O = np.concatenate([o.forward(p, all_logits=True), o.forward(p2, all_logits=True)]).astype(np.float64)
    T = np.stack([tm.step(int(t), i) for i, t in enumerate(list(p) + list(p2))])
    for name, sl in (("decode rows", slice(0, 6)), ("prefill rows", slice(6, 26))):
        nT = np.linalg.norm(T[sl])
        g2, o2 = np.linalg.norm(G[sl] - T[sl]) / nT, np.linalg.norm(O[sl] - T[sl]) / nT
        gm, om = np.abs(G[sl] - T[sl]).max() / np.abs(T[sl]).max(), np.abs(O[sl] - T[sl]).max() / np.abs(T[sl]).max()
        msg = "deepseek-v2-lite-2l %s: relative L2 to the f64 truth: hip %.3e, oracle %.3e; max-norm: hip %.3e, oracle %.3e; hip vs oracle L2 %.3e" % (
            name, g2, o2, gm, om, np.linalg.norm(G[sl] - O[sl]) / np.linalg.norm(O[sl]))
        print(msg)
        assert g2 <= 1.25 * o2 and gm <= 1.5 * om, msg
