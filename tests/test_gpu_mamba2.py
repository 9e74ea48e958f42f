"""GPU parity for the Mamba2 path: forward_with_ssm_state, the recurrent state, the
generate loop (executor_generate.rs:123-181) and the hipGraph decode, against oracle/orc_mamba2.c.

The recurrence lives in the absent boostr crate; the oracle restates the public Mamba2 single-token step (parity unpinned, see
oracle/orc_mamba2.c).  Bars as in test_gpu_llama.py: relative L2 of the logits, greedy ids bit-exact on fair prefixes.
"""
import ctypes as C

import numpy as np
import pytest

from blazr_amd import _lib as L
from blazr_amd import runtime, synth
from oracle import orc_py
from test_gpu_llama import _check_logits, _fair_prefix

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["tiny-mamba2", "tiny-mamba2-g2"])
def pair(request, device):
    model = synth.make_mamba2(request.param)
    return model, runtime.LoadedModel.from_synth(device, model), orc_py.OrcMamba2(model)


def test_accessors(pair):
    model, lm, _ = pair
    assert lm.needs_ssm_state() and not lm.needs_kv_cache()
    assert lm.mamba_config()["d_state"] == model["config"]["d_state"]
    resident, per_token = lm.weight_bytes()
    assert per_token == synth.mamba2_bytes_per_token(model["config"])[0]


def test_prefill_and_decode_logits(pair, device):
    model, lm, om = pair
    cfg = model["config"]
    p = synth.prompt_tokens(7, cfg["vocab"])
    st, ost = runtime.LayeredSsmState(lm), om.new_state()
    got = lm.forward_with_ssm_state(p, st, all_logits=True).to_numpy()
    want = om.forward(p, ost, all_logits=True)
    _check_logits(got, want, cfg["act_dtype"])
    tok = int(want[-1].argmax())
    for _ in range(24):   # the state keeps evolving: errors must not accumulate
        lg = lm.forward_with_ssm_state([tok], st).to_numpy()
        lo = om.forward([tok], ost)
        _check_logits(lg, lo, cfg["act_dtype"])
        tok = int(lo[0].argmax())
    orc_py.lib().orc_ssm_state_free(ost)


def test_state_reset_restarts_the_sequence(pair):
    model, lm, _ = pair
    p = synth.prompt_tokens(5, model["config"]["vocab"], seed=3)
    st = runtime.LayeredSsmState(lm)
    a = lm.forward_with_ssm_state(p, st).to_numpy()
    b = lm.forward_with_ssm_state(p, st).to_numpy()      # continues from the evolved state
    st.reset()
    c = lm.forward_with_ssm_state(p, st).to_numpy()
    assert np.array_equal(a, c) and not np.array_equal(a, b)


@pytest.mark.parametrize("mode", ["eager", "graph"])
def test_generate_greedy_token_parity(pair, mode):
    model, lm, om = pair
    cfg = model["config"]
    for seed in range(3, 40):
        p = synth.prompt_tokens(12, cfg["vocab"], seed=seed)
        want, trace = om.generate(p, 24, trace=True)
        n = _fair_prefix(trace)
        if n >= 8:
            break
    assert n >= 8, "no prompt seed gives a fair fixture"
    got = runtime.Executor(lm).generate(p, 24, use_graph=mode == "graph")
    assert got[:n].tolist() == want[:n].tolist(), (mode, got.tolist(), want.tolist(), n)


def test_graph_replay_equals_eager(pair):
    model, lm, _ = pair
    p = synth.prompt_tokens(6, model["config"]["vocab"], seed=5)
    ex = runtime.Executor(lm)
    assert ex.generate(p, 16, use_graph=True).tolist() == ex.generate(p, 16).tolist()


def test_error_behaviour(pair, device):
    model, lm, _ = pair
    cfg = model["config"]
    kv = runtime.LayeredKvCache(device, cfg["n_layers"], 1, 1, 8, 64, 64, L.F16)
    with pytest.raises(L.BlazrHipError):
        lm.forward_with_kv_cache([1, 2], kv, 0)                       # no KV cache on an SSM model
    with pytest.raises(L.BlazrHipError):
        runtime.DecodeGraph(lm, kv)
    other = runtime.LoadedModel.from_synth(device, synth.make_mamba2("tiny-mamba2", n_layers=1))
    st = runtime.LayeredSsmState(other)
    with pytest.raises(L.BlazrHipError):
        lm.forward_with_ssm_state([1], st)                            # state of another model
    with pytest.raises(L.BlazrHipError):
        L.check(L.lib().bz_ssm_state_create(lm.h, 2, lm.c.act_dtype, C.byref(C.c_void_p())))   # batch 1 only


def test_llama_model_rejects_ssm_calls(device):
    lm = runtime.LoadedModel.from_synth(device, synth.make_llama("tiny-bf16"))
    with pytest.raises(L.BlazrHipError):
        runtime.LayeredSsmState(lm)


@pytest.mark.parametrize("over", [dict(d_state=16), dict(conv_kernel=2), dict(n_groups=4, d_state=32), dict(act_dtype="f16"), dict(head_dim=32, n_heads=16)],
                         ids=["state16", "conv2", "groups4", "f16", "headdim32"])
def test_config_variants(device, over):
    model = synth.make_mamba2("tiny-mamba2", **over)
    cfg = model["config"]
    lm, om = runtime.LoadedModel.from_synth(device, model), orc_py.OrcMamba2(model)
    p = synth.prompt_tokens(6, cfg["vocab"], seed=19)
    st, ost = runtime.LayeredSsmState(lm), om.new_state()
    _check_logits(lm.forward_with_ssm_state(p, st, all_logits=True).to_numpy(), om.forward(p, ost, all_logits=True), cfg["act_dtype"])
    tok = 7
    for _ in range(6):
        lo = om.forward([tok], ost)
        _check_logits(lm.forward_with_ssm_state([tok], st).to_numpy(), lo, cfg["act_dtype"])
        tok = int(lo[0].argmax())
    orc_py.lib().orc_ssm_state_free(ost)


@pytest.mark.parametrize("over", [dict(), dict(d_state=16), dict(conv_kernel=2), dict(n_groups=4, d_state=64), dict(act_dtype="f16"), dict(head_dim=32, n_heads=16)],
                         ids=["base", "state16", "conv2", "groups4", "f16", "headdim32"])
@pytest.mark.parametrize("S", [8, 21, 70])
def test_batched_prefill_scan_matches_oracle_and_steps(device, over, S):
    """prompts of >= 8 tokens take the batched path (rows through the MFMA GEMMs, conv as a map over (token, channel), the recurrence as an
    in-kernel scan over the tokens); 21 and 70 leave a partial chunk of the scan's 8-token staging.  Checked against the oracle (every
    position's logits), and the state it leaves must continue exactly like the state the token-by-token path leaves: decode steps after the
    prompt are compared with the oracle too."""
    model = synth.make_mamba2("tiny-mamba2", **over)
    cfg = model["config"]
    lm, om = runtime.LoadedModel.from_synth(device, model), orc_py.OrcMamba2(model)
    p = synth.prompt_tokens(S, cfg["vocab"], seed=23 + S)
    st, ost = runtime.LayeredSsmState(lm), om.new_state()
    got = lm.forward_with_ssm_state(p, st, all_logits=True).to_numpy()
    want = om.forward(p, ost, all_logits=True)
    _check_logits(got, want, cfg["act_dtype"])
    # a second prompt continues from the carried conv window and SSM state
    p2 = synth.prompt_tokens(9, cfg["vocab"], seed=5)
    _check_logits(lm.forward_with_ssm_state(p2, st).to_numpy(), om.forward(p2, ost), cfg["act_dtype"])
    tok = int(want[-1].argmax())
    for _ in range(8):
        lo = om.forward([tok], ost)
        _check_logits(lm.forward_with_ssm_state([tok], st).to_numpy(), lo, cfg["act_dtype"])
        tok = int(lo[0].argmax())
    orc_py.lib().orc_ssm_state_free(ost)


def test_batched_prefill_crosses_the_row_chunk(device):
    """prompts longer than the 512-row workspace chunk: the conv window and the SSM state carry from one chunk of rows to the next"""
    model = synth.make_mamba2("tiny-mamba2")
    cfg = model["config"]
    lm, om = runtime.LoadedModel.from_synth(device, model), orc_py.OrcMamba2(model)
    p = synth.prompt_tokens(530, cfg["vocab"], seed=77)
    st, ost = runtime.LayeredSsmState(lm), om.new_state()
    got = lm.forward_with_ssm_state(p, st).to_numpy()
    want = om.forward(p, ost)
    _check_logits(got, want, cfg["act_dtype"])
    tok = int(want[-1].argmax())
    for _ in range(4):
        lo = om.forward([tok], ost)
        _check_logits(lm.forward_with_ssm_state([tok], st).to_numpy(), lo, cfg["act_dtype"])
        tok = int(lo[0].argmax())
    orc_py.lib().orc_ssm_state_free(ost)
