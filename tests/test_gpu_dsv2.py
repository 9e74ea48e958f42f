"""GPU parity for the DeepSeek-V2 path: MLA attention over the compressed-latent cache + MoE, against
oracle/orc_dsv2.c (weight-absorbed form; itself checked against a naive HF-form numpy model in tests/test_oracle.py).

The arithmetic lives in the absent boostr crate (parity unpinned, see oracle/orc_dsv2.c).  Bars as in test_gpu_llama.py.
"""
import numpy as np
import pytest

from blazr_amd import _lib as L
from blazr_amd import runtime, synth
from oracle import orc_py
from test_gpu_llama import _check_logits, _fair_prefix

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["tiny-dsv2", "tiny-dsv2-f32"])
def pair(request, device):
    model = synth.make_dsv2(request.param)
    return model, runtime.LoadedModel.from_synth(device, model), orc_py.OrcDsv2(model)


def test_accessors(pair):
    model, lm, _ = pair
    cfg = model["config"]
    assert lm.needs_kv_cache() and not lm.needs_ssm_state()
    assert lm.num_kv_heads() == 1 and lm.head_dim() == cfg["kv_lora_rank"] + cfg["rope_dim"]     # latent cache row
    assert lm.moe_config()["experts_per_tok"] == cfg["top_k"]
    assert lm.weight_bytes()[1] == synth.dsv2_bytes_per_token(cfg)


def test_prefill_and_decode_logits(pair, device):
    model, lm, om = pair
    cfg = model["config"]
    p = synth.prompt_tokens(9, cfg["vocab"])
    kv, okc = lm.new_kv_cache(12), om.new_cache(64)
    got = lm.forward_with_kv_cache(p, kv, 0, all_logits=True).to_numpy()
    want = om.forward(p, okc, 0, all_logits=True)
    _check_logits(got, want, cfg["act_dtype"])
    assert kv.seq_len() == 9
    # the latent cache itself (normalised latents | roped k_pe, rounded to the cache dtype)
    W = cfg["kv_lora_rank"] + cfg["rope_dim"]
    olat = np.ctypeslib.as_array((orc_py.C.c_float * (cfg["n_layers"] * 64 * W)).from_address(okc.contents.lat)).reshape(cfg["n_layers"], 64, W)
    glat = kv.read(1, 0, 0, 9)
    assert np.abs(glat - olat[1, :9]).max() <= 2 * {"bf16": 2 ** -7, "f32": 1e-5}[cfg["act_dtype"]] * np.abs(olat[1, :9]).max()
    tok = int(want[-1].argmax())
    for i in range(20):      # cache grows 12 -> 24 -> 48 under the decode
        lg = lm.forward_with_kv_cache([tok], kv, kv.seq_len()).to_numpy()
        lo = om.forward([tok], okc, 9 + i)
        _check_logits(lg, lo, cfg["act_dtype"])
        tok = int(lo[0].argmax())
    orc_py.lib().orc_mla_cache_free(okc)


@pytest.mark.parametrize("mode", ["eager", "graph"])
def test_generate_greedy_token_parity(pair, mode):
    model, lm, om = pair
    cfg = model["config"]
    for seed in range(3, 60):
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
    wrong = runtime.LayeredKvCache(device, cfg["n_layers"], 1, 2, 8, 64, 64, lm.c.act_dtype)
    with pytest.raises(L.BlazrHipError):
        lm.forward_with_kv_cache([1, 2], wrong, 0)                     # not the latent-cache shape
    bad = dict(cfg, q_lora_rank=63)
    with pytest.raises(L.BlazrHipError):
        runtime.LoadedModel(device, bad)                               # not a multiple of 8


_YARN = dict(type="yarn", factor=40, original_max_position_embeddings=48, beta_fast=32, beta_slow=1, mscale=0.707, mscale_all_dim=0.707)


@pytest.mark.parametrize("over", [dict(n_shared=0), dict(first_dense=0, n_layers=2), dict(top_k=8), dict(n_experts=0, first_dense=3), dict(rope_scaling=_YARN),
                                  dict(kv_lora_rank=64, nope_dim=32, v_dim=128, rope_dim=16), dict(act_dtype="f16"), dict(q_lora_rank=96),
                                  dict(q_lora_rank=64, act_dtype="f32")],
                         ids=["no-shared", "all-moe", "topk-all", "dense-only", "yarn-mscale", "odd-mla-dims", "f16", "q-lora", "q-lora-f32"])
def test_config_variants(device, over):
    # edges of the MoE / MLA configuration space against the oracle (prefill rows + a few decode steps)
    model = synth.make_dsv2("tiny-dsv2", **over)
    cfg = model["config"]
    lm, om = runtime.LoadedModel.from_synth(device, model), orc_py.OrcDsv2(model)
    p = synth.prompt_tokens(6, cfg["vocab"], seed=17)
    kv, okc = lm.new_kv_cache(16), om.new_cache(16)
    _check_logits(lm.forward_with_kv_cache(p, kv, 0, all_logits=True).to_numpy(), om.forward(p, okc, 0, all_logits=True), cfg["act_dtype"])
    tok = 5
    for i in range(4):
        lo = om.forward([tok], okc, 6 + i)
        _check_logits(lm.forward_with_kv_cache([tok], kv, 6 + i).to_numpy(), lo, cfg["act_dtype"])
        tok = int(lo[0].argmax())
    orc_py.lib().orc_mla_cache_free(okc)


@pytest.mark.parametrize("mode", ["paged", "paged-graph"])
def test_paged_latent_cache_generation_equals_contiguous(pair, mode):
    """executor_generate.rs:182-340: the paged branch applies to any KV model; the MLA latent cache pages as one 'head' of rank + rope values"""
    model, lm, om = pair
    cfg = model["config"]
    p = synth.prompt_tokens(11, cfg["vocab"], seed=21)
    ex = runtime.Executor(lm)
    base = ex.generate(p, 20)
    got = ex.generate(p, 20, paged=True, use_graph="graph" in mode)
    assert got.tolist() == base.tolist(), mode


def test_paged_latent_cache_forward_scattered_blocks(pair, device):
    """prompt chunk (batched rows where the model has that path) + decode steps over a scattered block table == the contiguous cache, bit for bit"""
    model, lm, om = pair
    cfg = model["config"]
    W = cfg["kv_lora_rank"] + cfg["rope_dim"]
    p = synth.prompt_tokens(13, cfg["vocab"], seed=4)
    kv = lm.new_kv_cache(32)
    a = lm.forward_with_kv_cache(p, kv, 0, all_logits=True).to_numpy()
    pk = runtime.LayeredPagedKvCache(device, cfg["n_layers"], 9, 4, 1, W, lm.c.act_dtype)
    pk.set_blocks([7, 2, 5, 0, 8, 3, 1])
    sm = pk.compute_slot_mapping(0, len(p))
    pk.set_seq_len(len(p))
    b = lm.forward_with_paged_kv_cache(p, pk, sm, pk.block_table_device_format(), len(p), 0, all_logits=True).to_numpy()
    assert np.array_equal(a, b)
    tok = int(a[-1].argmax())
    for i in range(6):
        n = len(p) + i + 1
        pk.set_seq_len(n)
        x = lm.forward_with_kv_cache([tok], kv, n - 1).to_numpy()
        y = lm.forward_with_paged_kv_cache([tok], pk, pk.compute_slot_mapping(n - 1, 1), pk.block_table_device_format(), n, n - 1).to_numpy()
        assert np.array_equal(x, y), i
        tok = int(x[0].argmax())


_TILE_CHILD = r"""
import sys
import numpy as np
from blazr_amd import _lib as L, runtime, synth
preset, over, S = sys.argv[2], eval(sys.argv[3]), int(sys.argv[4])
model = synth.make_dsv2(preset, **over)
cfg = model["config"]
dev = runtime.Device(0)
lm = runtime.LoadedModel.from_synth(dev, model)
p = synth.prompt_tokens(S, cfg["vocab"], seed=41)
kv = lm.new_kv_cache(S + 40)
a = lm.forward_with_kv_cache(p, kv, 0, all_logits=True).to_numpy()
p2 = synth.prompt_tokens(21, cfg["vocab"], seed=42)          # a second chunk behind the first: keys before the chunk + the causal part
b = lm.forward_with_kv_cache(p2, kv, S, all_logits=True).to_numpy()
np.save(sys.argv[1], np.concatenate([a.reshape(S, -1), b.reshape(21, -1)]))
"""


@pytest.mark.parametrize("preset,over,S", [("tiny-dsv2", "{}", 37), ("deepseek-v2-lite", "dict(n_layers=2, vocab=2048)", 70)], ids=["tiny", "v2-lite-widths"])
def test_token_tiled_mla_prompt_kernel_is_the_head_token_kernel_bit_for_bit(tmp_path, preset, over, S):
    """k_mla_attn_tile (4 or 8 tokens of one head per workgroup: weight rows and latent rows loaded once per tile) against k_mla_attn<BATCH> (one workgroup per
    (head, token)): the same sums in the same order, so every logit of every prompt row is the same bits -- first chunk and a second chunk at position S"""
    import os
    import subprocess
    import sys
    outs = {}
    for name, env in (("tile4", {"BZ_MLA_TILE": "4"}), ("tile8", {"BZ_MLA_TILE": "8"}), ("default", {}), ("head_token", {"BZ_NO_MLA_TILE": "1"})):
        f = str(tmp_path / (name + ".npy"))
        e = dict(os.environ)
        e.update(env)
        e["PYTHONPATH"] = os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + os.pathsep + e.get("PYTHONPATH", "")
        r = subprocess.run([sys.executable, "-c", _TILE_CHILD, f, preset, over, str(S)], env=e, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[name] = np.load(f)
    for name in ("tile4", "tile8", "default"):
        assert np.array_equal(outs[name], outs["head_token"]), (name, int((outs[name] != outs["head_token"]).sum()))
