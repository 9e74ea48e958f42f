"""Checkpoint ingestion end to end: a synthetic checkpoint is written in the on-disk format the reference loads, read back by
bz_load_model (detect -> config -> add tensors -> finalize), and must give bit-identical logits to the same model handed over tensor by tensor."""
import numpy as np
import pytest

from blazr_amd import _lib as L
from blazr_amd import runtime, synth
import ckpt_writer as W

pytestmark = pytest.mark.gpu


def _logits(lm, prompt):
    if lm.needs_ssm_state():
        return lm.forward_with_ssm_state(prompt, runtime.LayeredSsmState(lm), all_logits=True).to_numpy()
    return lm.forward_with_kv_cache(prompt, lm.new_kv_cache(16), 0, all_logits=True).to_numpy()


def _make(preset, over):
    if preset in synth.MAMBA_PRESETS:
        return synth.make_mamba2(preset, **over)
    if preset in synth.DSV2_PRESETS:
        return synth.make_dsv2(preset, **over)
    return synth.make_llama(preset, **over)


@pytest.mark.parametrize("preset,over,shards", [("tiny-bf16", {}, 1), ("tiny-awq", {}, 1), ("tiny-awq", {}, 3), ("tiny-gptq", dict(act_order=True, bias=True), 1),
                                                ("tiny-gptq", {}, 2), ("tiny-mamba2", {}, 1), ("tiny-mamba2-g2", {}, 1), ("tiny-dsv2", {}, 2), ("tiny-dsv2-f32", {}, 1)])
def test_safetensors_checkpoint_equals_in_memory_model(device, tmp_path, preset, over, shards):
    model = _make(preset, over)
    W.write_hf_checkpoint(str(tmp_path), model, shards=shards)
    loaded = runtime.load_model(device, str(tmp_path))
    direct = runtime.LoadedModel.from_synth(device, model)
    p = synth.prompt_tokens(7, model["config"]["vocab"], seed=23)
    assert np.array_equal(_logits(loaded, p), _logits(direct, p))
    assert loaded.weight_bytes() == direct.weight_bytes()


def test_bf16_tensors_of_an_awq_checkpoint_are_cast_to_f16(device, tmp_path):
    # awq.rs:93-103: embeddings / norms stored as BF16 are cast to F16 at load
    model = synth.make_llama("tiny-awq")
    t = W.hf_tensors(model)
    for name in ("model.embed_tokens.weight", "model.norm.weight", "lm_head.weight"):
        f = np.asarray(t[name], dtype=np.float32)
        t[name] = (synth.f32_to_bf16_bits(f))
    import json, os
    W.write_safetensors(str(tmp_path / "model.safetensors"), t)
    json.dump(W.hf_config(model["config"]), open(tmp_path / "config.json", "w"))
    loaded = runtime.load_model(device, str(tmp_path))
    m2 = dict(model)
    r = lambda a: synth.bf16_bits_to_f32(synth.f32_to_bf16_bits(np.asarray(a, np.float32))).astype(np.float16)
    m2["embed"] = r(model["embed"])
    m2["final_norm"] = r(model["final_norm"]).astype(np.float32)
    m2["lm_head"] = dict(model["lm_head"], weight=r(model["lm_head"]["weight"]))
    direct = runtime.LoadedModel.from_synth(device, m2)
    p = synth.prompt_tokens(5, 1024, seed=2)
    assert np.array_equal(_logits(loaded, p), _logits(direct, p))


def test_checkpoint_without_config_json_is_detected_from_tensor_shapes(device, tmp_path):
    # detect_arch.rs:13-63: hidden / vocab from embed_tokens, inter from gate_proj, heads = rows / 128 (default head_dim), defaults for the rest
    model = synth.make_llama("tiny-awq", hidden=256, n_heads=2, n_kv_heads=1, head_dim=128, rope_theta=10000.0, rms_eps=1e-5, max_seq_len=4096)
    W.write_hf_checkpoint(str(tmp_path), model, with_config=False)
    loaded = runtime.load_model(device, str(tmp_path / "model.safetensors"))
    assert (loaded.c.n_heads, loaded.c.n_kv_heads, loaded.c.head_dim, loaded.c.inter, loaded.c.n_layers, loaded.c.act_dtype) == (2, 1, 128, 512, 2, L.F16)
    direct = runtime.LoadedModel.from_synth(device, model)
    p = synth.prompt_tokens(6, 1024, seed=4)
    assert np.array_equal(_logits(loaded, p), _logits(direct, p))


@pytest.mark.parametrize("preset", ["tiny-q4km", "tiny-q8_0"])
def test_gguf_checkpoint_equals_in_memory_model(device, tmp_path, preset):
    model = synth.make_llama(preset)
    path = str(tmp_path / "model.gguf")
    W.write_gguf(path, model)
    loaded = runtime.load_model(device, path)
    direct = runtime.LoadedModel.from_synth(device, model)
    p = synth.prompt_tokens(7, model["config"]["vocab"], seed=29)
    assert np.array_equal(_logits(loaded, p), _logits(direct, p))
    # and greedy generation straight from the file
    assert runtime.Executor(loaded).generate(p, 8).tolist() == runtime.Executor(direct).generate(p, 8).tolist()


@pytest.mark.parametrize("preset", ["tiny-mamba2-g2", "tiny-mamba2"])
def test_mamba2_gguf_checkpoint(device, tmp_path, preset):
    """A Mamba2 GGUF file in llama.cpp's tensor naming (arch "mamba2"; the names the reference maps live in the absent boostr crate, so llama.cpp's converter is the
    convention): metadata -> config (loader/gguf.rs:214-262), ssm_* tensors -> the mixer, ssm_a = -exp(A_log) turned back into A_log.  GGUF files run with f32
    activations (gguf.rs:305), so the in-memory twin is the same weights under act_dtype f32; A_log goes through exp and log once (1 ulp), hence a bar, not bit equality."""
    model = synth.make_mamba2(preset)
    path = str(tmp_path / "mamba2.gguf")
    W.write_gguf_mamba2(path, model)
    cfg, info = runtime.config_from_gguf(path)
    mc = model["config"]
    assert cfg.arch == L.ARCH_MAMBA2 and (cfg.hidden, cfg.n_layers, cfg.ssm_d_inner, cfg.ssm_d_state, cfg.ssm_n_heads, cfg.ssm_head_dim, cfg.ssm_n_groups, cfg.ssm_conv_kernel) == (
        mc["hidden"], mc["n_layers"], mc["d_inner"], mc["d_state"], mc["n_heads"], mc["head_dim"], mc["n_groups"], mc["conv_kernel"])
    loaded = runtime.load_model(device, path)
    assert loaded.c.act_dtype == L.F32 and loaded.c.tie_embeddings == (1 if mc.get("tie_embeddings") else 0)
    twin = dict(model)
    twin["config"] = dict(mc, act_dtype="f32")
    direct = runtime.LoadedModel.from_synth(device, twin)
    p = synth.prompt_tokens(9, mc["vocab"], seed=30)
    a, b = _logits(loaded, p).astype(np.float64), _logits(direct, p).astype(np.float64)
    rel = float(np.linalg.norm(a - b) / np.linalg.norm(b))
    assert rel <= 1e-5, rel
    assert runtime.Executor(loaded).generate(p, 8).tolist() == runtime.Executor(direct).generate(p, 8).tolist()


def test_loader_errors(device, tmp_path):
    model = synth.make_llama("tiny-awq")
    t = W.hf_tensors(model)
    del t["model.layers.1.mlp.down_proj.scales"]                         # incomplete triplet
    W.write_safetensors(str(tmp_path / "model.safetensors"), t)
    import json
    json.dump(W.hf_config(model["config"]), open(tmp_path / "config.json", "w"))
    with pytest.raises(L.BlazrHipError):
        runtime.load_model(device, str(tmp_path))
    t = W.hf_tensors(model)
    t["model.layers.0.self_attn.rotary_emb.inv_freq"] = np.zeros(32, np.float32)     # legacy tensor: skipped
    t["model.layers.0.unknown.weight"] = np.zeros((4, 4), np.float32)                # unknown tensor: loud
    W.write_safetensors(str(tmp_path / "model.safetensors"), t)
    with pytest.raises(L.BlazrHipError):
        runtime.load_model(device, str(tmp_path))
    with pytest.raises(L.BlazrHipError):
        runtime.load_model(device, str(tmp_path / "missing"))


def test_lying_quantised_triplets_and_gguf_extents_are_rejected(device, tmp_path):
    """an AWQ / GPTQ triplet whose shapes disagree, or a GGUF tensor whose offset / extent leaves the file, must fail with an error
    code -- not read out of bounds and upload neighbouring memory as weights"""
    import json
    model = synth.make_llama("tiny-awq")
    json.dump(W.hf_config(model["config"]), open(tmp_path / "config.json", "w"))
    base = "model.layers.1.mlp.down_proj"
    for mutate in (lambda t: t.__setitem__(base + ".scales", t[base + ".scales"][:1]),                 # one group of scales instead of K/128
                   lambda t: t.__setitem__(base + ".qzeros", t[base + ".qzeros"][:, :4]),              # too few zero-point words
                   lambda t: t.__setitem__(base + ".qweight", t[base + ".qweight"][:100]),             # K no multiple of the group size
                   lambda t: t.__setitem__(base + ".scales", t[base + ".scales"].astype(np.float32))): # F32 scales
        t = W.hf_tensors(model)
        mutate(t)
        W.write_safetensors(str(tmp_path / "model.safetensors"), t)
        with pytest.raises(L.BlazrHipError):
            runtime.load_model(device, str(tmp_path))
    gm = synth.make_llama("tiny-gptq", act_order=True, bias=True)
    d2 = tmp_path / "gptq"
    d2.mkdir()
    json.dump(W.hf_config(gm["config"]), open(d2 / "config.json", "w"))
    for mutate in (lambda t: t.__setitem__(base + ".g_idx", t[base + ".g_idx"][:7]), lambda t: t.__setitem__(base + ".bias", t[base + ".bias"][:3])):
        t = W.hf_tensors(gm)
        mutate(t)
        W.write_safetensors(str(d2 / "model.safetensors"), t)
        with pytest.raises(L.BlazrHipError):
            runtime.load_model(device, str(d2))
    # GGUF: patch one tensor's offset / extent in place
    qm = synth.make_llama("tiny-q8_0")
    g = tmp_path / "m.gguf"
    W.write_gguf(str(g), qm)
    raw = bytearray(g.read_bytes())
    name = b"blk.1.ffn_down.weight"
    at = raw.find(name) + len(name)            # n_dims u32, ne[0] u64, ne[1] u64, type u32, offset u64
    import struct
    nd, = struct.unpack_from("<I", raw, at)
    assert nd == 2
    for field_off, value in ((at + 4 + 16 + 4, (1 << 64) - 64), (at + 4 + 8, 1 << 39), (at + 4 + 16 + 4, len(raw))):   # wrapping offset, absurd rows, offset past the end
        bad = bytearray(raw)
        struct.pack_into("<Q", bad, field_off, value)
        (tmp_path / "bad.gguf").write_bytes(bytes(bad))
        with pytest.raises(L.BlazrHipError):
            runtime.load_model(device, str(tmp_path / "bad.gguf"))


def test_bz_run_cpp_driver_generates_the_same_ids(device, tmp_path):
    # tools/bz_run.cpp: the C ABI driven from compiled C++ (no Python in that process), cli/run.rs restated around the hot path
    import os, subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "blazr_amd", "bz-run")
    assert os.path.exists(exe), "bz-run was not built (python -c 'import __graft_entry__ as g; g.build()')"
    model = synth.make_llama("tiny-awq")
    W.write_hf_checkpoint(str(tmp_path), model, shards=2)
    p = synth.prompt_tokens(9, 1024, seed=31)
    want = runtime.Executor(runtime.LoadedModel.from_synth(device, model)).generate(p, 12).tolist()
    for extra in ([], ["--graphs"], ["--paged-attention"]):
        r = subprocess.run([exe, str(tmp_path), "--prompt", ",".join(str(int(x)) for x in p), "--max-tokens", "12", "--stats"] + extra, capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr
        assert [int(x) for x in r.stdout.strip().split(",")] == want, (extra, r.stdout, r.stderr)
    r = subprocess.run([exe, str(tmp_path / "nope"), "--prompt", "1"], capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "bz-run: load" in r.stderr
