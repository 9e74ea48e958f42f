"""The persistent decode launch (blazr_amd/csrc/bz_persist.hip: all layers of a step in ONE launch, weights of the next phase requested before every grid
barrier) against the launch-per-phase path and the oracle.  Same arithmetic and integer accumulators on both GPU paths => logits, K/V cache rows and greedy ids
must be IDENTICAL BITS.  The persistent launch is opt-in (BZ_PERSIST=1: measured slower than the launches), so it runs in a second process with the
switch set (the library reads its switches once).
Reference anchors: /root/reference/src/engine/cuda_graphs.rs:97-170 (what one graph-mode step computes), executor_generate.rs:357,372."""
import os
import subprocess
import sys

import numpy as np
import pytest

from blazr_amd import _lib as L
from blazr_amd import runtime, synth
from oracle import orc_py

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))

PROBE = r"""
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
from blazr_amd import _lib as L, runtime, synth
model = synth.make_llama("llama3-8b-awq-2l", n_layers=3, max_seq_len=640)
cfg = model["config"]
dev = runtime.Device(0)
lm = runtime.LoadedModel.from_synth(dev, model)
out = {}
for paged in (0, 1):
    p = synth.prompt_tokens(5, cfg["vocab"], seed=31)
    rows = []
    if not paged:
        kv = runtime.LayeredKvCache(dev, cfg["n_layers"], 1, cfg["n_kv_heads"], 320, cfg["max_seq_len"], cfg["head_dim"], L.F16)
    else:
        pk = runtime.LayeredPagedKvCache(dev, cfg["n_layers"], 24, 16, cfg["n_kv_heads"], cfg["head_dim"], L.F16)
        pk.set_blocks([int(b) for b in np.random.default_rng(5).permutation(24)])
    tok = int(p[0])
    for i in range(300):                        # contexts 1 .. 300: one 256-position chunk, then two
        if not paged:
            lg = lm.forward_with_kv_cache([tok], kv, i).to_numpy().reshape(-1)
        else:
            pk.set_seq_len(i + 1)
            lg = lm.forward_with_paged_kv_cache([tok], pk, pk.compute_slot_mapping(i, 1), pk.block_table_device_format(), i + 1, i).to_numpy().reshape(-1)
        if i < 40 or i %% 16 == 0 or i > 250:
            rows.append(lg.copy())
        tok = int(p[i + 1]) if i + 1 < len(p) else int(lg.argmax())
    out["rows%%d" %% paged] = np.stack(rows)
    if not paged:
        out["k"] = kv.read(2, 3, 0, 300); out["v"] = kv.read(1, 5, 1, 300)
ex = runtime.Executor(lm)
out["ids_graph"] = ex.generate(synth.prompt_tokens(7, cfg["vocab"], seed=8), 48, use_graph=True)
out["ids_eager"] = ex.generate(synth.prompt_tokens(7, cfg["vocab"], seed=8), 48)
out["ids_paged_graph"] = ex.generate(synth.prompt_tokens(7, cfg["vocab"], seed=8), 48, paged=True, use_graph=True)
np.savez(sys.argv[1], **out)
dev.close()
"""


def _run(tmp_path, name, env_extra):
    e = dict(os.environ)
    e.update(env_extra)
    out = tmp_path / (name + ".npz")
    src = PROBE % (os.path.dirname(HERE), HERE)
    r = subprocess.run([sys.executable, "-c", src, str(out)], env=e, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (name, r.stdout[-3000:], r.stderr[-3000:])
    return np.load(out)


@pytest.mark.watchdog(1000)
def test_persistent_step_is_bit_identical_to_the_launch_per_phase_step(tmp_path):
    a = _run(tmp_path, "persist", {"BZ_PERSIST": "1"})
    b = _run(tmp_path, "phases", {})
    for k in ("rows0", "rows1", "k", "v", "ids_graph", "ids_eager", "ids_paged_graph"):
        assert np.array_equal(a[k], b[k]), (k, int((a[k] != b[k]).sum()))
    assert np.array_equal(a["rows0"], a["rows1"])                    # paged == contiguous, bit for bit
    assert a["ids_graph"].tolist() == a["ids_eager"].tolist() == a["ids_paged_graph"].tolist()


def test_decode_step_matches_the_oracle_to_1e_4(device):
    """the default decode path against the CPU oracle at the real layer widths (3 layers): every sub-op of the int4 path is bit-identical to the oracle's
    (scripts/parity_depth.py: zero differing elements per layer), what remains is the dense lm_head's f32 summation order: 1e-4, every step"""
    model = synth.make_llama("llama3-8b-awq-2l", n_layers=3)
    cfg = model["config"]
    lm, om = runtime.LoadedModel.from_synth(device, model), orc_py.OrcLlama(model)
    kv = runtime.LayeredKvCache(device, cfg["n_layers"], 1, cfg["n_kv_heads"], 40, cfg["max_seq_len"], cfg["head_dim"], L.F16)
    okv = om.new_kv(40)
    tok, worst = 11, 0.0
    for i in range(24):
        lo = np.asarray(om.forward_kv([tok], okv, i)).reshape(-1)
        lg = lm.forward_with_kv_cache([tok], kv, i).to_numpy().reshape(-1)
        err = float(np.linalg.norm(lg.astype(np.float64) - lo) / np.linalg.norm(lo))
        worst = max(worst, err)
        assert err <= 1e-4, (i, err)
        tok = int(lo.argmax())
    print("decode step vs oracle, 3 layers at 8B widths, 24 steps: worst relative L2 %.3e" % worst)
    orc_py.lib().orc_kv_free(okv)
