"""Generates tests/golden/*.json + *.npz from the oracle (oracle/liborc.so) on seeded synthetic checkpoints.

The reference cannot be run here (no Rust toolchain, boostr/numr absent) and ships no vectors for this path, so these
fixtures pin OUR oracle against regressions; they are inputs + expected outputs only (no reference source).
Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from blazr_amd import synth  # noqa: E402
from oracle import orc_py  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = [("tiny-awq", {}), ("tiny-gptq", dict(act_order=True, bias=True)), ("tiny-bf16", {}), ("tiny-q4km", {}), ("tiny-q8_0", {}),
         ("tiny-mamba2", {}), ("tiny-mamba2-g2", {}), ("tiny-dsv2", {}), ("tiny-dsv2-f32", {})]

def build(preset, over):
    """(model dict, oracle model) for a fixture's preset -- shared with the tests that replay the fixtures"""
    if preset in synth.MAMBA_PRESETS:
        m = synth.make_mamba2(preset, **over)
        return m, orc_py.OrcMamba2(m)
    if preset in synth.DSV2_PRESETS:
        m = synth.make_dsv2(preset, **over)
        return m, orc_py.OrcDsv2(m)
    m = synth.make_llama(preset, **over)
    return m, orc_py.OrcLlama(m)


def first_linear(m):
    lay = m["layers"][0]
    return lay.get("q") or lay.get("in_proj") or lay["q_proj"]


for preset, over in CASES if __name__ == "__main__" else []:
    m, om = build(preset, over)
    prompt = synth.prompt_tokens(8, m["config"]["vocab"], seed=3)
    toks, trace = om.generate(prompt, 12, trace=True)
    idx = np.argsort(trace[-1])[-16:]
    dq = orc_py.OrcLinear(first_linear(m)).dequant()
    rows, cols = np.arange(0, dq.shape[0], 37), np.arange(0, dq.shape[1], 29)
    name = preset + ("-actorder" if over else "")
    json.dump(dict(preset=preset, over=over, prompt=prompt.tolist(), max_tokens=12, tokens=toks.tolist()),
              open(os.path.join(HERE, name + ".json"), "w"))
    np.savez(os.path.join(HERE, name + ".npz"), logit_idx=idx, logit_val=trace[-1][idx], dq_rows=rows, dq_cols=cols,
             dq_val=dq[rows][:, cols])
    print(name, toks.tolist())
