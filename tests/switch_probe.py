"""One process per run-time switch (the library reads BZ_* once): run a case on the GPU and through the oracle, write both to an .npz.

usage: python tests/switch_probe.py <case> <out.npz>      (case: a key of fullwidth_cases.CASES or a tiny synth preset)
Called by tests/test_gpu_fullwidth.py::test_switch_paths_match_the_oracle with the switch in the environment; the parent asserts.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

from blazr_amd import runtime, synth  # noqa: E402
import fullwidth_cases as fc  # noqa: E402


def main():
    name, out = sys.argv[1], sys.argv[2]
    if name in fc.CASES:
        fam, model = fc.make(name)
    elif name in synth.MAMBA_PRESETS:
        fam, model = "mamba2", synth.make_mamba2(name)
    elif name in synth.DSV2_PRESETS:
        fam, model = "dsv2", synth.make_dsv2(name)
    else:
        fam, model = "llama", synth.make_llama(name)
    cfg = model["config"]
    dev = runtime.Device(0)
    g, o = fc.GpuRun(dev, fam, model), fc.OrcRun(fam, model, cap=96)
    prompt = synth.prompt_tokens(6, cfg["vocab"], seed=2)
    want = o.forward(prompt, all_logits=True)
    got = g.forward(prompt, all_logits=True)
    ids = []
    tok = int(want[-1].argmax())
    for _ in range(10):                       # teacher forced with the oracle's ids: every row is comparable
        ids.append(tok)
        lo, lg = o.forward([tok]), g.forward([tok])
        want, got = np.concatenate([want, lo.reshape(1, -1)]), np.concatenate([got, lg.reshape(1, -1)])
        tok = int(lo.reshape(-1).argmax())
    p2 = synth.prompt_tokens(20, cfg["vocab"], seed=9)       # a second chunk: the batched prefill path, where the family has one
    want, got = np.concatenate([want, o.forward(p2, all_logits=True)]), np.concatenate([got, g.forward(p2, all_logits=True)])
    # free generation, ids compared on the fair prefix
    om = fc.make_oracle(fam, model)
    gp = synth.prompt_tokens(10, cfg["vocab"], seed=5)
    ids_want, trace = om.generate(gp, 16, trace=True)
    srt = np.sort(trace, axis=1)
    bad = np.nonzero((srt[:, -1] - srt[:, -2]) < 4e-3 * np.abs(trace).max(axis=1))[0]
    fair = int(bad[0]) if len(bad) else len(trace)
    ids_got = runtime.Executor(g.lm).generate(gp, 16, use_graph=True)
    np.savez(out, got=got, want=want, act=cfg["act_dtype"], ids_got=np.asarray(ids_got), ids_want=np.asarray(ids_want), fair=fair)
    dev.close()


if __name__ == "__main__":
    main()
