#!/usr/bin/env python3
"""Mamba2 prompt prefill timing: Mamba2-2.7B shape, prompt of S tokens through bz_forward_ssm.  S >= 8 takes the batched path (MFMA GEMMs +
in-kernel scan over the tokens); BZ_NO_MFMA_PREFILL=1 forces the recurrence token by token for comparison."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blazr_amd import runtime, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--preset", default="mamba2-2.7b")
ap.add_argument("--lens", default="64,512,2048")
args = ap.parse_args()
cfg = synth.make_mamba_config(args.preset)
dev = runtime.Device(0)
lm = runtime.LoadedModel.from_synth(dev, synth.make_mamba2(args.preset))
out = []
for S in [int(x) for x in args.lens.split(",")]:
    p = synth.prompt_tokens(S, cfg["vocab"])
    best = None
    for rep in range(3):
        st = runtime.LayeredSsmState(lm)
        dev.synchronize()
        t0 = time.perf_counter()
        lm.forward_with_ssm_state(p, st)
        dev.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    out.append({"prompt_tokens": S, "ms": round(best * 1e3, 3), "prefill_tok_s": round(S / best, 1)})
print(json.dumps({"preset": args.preset, "batched": not os.environ.get("BZ_NO_MFMA_PREFILL"), "results": out}))
