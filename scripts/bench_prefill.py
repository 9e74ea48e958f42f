#!/usr/bin/env python3
"""Prefill timing (TTFT side of cli/bench.rs): dense bf16 Llama-3.2-1B shape, prompt of S tokens through bz_forward_kv.
S >= 8 takes the batched MFMA path (bz_prefill.hip); BZ_NO_MFMA_PREFILL=1 forces the token-by-token GEMV path for comparison."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blazr_amd import runtime, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--preset", default="llama3.2-1b-bf16")
ap.add_argument("--lens", default="64,512,2048")
args = ap.parse_args()
cfg = synth.make_config(args.preset)
dev = runtime.Device(0)
lm = runtime.LoadedModel.from_synth_streamed(dev, cfg)
H, I, V, L = cfg["hidden"], cfg["inter"], cfg["vocab"], cfg["n_layers"]
qn = (cfg["n_heads"] + 2 * cfg["n_kv_heads"]) * cfg["head_dim"]
params = L * (qn * H + H * cfg["n_heads"] * cfg["head_dim"] + 3 * H * I)
out = []
for S in [int(x) for x in args.lens.split(",")]:
    p = synth.prompt_tokens(S, V)
    best = None
    for rep in range(3):
        kv = lm.new_kv_cache(S + 8, max(S + 8, cfg["max_seq_len"]))
        dev.synchronize()
        t0 = time.perf_counter()
        lm.forward_with_kv_cache(p, kv, 0)
        dev.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    out.append({"prompt_tokens": S, "ms": round(best * 1e3, 3), "prefill_tok_s": round(S / best, 1), "gemm_tflops": round(2.0 * params * S / best / 1e12, 2)})
print(json.dumps({"preset": args.preset, "mfma": not os.environ.get("BZ_NO_MFMA_PREFILL"), "results": out}))
