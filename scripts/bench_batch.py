#!/usr/bin/env python3
"""Batched paged decode (process_decode_batch, batch_decode.rs:35-150) on the Llama-3-8B AWQ shape: aggregate tokens/s for N sequences decoded
together through bz_forward_paged_batch (weights shared across the batch, one pass per 8 rows) against N times the single-stream step
(BZ_NO_BATCH_SHARING=1); --graph: the same step captured once as a hipGraph (bz_decode_batch_graph_*, cuda_graphs_batched.rs) and replayed with the
tokens, positions and slots resident on the device."""
import argparse
import json
import os
import sys
import time

import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blazr_amd import _lib as L   # noqa: E402
from blazr_amd import runtime, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--preset", default="llama3-8b-awq")
ap.add_argument("--batches", default="1,2,4,8,16,32")
ap.add_argument("--steps", type=int, default=16)
ap.add_argument("--graph", action="store_true")
args = ap.parse_args()
cfg = synth.make_config(args.preset)
dev = runtime.Device(0)
lm = runtime.LoadedModel.from_synth_streamed(dev, cfg)
bs, per = 16, 4
out = []
for N in [int(x) for x in args.batches.split(",")]:
    pool = runtime.LayeredPagedKvCache(dev, cfg["n_layers"], N * per, bs, cfg["n_kv_heads"], cfg["head_dim"], L.F16)
    tables = [[i + N * j for j in range(per)] for i in range(N)]
    lens = [8 + i % 5 for i in range(N)]
    for i in range(N):     # prompts
        p = synth.prompt_tokens(lens[i], cfg["vocab"], seed=i)
        lm.forward_with_paged_kv_cache(p, pool, [tables[i][k // bs] * bs + k % bs for k in range(lens[i])], tables[i], lens[i], 0)
    toks = [1 + i for i in range(N)]
    if args.graph and N >= 2:
        g = runtime.BatchDecodeGraph(lm, pool, N, per)
        g.seed(toks, [n + 1 for n in lens], tables)
        g.replay(); dev.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            g.replay()
        dev.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        out.append({"sequences": N, "ms_per_step": round(dt * 1e3, 3), "aggregate_tok_s": round(N / dt, 1), "graph": True})
        del g
        continue
    def step():
        global lens, toks
        lens = [n + 1 for n in lens]
        slots = [tb[(n - 1) // bs] * bs + (n - 1) % bs for n, tb in zip(lens, tables)]
        lg = lm.forward_paged_batch(toks, pool, slots, [tb[:(n + bs - 1) // bs] for n, tb in zip(lens, tables)], lens)
        return lg
    step(); dev.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    dev.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    out.append({"sequences": N, "ms_per_step": round(dt * 1e3, 3), "aggregate_tok_s": round(N / dt, 1)})
print(json.dumps({"preset": args.preset, "shared_weights": not os.environ.get("BZ_NO_BATCH_SHARING"), "results": out}))
