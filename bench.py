#!/usr/bin/env python3
"""bench.py -- decode tokens/sec + HBM-roofline fraction, Llama-3-8B AWQ-INT4 (gs=128), seq=1 greedy decode.

Contract (driver):  python bench.py --gpus N --steps K --warmup W   prints ONE JSON line on rank 0.
  * a "step" is one pass of the hot path = one decoded token (one hipGraph replay of the whole decode step).
  * N > 1 (launched by torch.distributed.run, one rank per GPU): the decode loop does not shard (SURVEY.md 8e,
    "replicas only"): every rank runs an independent replica on its own GPU, no data-path collective;
    value = tokens decoded by all ranks / max-over-ranks time ("scaling": "weak").
  * weights are seeded synthetic at the reference shapes (no network, no checkpoints): SURVEY.md 8d.
  * timed region: inputs resident in HBM (model loaded, prompt prefilled, graph captured) before the first event.
Extra objects: "roofline" (dominant kernel, pure dispatch time from HIP events on the launch stream vs 8 TB/s) and
"cpu_baseline" (the CPU oracle -- a port, the reference's own --cpu path cannot be built here: BASELINE.md 4).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling


def hip_events(stream):
    """Minimal HIP event timer on a given stream (torch.cuda.Event only sees torch's current stream)."""
    hip = C.CDLL("libamdhip64.so")
    hip.hipEventCreate.argtypes = [C.POINTER(C.c_void_p)]
    hip.hipEventRecord.argtypes = [C.c_void_p, C.c_void_p]
    hip.hipEventSynchronize.argtypes = [C.c_void_p]
    hip.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), C.c_void_p, C.c_void_p]

    class T:
        def __init__(self):
            self.a, self.b = C.c_void_p(), C.c_void_p()
            assert hip.hipEventCreate(C.byref(self.a)) == 0 and hip.hipEventCreate(C.byref(self.b)) == 0

        def start(self):
            assert hip.hipEventRecord(self.a, stream) == 0

        def stop(self):
            assert hip.hipEventRecord(self.b, stream) == 0

        def ms(self):
            assert hip.hipEventSynchronize(self.b) == 0
            out = C.c_float()
            assert hip.hipEventElapsedTime(C.byref(out), self.a, self.b) == 0
            return float(out.value)
    return T()


def pmc_traffic(label):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes of this build (profiles/r01_pmc_traffic.json,
    made by scripts/pmc_traffic.py: separate FETCH_SIZE / WRITE_SIZE passes, gfx950 correction (2*FETCH + WRITE) * 1024).  A process cannot
    collect PMC counters on itself, so this is read back from the file; None when the file or the kernel is missing."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
        for lbl, sym in d["labels"].items():
            if label.startswith(lbl):
                for name, v in d["kernels"].items():
                    if name.startswith(sym):
                        return v["hbm_bytes_per_launch"]
    except (OSError, ValueError, KeyError):
        pass
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=128)      # /root/reference/src/cli/bench.rs:27 decode tokens per run
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--preset", default="llama3-8b-awq")
    ap.add_argument("--prompt-len", type=int, default=16)   # 16 fixed prompt ids, context <= 144 + warmup
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-tokens", type=int, default=32)    # ~10 s of CPU work on the GPU box's host cores (bounded sample)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    n_gpus = max(world, 1)

    from blazr_amd import runtime, synth

    if args.preset in synth.MAMBA_PRESETS or args.preset in synth.DSV2_PRESETS:
        return bench_aux(args, rank, local_rank, world, dist)
    cfg = synth.make_config(args.preset)
    need = args.prompt_len + args.warmup + args.steps + 8
    if need > cfg["max_seq_len"]:
        cfg["max_seq_len"] = need
    dev = runtime.Device(local_rank)
    t0 = time.time()
    want_cpu = (rank == 0 and world == 1 and not args.no_cpu_baseline)
    lm = runtime.LoadedModel(dev, cfg)
    host_layers = []
    for i in range(cfg["n_layers"]):
        lay = synth.llama_layer(cfg, i)
        lm.add_llama_layer(i, lay)
        if want_cpu:
            host_layers.append(lay)
    emb, fnorm, lmh = synth.llama_head(cfg)
    lm.add_llama_head(emb, fnorm, lmh)
    lm.finalize()
    load_s = time.time() - t0
    resident, per_token = lm.weight_bytes()
    algo_bytes = synth.algorithmic_bytes_per_token(cfg)
    assert per_token == algo_bytes, (per_token, algo_bytes)

    from blazr_amd import _lib as L
    prompt = synth.prompt_tokens(args.prompt_len, cfg["vocab"])
    kv_dt = {"f16": L.F16, "bf16": L.BF16, "f32": L.F32}[cfg["act_dtype"]]
    kv = runtime.LayeredKvCache(dev, cfg["n_layers"], 1, cfg["n_kv_heads"], need, cfg["max_seq_len"], cfg["head_dim"], kv_dt)
    logits = lm.forward_with_kv_cache(prompt, kv, 0)                       # prefill (untimed)
    first = int(runtime.logits_to_token(dev, logits, [], []).to_numpy()[0])  # argmax_on_gpu (cuda_graphs.rs:149-163)
    graph = runtime.DecodeGraph(lm, kv)
    graph.seed_next_token(first, args.prompt_len)
    for _ in range(args.warmup):
        graph.replay()
    dev.synchronize()

    timer = hip_events(C.c_void_p(dev.stream()))
    if dist is not None:
        dist.barrier()
    dev.synchronize()
    t_host0 = time.perf_counter()
    timer.start()
    for _ in range(args.steps):
        graph.replay()
    timer.stop()
    gpu_ms = timer.ms()
    dev.synchronize()
    host_ms = (time.perf_counter() - t_host0) * 1e3
    from blazr_amd import replicas
    if dist is not None:
        dist.barrier()
    tok_s, wall_ms = replicas.aggregate_tokens_per_s(max(gpu_ms, host_ms), args.steps, dist, "cuda" if dist is not None else None)
    tokens = [first] + [graph.read_token(i) for i in range(args.warmup + args.steps)]

    # per-kernel dispatch times of real decode steps (pure kernel time, hipExtLaunchKernelGGL start/stop events)
    pos = args.prompt_len + args.warmup + args.steps
    prof = lm.profile_step(kv, tokens[-1], min(pos, cfg["max_seq_len"] - 5), iters=4)
    for p in prof:
        p["avg_us"] = 1e3 * p["total_ms"] / max(p["launches"], 1)
        p["gbs"] = (p["algo_bytes"] / 1e9) / (p["total_ms"] / 1e3) if p["total_ms"] > 0 and p["algo_bytes"] > 0 else None
    dom = max(prof, key=lambda p: p["total_ms"])
    roof = {"bound": "hbm", "kernel": dom["name"], "achieved": round(dom["gbs"], 1) if dom["gbs"] else None, "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(dom["gbs"] / HBM_PEAK_GBS, 4) if dom["gbs"] else None, "traffic": pmc_traffic(dom["name"]),
            "bytes_per_launch": dom["algo_bytes"] / max(dom["launches"], 1), "avg_launch_us": round(dom["avg_us"], 2),
            "whole_step_frac": round(algo_bytes * (tok_s / n_gpus) / (HBM_PEAK_GBS * 1e9), 4)}

    out = {"metric": "decode tokens/sec, Llama-3-8B AWQ-INT4 seq=1 on MI355X (and HBM-roofline fraction)", "value": round(tok_s, 2),
           "unit": "tokens/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(wall_ms / args.steps, 5),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "i8", "data": "synthetic",
           "config": {"workload": "%s greedy decode, batch 1, seq=1, prompt %d, whole step as one hipGraph; N>1 = independent replicas"
                                  % (args.preset, args.prompt_len), "algorithmic_bytes_per_token": algo_bytes,
                      "resident_weight_bytes": resident, "context_at_end": pos},
           "roofline": roof,
           "kernels": [{k: (round(v, 3) if isinstance(v, float) else v) for k, v in p.items()} for p in prof],
           "gpu_ms_events": round(gpu_ms, 3), "host_ms": round(host_ms, 3), "load_s": round(load_s, 1)}

    if want_cpu:
        # CPU baseline: the oracle (a port -- BASELINE.md 4), same weights, same prompt, bounded sample; and the parity gate
        from oracle import orc_py
        model = dict(config=cfg, embed=emb, final_norm=fnorm, lm_head=lmh, layers=host_layers)
        om = orc_py.OrcLlama(model)
        n_cpu = max(2, args.cpu_tokens)
        okv = om.new_kv(args.prompt_len + n_cpu + 1)
        tp0 = time.perf_counter()
        lo = om.forward_kv(prompt, okv, 0)
        t_prefill = time.perf_counter() - tp0
        def near_tie(row):      # the same guard as the tests: ids are only comparable while the oracle's top-2 gap is not a rounding-level tie
            srt = np.sort(row)
            return bool(srt[-1] - srt[-2] < 4e-3 * np.abs(row).max())
        cpu_tokens = [int(lo[0].argmax())]
        fair = 0 if near_tie(lo[0]) else 1
        tied = fair == 0
        td0 = time.perf_counter()
        for i in range(n_cpu - 1):
            lo = om.forward_kv([cpu_tokens[-1]], okv, args.prompt_len + i)
            cpu_tokens.append(int(lo[0].argmax()))
            if not tied:
                tied = near_tie(lo[0])
                fair += 0 if tied else 1
        t_decode = time.perf_counter() - td0
        orc_py.lib().orc_kv_free(okv)
        cpu_tok_s = (n_cpu - 1) / t_decode          # bench.rs:299-306: (tokens - 1) / (total - TTFT)
        out["cpu_baseline"] = {"value": round(cpu_tok_s, 3), "unit": "tokens/s", "cores": orc_py.lib().orc_num_threads(), "kind": "port",
                               "sample": "oracle/liborc.so (C + OpenMP), same synthetic weights and prompt: %d-token prefill (%.1f s) + %d greedy "
                                         "decode tokens (%.1f s); host has %d logical CPUs" % (args.prompt_len, t_prefill, n_cpu - 1, t_decode, os.cpu_count())}
        n_cmp = max(fair, 1)
        n_same = next((i for i, (a, g) in enumerate(zip(cpu_tokens, tokens)) if a != g), min(len(cpu_tokens), len(tokens)))
        out["parity"] = {"greedy_ids_match": cpu_tokens[:n_cmp] == tokens[:n_cmp], "n_compared": n_cmp, "n_identical_prefix": n_same, "n_tokens": n_cpu,
                         "note": "ids compared up to the first step whose oracle top-2 gap is a rounding-level tie", "cpu": cpu_tokens[:16], "gpu": tokens[:16]}

    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def bench_aux(args, rank, local_rank, world, dist):
    """Secondary workloads (BASELINE.json configs[3], [4]): Mamba2-2.7B (state + weights streamed once per token) and
    DeepSeek-V2-Lite (MLA latent cache + MoE: only the routed + shared experts' weights are streamed)."""
    from blazr_amd import replicas, runtime, synth
    mamba = args.preset in synth.MAMBA_PRESETS
    cfg = synth.make_mamba_config(args.preset) if mamba else synth.make_dsv2_config(args.preset)
    need = args.prompt_len + args.warmup + args.steps + 8
    if not mamba and need > cfg["max_seq_len"]:
        cfg["max_seq_len"] = need
    dev = runtime.Device(local_rank)
    t0 = time.time()
    lm = runtime.LoadedModel.from_synth_streamed(dev, cfg)
    load_s = time.time() - t0
    w_bytes, s_bytes = synth.mamba2_bytes_per_token(cfg) if mamba else (synth.dsv2_bytes_per_token(cfg), 0)
    resident, per_token = lm.weight_bytes()
    assert per_token == w_bytes, (per_token, w_bytes)
    prompt = synth.prompt_tokens(args.prompt_len, cfg["vocab"])
    if mamba:
        st = runtime.LayeredSsmState(lm)
        logits = lm.forward_with_ssm_state(prompt, st)
    else:
        st = lm.new_kv_cache(need)
        logits = lm.forward_with_kv_cache(prompt, st, 0)
    first = int(runtime.logits_to_token(dev, logits, [], []).to_numpy()[0])
    graph = runtime.DecodeGraph(lm, st)
    graph.seed_next_token(first, args.prompt_len)
    for _ in range(args.warmup):
        graph.replay()
    dev.synchronize()
    timer = hip_events(C.c_void_p(dev.stream()))
    if dist is not None:
        dist.barrier()
    dev.synchronize()
    t_host0 = time.perf_counter()
    timer.start()
    for _ in range(args.steps):
        graph.replay()
    timer.stop()
    gpu_ms = timer.ms()
    dev.synchronize()
    host_ms = (time.perf_counter() - t_host0) * 1e3
    if dist is not None:
        dist.barrier()
    n_gpus = max(world, 1)
    tok_s, wall_ms = replicas.aggregate_tokens_per_s(max(gpu_ms, host_ms), args.steps, dist, "cuda" if dist is not None else None)
    tokens = [first] + [graph.read_token(i) for i in range(args.warmup + args.steps)]
    prof = lm.profile_step_ssm(st, tokens[-1], iters=4) if mamba else lm.profile_step(st, tokens[-1], args.prompt_len + args.warmup + args.steps, iters=4)
    for p in prof:
        p["avg_us"] = 1e3 * p["total_ms"] / max(p["launches"], 1)
        p["gbs"] = (p["algo_bytes"] / 1e9) / (p["total_ms"] / 1e3) if p["total_ms"] > 0 and p["algo_bytes"] > 0 else None
    dom = max(prof, key=lambda p: p["total_ms"])
    algo = w_bytes + s_bytes
    out = {"metric": "decode tokens/sec, %s seq=1 on MI355X (and HBM-roofline fraction)" % args.preset, "value": round(tok_s, 2), "unit": "tokens/s",
           "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(wall_ms / args.steps, 5), "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": cfg["act_dtype"], "data": "synthetic",
           "config": {"workload": "%s greedy decode, batch 1, prompt %d, whole step as one hipGraph" % (args.preset, args.prompt_len),
                      "algorithmic_bytes_per_token": algo, "weight_bytes": w_bytes, "state_bytes_rw": s_bytes, "resident_weight_bytes": resident},
           "roofline": {"bound": "hbm", "kernel": dom["name"], "achieved": round(dom["gbs"], 1) if dom["gbs"] else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(dom["gbs"] / HBM_PEAK_GBS, 4) if dom["gbs"] else None, "traffic": None,
                        "whole_step_frac": round(algo * (tok_s / n_gpus) / (HBM_PEAK_GBS * 1e9), 4)},
           "kernels": [{k: (round(v, 3) if isinstance(v, float) else v) for k, v in p.items()} for p in prof],
           "gpu_ms_events": round(gpu_ms, 3), "host_ms": round(host_ms, 3), "load_s": round(load_s, 1), "tokens_head": tokens[:8]}
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
