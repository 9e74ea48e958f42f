"""One prompt prefill of a preset (for rocprofv3 --kernel-trace --stats): python3 scripts/prof_prefill.py <preset> <prompt_len> [reps] [n_layers]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blazr_amd import runtime, synth
preset, n, reps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 2
dev = runtime.Device(0)
if preset in synth.DSV2_PRESETS:
    cfg = synth.make_dsv2_config(preset)
elif preset in synth.MAMBA_PRESETS:
    cfg = synth.make_mamba_config(preset)
else:
    cfg = synth.make_config(preset)
cfg["max_seq_len"] = max(cfg.get("max_seq_len", 0), n + 16)
if len(sys.argv) > 4:
    cfg["n_layers"] = int(sys.argv[4])
lm = runtime.LoadedModel.from_synth_streamed(dev, cfg)
p = synth.prompt_tokens(n, cfg["vocab"])
for r in range(reps):
    st = runtime.LayeredSsmState(lm) if preset in synth.MAMBA_PRESETS else lm.new_kv_cache(n + 16)
    dev.synchronize(); t = time.time()
    if preset in synth.MAMBA_PRESETS:
        lm.forward_with_ssm_state(p, st)
    else:
        lm.forward_with_kv_cache(p, st, 0)
    dev.synchronize()
    print("prefill %d tokens: %.2f ms" % (n, (time.time() - t) * 1e3), flush=True)
dev.close()
