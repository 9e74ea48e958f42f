"""Phase timeline of the q/k/v int4 GEMV (diagnostic): s_memrealtime stamps of the first and the last block, printed by the library."""
import os, sys
sys.path.insert(0, ".")
os.environ["BZ_QKV_STAMPS"] = "1"
from blazr_amd import runtime, synth, _lib as L
cfg = synth.make_config("llama3-8b-awq-2l", n_layers=4)
dev = runtime.Device(0)
lm = runtime.LoadedModel.from_synth_streamed(dev, cfg)
kv = runtime.LayeredKvCache(dev, 4, 1, 8, 64, cfg["max_seq_len"], 128, L.F16)
p = synth.prompt_tokens(4, cfg["vocab"])
lm.forward_with_kv_cache(p, kv, 0)
for i in range(8):
    lm.forward_with_kv_cache([5], kv, 4 + i)
dev.synchronize()
