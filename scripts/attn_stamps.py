import os, sys
sys.path.insert(0, ".")
os.environ["BZ_ATTN_STAMPS"] = "1"
os.environ["BZ_ATTN_STAMPS_PRINT"] = "1"
os.environ["BZ_MLP_STAMPS"] = "1"
from blazr_amd import runtime, synth, _lib as L
cfg = synth.make_config("llama3-8b-awq-2l")
dev = runtime.Device(0)
m = synth.make_llama("llama3-8b-awq-2l")
lm = runtime.LoadedModel.from_synth(dev, m)
kv = runtime.LayeredKvCache(dev, 2, 1, 8, 256, cfg["max_seq_len"], 128, L.F16)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 150
p = synth.prompt_tokens(N, cfg["vocab"])
lm.forward_with_kv_cache(p, kv, 0)
for i in range(6):
    lm.forward_with_kv_cache([5], kv, N + i)
dev.synchronize()
