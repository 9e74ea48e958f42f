"""Phase stamps of the DeepSeek route + gate/up launch (BZ_MOE_STAMPS=1): a few eager decode steps of the full-width 4-layer model."""
import os, sys
os.environ["BZ_MOE_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blazr_amd import runtime, synth
model = synth.make_dsv2("deepseek-v2-lite", n_layers=4, vocab=4096, max_seq_len=256)
dev = runtime.Device(0)
lm = runtime.LoadedModel.from_synth(dev, model)
kv = lm.new_kv_cache(64)
for i in range(4):
    lm.forward_with_kv_cache([5 + i], kv, i)
dev.close()
