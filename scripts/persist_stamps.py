"""Per-wave phase stamps of the persistent decode launch (BZ_PERSIST_STAMPS=1): a few eager decode steps of the Llama-3-8B AWQ shape (n layers),
the library prints the stamps of layer 1 to stderr.  usage: BZ_PERSIST_STAMPS=1 python scripts/persist_stamps.py [n_layers=4]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blazr_amd import _lib as L, runtime, synth  # noqa: E402

nl = int(sys.argv[1]) if len(sys.argv) > 1 else 4
model = synth.make_llama("llama3-8b-awq-2l", n_layers=nl)
cfg = model["config"]
dev = runtime.Device(0)
lm = runtime.LoadedModel.from_synth(dev, model)
kv = runtime.LayeredKvCache(dev, nl, 1, cfg["n_kv_heads"], 64, cfg["max_seq_len"], cfg["head_dim"], L.F16)
tok = 5
for i in range(40):
    lg = lm.forward_with_kv_cache([tok], kv, i).to_numpy().reshape(-1)
    tok = int(lg.argmax())
dev.close()
