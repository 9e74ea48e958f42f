import os, sys
sys.path.insert(0, ".")
import numpy as np
from blazr_amd import runtime, synth, _lib as L
from oracle import orc_py
dev = runtime.Device(0)
model = synth.make_llama("llama3-8b-awq-2l")
cfg = model["config"]
lm = runtime.LoadedModel.from_synth(dev, model)
om = orc_py.OrcLlama(model)
n = 150
p = synth.prompt_tokens(n, cfg["vocab"], seed=5)
okv = om.new_kv(256)
want = om.forward_kv(p, okv, 0, all_logits=True)
for mn in (100000, 4):
    os.environ["BZ_SPLIT_MIN"] = str(mn)
    kv = runtime.LayeredKvCache(dev, cfg["n_layers"], 1, cfg["n_kv_heads"], 8, cfg["max_seq_len"], cfg["head_dim"], L.F16)
    errs = []
    for i in range(n):
        lg = lm.forward_with_kv_cache([int(p[i])], kv, i).to_numpy()[0].astype(np.float64)
        w = want[i].astype(np.float64)
        errs.append(np.linalg.norm(lg - w) / np.linalg.norm(w))
    errs = np.array(errs)
    print("split_min", mn, "max %.3e mean %.3e" % (errs.max(), errs.mean()), "at", int(errs.argmax()), ["%.2e" % errs[i] for i in (2, 5, 64, 127, 128, 129, 149)])
