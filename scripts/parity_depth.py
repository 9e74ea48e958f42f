"""Where does the GPU-vs-oracle logit difference of a deep f16 model come from?  Runs an N-layer Llama-3-8B-AWQ-shaped model one layer at a
time on both sides (pieces API) and prints, per layer: the accumulated difference of the residual stream (both sides free-running) and the
LOCAL difference (the GPU layer fed with the oracle's inputs), as relative L2 and as the number of elements that differ at all.
usage: python scripts/parity_depth.py [n_layers=12]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blazr_amd import _lib as L, runtime, synth  # noqa: E402
from oracle import orc_py  # noqa: E402

nl = int(sys.argv[1]) if len(sys.argv) > 1 else 12
model = synth.make_llama("llama3-8b-awq-2l", n_layers=nl)
cfg = model["config"]
dev = runtime.Device(0)
lm, om = runtime.LoadedModel.from_synth(dev, model), orc_py.OrcLlama(model)
H = cfg["hidden"]


def rel(a, b):
    return float(np.linalg.norm(a.astype(np.float64) - b) / max(np.linalg.norm(b), 1e-30))


for tok, pos in ((17, 0), (4242, 1), (99, 2)):
    kv_a = runtime.LayeredKvCache(dev, nl, 1, cfg["n_kv_heads"], 8, cfg["max_seq_len"], cfg["head_dim"], L.F16) if pos == 0 else kv_a
    kv_l = runtime.LayeredKvCache(dev, nl, 1, cfg["n_kv_heads"], 8, cfg["max_seq_len"], cfg["head_dim"], L.F16) if pos == 0 else kv_l
    okv = om.new_kv(8) if pos == 0 else okv
    oh, opm = om.embed([tok]), None
    gh, gpm = lm.forward_embed([tok]), None
    print("token %d at position %d" % (tok, pos))
    for l in range(nl):
        # local: GPU layer l on the oracle's inputs (kv_l holds GPU-computed K/V of oracle-fed layers)
        if os.environ.get("PRE_ADD"):      # the deferred residual added on the host: the layer starts without one (the persistent launch takes such steps)
            lh = dev.tensor((oh if opm is None else (oh + opm).astype(np.float16)).astype(np.float32))
            lpm = None
        else:
            lh = dev.tensor(oh.astype(np.float32))
            lpm = None if opm is None else dev.tensor(opm.astype(np.float32))
        lh, lpm = lm.forward_layers_range(lh, lpm, kv_l, l, l + 1, pos)
        gh, gpm = lm.forward_layers_range(gh, gpm, kv_a, l, l + 1, pos)
        oh, opm = om.layers_range(oh, opm, okv, l, l + 1, pos)
        lo, go, oo = lh.to_numpy() + lpm.to_numpy(), gh.to_numpy() + gpm.to_numpy(), oh + opm
        print("  layer %2d: h + mlp   accumulated rel L2 %.2e (%4d of %d differ)   local rel L2 %.2e (%4d differ; mlp out alone %4d)"
              % (l, rel(go, oo), int((go != oo).sum()), H, rel(lo, oo), int((lo != oo).sum()), int((lpm.to_numpy() != opm).sum())))
dev.close()
