"""How far is the batched (MFMA) prompt path from the token-by-token decode path, and how does that grow with depth?
For n_layers in a list: the same synthetic Llama-3-8B-AWQ-shaped model, one prompt; logits of every prompt row from (a) ONE forward call over
the whole prompt (the batched prefill when it is eligible) and (b) one call per token (the decode kernels, which are the oracle's bits:
scripts/parity_depth.py).  Also compares the K / V rows both paths wrote.
usage: python scripts/parity_prefill.py [prompt_len=16] [layers=1,2,4,8] [preset=llama3-8b-awq-2l] [vocab]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blazr_amd import _lib as L, runtime, synth  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 16
layers = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "1,2,4,8").split(",")]
preset = sys.argv[3] if len(sys.argv) > 3 else "llama3-8b-awq-2l"
dev = runtime.Device(0)
rng = np.random.default_rng(5)


def rel(a, b):
    return float(np.linalg.norm(a.astype(np.float64) - b) / max(np.linalg.norm(b), 1e-30))


for nl in layers:
    model = synth.make_llama(preset, n_layers=nl, **({"vocab": int(sys.argv[4])} if len(sys.argv) > 4 else {}))
    cfg = model["config"]
    lm = runtime.LoadedModel.from_synth(dev, model)
    toks = [int(t) for t in rng.integers(0, cfg["vocab"], S)]
    kvdt = {"f16": L.F16, "bf16": L.BF16, "f32": L.F32}[cfg.get("act_dtype", "f16")]
    mk = lambda: runtime.LayeredKvCache(dev, nl, 1, cfg["n_kv_heads"], max(S, 8), cfg["max_seq_len"], cfg["head_dim"], kvdt)
    kv_a, kv_b = mk(), mk()
    a = lm.forward_with_kv_cache(toks, kv_a, 0, all_logits=True).to_numpy().reshape(S, -1)
    b = np.stack([lm.forward_with_kv_cache([t], kv_b, i).to_numpy().reshape(-1) for i, t in enumerate(toks)])
    per_row = [rel(a[i], b[i]) for i in range(S)]
    print("layers %2d  prompt %d: batched vs token-by-token logits  rel L2 max %.3e  mean %.3e  last row %.3e   rows differing at all: %d"
          % (nl, S, max(per_row), float(np.mean(per_row)), per_row[-1], sum(1 for i in range(S) if (a[i] != b[i]).any())), flush=True)
    nkv = cfg["n_kv_heads"]
    for l in range(min(nl, 3)):
        for which, nm in ((0, "K"), (1, "V")):
            ra = np.concatenate([kv_a.read(l, hh, which, S).reshape(-1) for hh in range(nkv)])
            rb = np.concatenate([kv_b.read(l, hh, which, S).reshape(-1) for hh in range(nkv)])
            print("     layer %d %s rows: %6d of %d elements differ, rel L2 %.2e" % (l, nm, int((ra != rb).sum()), ra.size, rel(ra, rb)))
    del lm
dev.close()
