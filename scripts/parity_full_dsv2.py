"""DeepSeek-V2-Lite at FULL depth (27 layers, 64 experts per MoE layer: a 31 GB host copy of the synthetic weights) against the CPU oracle: a short prompt through the
decode kernels and a few teacher-forced steps; prints how many logits differ.  (bench.py has no parity leg for this preset: generating the host copy takes minutes.)
usage: python scripts/parity_full_dsv2.py [n_prompt=10] [n_steps=6] [n_layers=27]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blazr_amd import runtime, synth  # noqa: E402
from oracle import orc_py  # noqa: E402

n_prompt = int(sys.argv[1]) if len(sys.argv) > 1 else 10
n_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
over = {"n_layers": int(sys.argv[3])} if len(sys.argv) > 3 else {}
t0 = time.time()
model = synth.make_dsv2("deepseek-v2-lite", **over)
cfg = model["config"]
print("host model: %d layers, generated in %.0f s" % (cfg["n_layers"], time.time() - t0), flush=True)
dev = runtime.Device(0)
lm = runtime.LoadedModel.from_synth(dev, model)
orc_py.set_threads(16)
om = orc_py.OrcDsv2(model)
print("loaded on both sides after %.0f s" % (time.time() - t0), flush=True)
p = [int(t) for t in synth.prompt_tokens(n_prompt, cfg["vocab"], seed=81)]
kv, okc = lm.new_kv_cache(n_prompt + n_steps + 8), om.new_cache(n_prompt + n_steps + 8)
got = [lm.forward_with_kv_cache(p, kv, 0, all_logits=True).to_numpy().reshape(n_prompt, -1)]
want = [om.forward(p, okc, 0, all_logits=True).reshape(n_prompt, -1)]
tok = int(want[0][-1].argmax())
for i in range(n_steps):
    got.append(lm.forward_with_kv_cache([tok], kv, n_prompt + i).to_numpy().reshape(1, -1))
    want.append(om.forward([tok], okc, n_prompt + i).reshape(1, -1))
    tok = int(want[-1][0].argmax())
    print("step %d done (%.0f s)" % (i, time.time() - t0), flush=True)
got, want = np.concatenate(got), np.concatenate(want)
rel = float(np.linalg.norm(got.astype(np.float64) - want) / np.linalg.norm(want))
print("deepseek-v2-lite, %d layers, %d prompt rows + %d decode rows: %d of %d logits differ from the oracle, relative L2 %.3e, argmax equal on %d of %d rows"
      % (cfg["n_layers"], n_prompt, n_steps, int((got != want).sum()), got.size, rel, int((got.argmax(1) == want.argmax(1)).sum()), got.shape[0]))
dev.close()
