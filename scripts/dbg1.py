import sys, numpy as np
sys.path.insert(0, '.')
from blazr_amd import runtime, synth
from oracle import orc_py
dev = runtime.Device(0)
print(dev.name())
model = synth.make_llama("tiny-awq")
lm = runtime.LoadedModel.from_synth(dev, model)
for l in range(2):
    p = "model.layers.%d." % l
    for short, hf in (("q","self_attn.q_proj"),("k","self_attn.k_proj"),("v","self_attn.v_proj"),("o","self_attn.o_proj"),("gate","mlp.gate_proj"),("up","mlp.up_proj"),("down","mlp.down_proj")):
        want = orc_py.OrcLinear(model["layers"][l][short]).dequant()
        got = lm.dequant(p+hf+".weight")
        bad = np.argwhere(got != want)
        print(l, short, want.shape, "nbad", len(bad), "first", bad[:3].tolist(), "rows bad", np.unique(bad[:,0])[:10].tolist() if len(bad) else [])
dev.close()
