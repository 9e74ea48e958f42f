"""Names the first sub-op whose GPU output differs from the oracle's inside ONE layer of a Llama-3-8B-AWQ-shaped model.
Both sides start from the oracle's input of that layer (position 0: attention over a single token is the identity on v, so every difference is a
linear layer, a norm or SiLU).  Linear layers run through bz_quant_matmul (the generic int4 GEMV: same plane arithmetic as the fused kernels), the
exact product through numpy float64 on the dequantised weights.
usage: python scripts/parity_subops.py [layer=3] [n_layers=4] [token=17]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blazr_amd import _lib as L, runtime, synth  # noqa: E402
from oracle import orc_py  # noqa: E402

layer = int(sys.argv[1]) if len(sys.argv) > 1 else 3
nl = int(sys.argv[2]) if len(sys.argv) > 2 else layer + 1
tok = int(sys.argv[3]) if len(sys.argv) > 3 else 17
model = synth.make_llama("llama3-8b-awq-2l", n_layers=nl)
cfg = model["config"]
dev = runtime.Device(0)
lm, om = runtime.LoadedModel.from_synth(dev, model), orc_py.OrcLlama(model)
H, I, nq, nkv, hd = cfg["hidden"], cfg["inter"], cfg["n_heads"], cfg["n_kv_heads"], cfg["head_dim"]
lay = model["layers"][layer]
lo = orc_py.lib()


def f16(a):
    return np.asarray(a, dtype=np.float32).astype(np.float16).astype(np.float32)


def report(name, got, want, exact=None):
    got, want = np.asarray(got, np.float32).reshape(-1), np.asarray(want, np.float32).reshape(-1)
    bad = np.nonzero(got != want)[0]
    msg = "  %-28s %5d of %5d differ" % (name, len(bad), len(want))
    if len(bad):
        i = int(bad[0])
        msg += "   first at %d: gpu %.9g oracle %.9g" % (i, got[i], want[i])
        if exact is not None:
            msg += " exact %.12g" % float(np.asarray(exact).reshape(-1)[i])
    print(msg)


def rms(x, w):
    out = np.empty(H, dtype=np.float32)
    x = np.ascontiguousarray(x, dtype=np.float32)
    w = np.ascontiguousarray(w, dtype=np.float32)
    lo.orc_rms_norm(x.ctypes.data_as(orc_py.C.c_void_p), w.ctypes.data_as(orc_py.C.c_void_p), H, float(cfg["rms_eps"]), orc_py.F16, out.ctypes.data_as(orc_py.C.c_void_p))
    return out


def gpu_rms(x, w):
    xt, wt, yt = dev.tensor(np.ascontiguousarray(x, np.float32).reshape(1, H)), dev.tensor(np.ascontiguousarray(w, np.float32)), dev.zeros((1, H), L.F32)
    L.check(L.lib().bz_rms_norm(dev.h, xt.h, None, wt.h, 1, H, float(cfg["rms_eps"]), L.F16, yt.h, None))
    return yt.to_numpy().reshape(-1)


def lin(short, hf, x):
    name = "model.layers.%d.%s.weight" % (layer, hf)
    ol = orc_py.OrcLinear(lay[short])
    want = f16(ol.forward(x).reshape(-1))
    got = f16(lm.quant_matmul(name, x).reshape(-1))
    exact = ol.dequant().astype(np.float64) @ np.asarray(x, np.float64)
    report(short + " = R(W x)", got, want, exact)
    # how many elements of x are below 2^-19.5 of their 128-group maximum (inexact on the 32-bit grid)?
    xg = np.abs(np.asarray(x, np.float64)).reshape(-1, 128)
    mx = xg.max(axis=1, keepdims=True)
    small = ((xg > 0) & (xg < mx * 2.0 ** -19.5)).sum()
    if small:
        print("      (%d input elements lie below 2^-19.5 of their group maximum)" % small)
    return want


okv = om.new_kv(8)
oh, opm = om.embed([tok]), None
if layer > 0:
    oh, opm = om.layers_range(oh, opm, okv, 0, layer, 0)
h = f16(oh + (opm if opm is not None else 0.0)).reshape(-1)
print("layer %d of %d, token %d at position 0: sub-op outputs, GPU vs oracle, both fed the oracle's values" % (layer, nl, tok))
print("  |h| max %.4g, rms %.4g" % (np.abs(h).max(), np.sqrt((h.astype(np.float64) ** 2).mean())))
xn = rms(h, lay["attn_norm"])
report("xn = rmsnorm(h)", gpu_rms(h, lay["attn_norm"]), xn)
q = lin("q", "self_attn.q_proj", xn)
k = lin("k", "self_attn.k_proj", xn)
v = lin("v", "self_attn.v_proj", xn)
att = np.concatenate([v[(hq // (nq // nkv)) * hd:(hq // (nq // nkv) + 1) * hd] for hq in range(nq)])      # one cached position: softmax weight 1
o = lin("o", "self_attn.o_proj", att)
h2 = f16(h + o)
xn2 = rms(h2, lay["ffn_norm"])
report("xn2 = rmsnorm(h + o)", gpu_rms(h2, lay["ffn_norm"]), xn2)
g = lin("gate", "mlp.gate_proj", xn2)
u = lin("up", "mlp.up_proj", xn2)
a = f16(f16(np.array([lo.orc_silu(float(t)) for t in g], dtype=np.float32)) * u)
gt, ut, yt = dev.tensor(g.reshape(1, -1)), dev.tensor(u.reshape(1, -1)), dev.zeros((1, I), L.F32)
L.check(L.lib().bz_silu_mul(dev.h, gt.h, ut.h, I, L.F16, yt.h))
report("act = R(R(silu g) u)", yt.to_numpy().reshape(-1), a)
d = lin("down", "mlp.down_proj", a)
# and the fused layer itself on the same input
lh = dev.tensor(oh.astype(np.float32))
lpm = None if opm is None else dev.tensor(opm.astype(np.float32))
kv = runtime.LayeredKvCache(dev, nl, 1, nkv, 8, cfg["max_seq_len"], hd, L.F16)
gh, gpm = lm.forward_layers_range(lh, lpm, kv, layer, layer + 1, 0)
oh2, opm2 = om.layers_range(oh, opm, okv, layer, layer + 1, 0)
gk = np.concatenate([kv.read(layer, hh, 0, 1).reshape(-1) for hh in range(nkv)])
gv = np.concatenate([kv.read(layer, hh, 1, 1).reshape(-1) for hh in range(nkv)])
report("fused layer: K row (cache)", gk, k, orc_py.OrcLinear(lay["k"]).dequant().astype(np.float64) @ xn.astype(np.float64))
report("fused layer: V row (cache)", gv, v, orc_py.OrcLinear(lay["v"]).dequant().astype(np.float64) @ xn.astype(np.float64))
bad = np.nonzero(gv != v)[0]
for i in bad[:6]:
    ex = float(orc_py.OrcLinear(lay["v"]).dequant()[i].astype(np.float64) @ xn.astype(np.float64))
    print("      v[%d]: gpu %.9g oracle %.9g exact %.12g (f16 neighbours %.9g / %.9g)" % (i, gv[i], v[i], ex, np.float16(ex), np.nextafter(np.float16(ex), np.float16(np.inf if ex > float(np.float16(ex)) else -np.inf))))
report("fused layer: h'", gh.to_numpy(), oh2)
report("fused layer: mlp out", gpm.to_numpy(), opm2)
report("   (oracle h' vs sub-op h2)", h2, oh2)
report("   (oracle mlp vs sub-op down)", d, opm2)
dev.close()
