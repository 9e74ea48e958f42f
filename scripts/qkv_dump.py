"""Diagnostic: the raw fixed-point q/k/v accumulator of ONE layer from the slim kernel and from the generic kernel (BZ_NO_SLIM_QKV=1), same input, and the
exact dot products in numpy -- which columns differ, by how many grid units, and how the difference is structured over tiles.
usage: python scripts/qkv_dump.py [layer=3]   (spawns itself twice with BZ_DUMP_QKV)"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
layer = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1] != "child" else 3

if len(sys.argv) > 1 and sys.argv[1] == "child":
    layer = int(sys.argv[2])
    from blazr_amd import _lib as L, runtime, synth
    from oracle import orc_py
    model = synth.make_llama("llama3-8b-awq-2l", n_layers=layer + 1)
    cfg = model["config"]
    dev = runtime.Device(0)
    lm, om = runtime.LoadedModel.from_synth(dev, model), orc_py.OrcLlama(model)
    okv = om.new_kv(8)
    oh, opm = om.embed([17]), None
    if layer > 0:
        oh, opm = om.layers_range(oh, opm, okv, 0, layer, 0)
    h = (oh + (opm if opm is not None else 0.0)).astype(np.float16).astype(np.float32)
    np.save(sys.argv[3] + ".h.npy", h)
    kv = runtime.LayeredKvCache(dev, layer + 1, 1, cfg["n_kv_heads"], 8, cfg["max_seq_len"], cfg["head_dim"], L.F16)
    lm.forward_layers_range(dev.tensor(h), None, kv, layer, layer + 1, 0)
    dev.synchronize()
    hd = h.astype(np.float64).reshape(-1)
    print("exact sum of squares %.17g; per 512-element wave share: %s" % (float((hd ** 2).sum()), " ".join("%.17g" % float((hd[i * 512:(i + 1) * 512] ** 2).sum()) for i in range(8))), file=sys.stderr)
    dev.close()
    sys.exit(0)

out = {}
for name, env in (("slim", {"BZ_NO_PERSIST": "1"}), ("generic", {"BZ_NO_PERSIST": "1", "BZ_NO_SLIM_QKV": "1"})):
    e = dict(os.environ); e.update(env); e["BZ_DUMP_QKV"] = "/tmp/qkv_%s.bin" % name; e["BZ_DUMP_LAYER"] = str(layer)
    subprocess.run([sys.executable, os.path.abspath(__file__), "child", str(layer), "/tmp/qkv_%s" % name], env=e, check=True)
    out[name] = np.fromfile("/tmp/qkv_%s.bin" % name, dtype=np.int64)
a, b = out["slim"], out["generic"]
d = a - b
print("layer %d: accumulator entries that differ: %d of %d; |diff| in grid units (2^-44): max %d, values: %s" % (layer, int((d != 0).sum()), len(d), int(np.abs(d).max()), np.unique(d)[:12]))
# exact values
from blazr_amd import synth
from oracle import orc_py
model = synth.make_llama("llama3-8b-awq-2l", n_layers=layer + 1)
lay = model["layers"][layer]
h = np.load("/tmp/qkv_slim.h.npy").reshape(-1)
H = len(h)
xn = np.empty(H, dtype=np.float32)
nw = np.ascontiguousarray(lay["attn_norm"], dtype=np.float32)
orc_py.lib().orc_rms_norm(h.ctypes.data_as(orc_py.C.c_void_p), nw.ctypes.data_as(orc_py.C.c_void_p), H, float(model["config"]["rms_eps"]), orc_py.F16, xn.ctypes.data_as(orc_py.C.c_void_p))
W = np.concatenate([orc_py.OrcLinear(lay[k]).dequant().astype(np.float64) for k in ("q", "k", "v")], axis=0)
exact = W @ xn.astype(np.float64)
for name in ("slim", "generic"):
    err = out[name].astype(np.float64) * 2.0 ** -44 - exact
    print("  %-8s vs exact: max |err| %.3e, rms %.3e; columns with |err| > 1e-9: %d" % (name, np.abs(err).max(), np.sqrt((err ** 2).mean()), int((np.abs(err) > 1e-9).sum())))
    bad = np.abs(err) > 1e-9
    if bad.any():
        tiles = np.nonzero(bad.reshape(-1, 64).any(axis=1))[0]
        print("           tiles with a bad column: %d of %d (first: %s)" % (len(tiles), len(bad) // 64, tiles[:16]))
        # error = W . dx for which dx?  6144 equations, 4096 unknowns: solve for the activation perturbation that explains it
        sol, res, *_ = np.linalg.lstsq(W, err, rcond=None)
        r = np.linalg.norm(W @ sol - err) / np.linalg.norm(err)
        nz = np.nonzero(np.abs(sol) > 1e-8)[0]
        print("           W . dx = err solved: residual %.2e; dx nonzero at %d positions" % (r, len(nz)))
        if len(nz):
            print("           k range %d..%d; per 128-group counts: %s" % (nz.min(), nz.max(), np.bincount(nz // 128, minlength=H // 128).tolist()))
            for k in nz[:24]:
                xv = float(xn[k]); g = k // 128
                gm = float(np.abs(xn[g * 128:(g + 1) * 128]).max())
                print("             k %4d (group %2d, octet %3d, lane-in-octet %d): x %.9g  dx %.4e  dx/x %.3e  group max %.6g  dx / (group max 2^-29) %.3f" %
                      (k, g, k // 8, k % 8, xv, sol[k], sol[k] / xv if xv else 0.0, gm, sol[k] / (gm * 2.0 ** -29)))
