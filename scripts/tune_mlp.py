"""Fused MLP kernel alone (bz_tune_mlp): mean dispatch time on cold weights + the diagnostic build's per-wave phase timeline.
usage: python scripts/tune_mlp.py [--stamps]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blazr_amd import _lib as L, runtime  # noqa: E402

dev = runtime.Device(0)
us = C.c_double()
for rep in range(3):
    L.check(L.lib().bz_tune_mlp(dev.h, 4096, 14336, 6, 40, 0, C.byref(us), None))
    print("mlp H=4096 I=14336: %.2f us per launch (91.52 MB -> %.0f GB/s)" % (us.value, 91.521024e6 / us.value / 1e3))
if "--stamps" in sys.argv:
    st = (C.c_longlong * 512)()
    L.check(L.lib().bz_tune_mlp(dev.h, 4096, 14336, 6, 10, 0, C.byref(us), st))
    names = ["entry", "loads issued", "norm done", "xs written", "quant done", "g0 before", "g0 consumed", "g1 before", "g1 consumed", "dots done", "part barrier",
             "silu+quant64", "down tile 0 done", "atomics issued", "atomics drained"]
    for b in range(2):
        print("workgroup %s (stamp build: %.2f us per launch); columns = waves 0, 5, 10, 15; us since the workgroup's first stamp" % ("0" if b == 0 else "113", us.value))
        for i, nm in enumerate(names):
            print("  %-16s" % nm, " ".join("%6.2f" % (st[(b * 16 + w) * 16 + i] / 100.0) for w in (0, 5, 10, 15)))
dev.close()
