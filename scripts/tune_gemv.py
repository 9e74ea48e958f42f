"""GEMV kernel tuning sweep on the 8B shapes (run on the GPU box): python scripts/tune_gemv.py"""
import ctypes as C
import sys
sys.path.insert(0, ".")
from blazr_amd import _lib as L
from blazr_amd import runtime

dev = runtime.Device(0)
SHAPES = {"qkv": (6144, 4096, 1), "o": (4096, 4096, 0), "gateup": (28672, 4096, 1), "down": (4096, 14336, 2)}


def run(N, K, gw, mode, flags=0, iters=24):
    nbuf = max(2, min(16, int(400e6 / (N * K / 2)) + 1))
    t = C.c_double()
    L.check(L.lib().bz_tune_gemv(dev.h, N, K, gw, mode, nbuf, iters, flags, C.byref(t)))
    return t.value


for name, (N, K, mode) in SHAPES.items():
    G = K // 128
    mb = (N * K / 2 + N * G * 2.5) / 1e6
    for gw in [g for g in (1, 2, 4, 7, 8, 14, 16) if G % g == 0]:
        wgs = ((N + 255) // 256) * (G // gw)
        row = []
        for flags in (0, 8, 7, 15):
            us = run(N, K, gw, mode, flags)
            row.append("%6.1f" % us)
        us0 = float(row[0])
        print("%-7s gw=%2d wgs=%5d  us[npf2, npf4, npf2-bare, npf4-bare]= %s   -> %.0f GB/s" % (name, gw, wgs, " ".join(row), mb / us0 * 1e3), flush=True)
    # plain prologue for comparison on the norm/silu shapes
    if mode:
        us = run(N, K, [g for g in (4, 2, 1) if G % g == 0][0], 0)
        print("%-7s plain-prologue gw=%d: %.1f us" % (name, [g for g in (4, 2, 1) if G % g == 0][0], us), flush=True)
dev.close()
