"""The dense row GEMV alone (bz_tune_rows) on the decode shapes of the 16-bit configs: mean dispatch time and GB/s per shape."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blazr_amd import _lib as L, runtime
SHAPES = [
    # name, N, K, wdt, mode (0 plain, 1 norm, 2 silu), sk (0 = auto)
    ("llama1b qkv", 3072, 2048, L.BF16, 1, 0),
    ("llama1b o_proj", 2048, 2048, L.BF16, 0, 0),
    ("llama1b gate_up", 16384, 2048, L.BF16, 1, 0),
    ("llama1b down", 2048, 8192, L.BF16, 2, 0),
    ("llama1b lm_head", 128256, 2048, L.BF16, 1, 0),
    ("mamba2 in_proj", 10576, 2560, L.BF16, 1, 0),
    ("mamba2 out_proj", 2560, 5120, L.BF16, 0, 0),
    ("dsv2 q+kv_a", 3648, 2048, L.BF16, 1, 0),
    ("dsv2 dense gate_up", 21888, 2048, L.BF16, 1, 0),
    ("dsv2 dense down", 2048, 10944, L.BF16, 2, 0),
]
only = sys.argv[1] if len(sys.argv) > 1 else None
dev = runtime.Device(0)
us = C.c_double()
for name, N, K, wdt, mode, sk in SHAPES:
    if only and only not in name:
        continue
    mb = N * K * 2 / 1e6
    nbuf = max(2, min(24, int(600 / mb) + 1))
    best = 1e9
    for rep in range(2):
        L.check(L.lib().bz_tune_rows(dev.h, N, K, wdt, mode, sk, nbuf, 40, C.byref(us)))
        best = min(best, us.value)
    print("%-20s N=%-6d K=%-5d %6.1f MB  %7.2f us  %6.0f GB/s" % (name, N, K, mb, best, mb * 1e3 / best), flush=True)
dev.close()
