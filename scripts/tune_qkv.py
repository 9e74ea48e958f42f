"""The q/k/v GEMV kernel alone (bz_tune_gemv, N 6144, K 4096, fused residual + RMSNorm prologue): mean dispatch time + per-wave stamps."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blazr_amd import _lib as L, runtime
dev = runtime.Device(0)
us = C.c_double()
for rep in range(3):
    L.check(L.lib().bz_tune_gemv(dev.h, 6144, 4096, 2, 1, 6, 40, 0, C.byref(us)))
    print("qkv slim N=6144 K=4096: %.2f us" % us.value)
L.check(L.lib().bz_tune_gemv(dev.h, 6144, 4096, 2, 1, 6, 6, 16, C.byref(us)))
print("stamp build: %.2f us" % us.value)
dev.close()
