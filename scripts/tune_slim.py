"""Back-to-back (warm instruction cache) timing of the q/k/v GEMV kernels: slim vs the generic split-K kernel (BZ_NO_SLIM_QKV=1)."""
import ctypes as C
import os
import sys
sys.path.insert(0, ".")
from blazr_amd import _lib as L
from blazr_amd import runtime
dev = runtime.Device(0)
t = C.c_double()
for it in range(3):
    L.check(L.lib().bz_tune_gemv(dev.h, 6144, 4096, 4, 1, 16, 24, 0, C.byref(t)))
    print("qkv N=6144 K=4096 norm prologue, back-to-back: %.2f us (slim=%s)" % (t.value, not os.environ.get("BZ_NO_SLIM_QKV")), flush=True)
