"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads and exports every symbol that
include/blazr_hip.h declares; the product path fails loudly (no CPU fallback) when there is no GPU."""
import os
import re

import pytest

from blazr_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "blazr_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(bz_[a-z0-9_]+)\s*\(", src)))


def test_header_and_ctypes_table_agree():
    assert _declared() == sorted(L.SYMBOLS)


def test_library_exports_every_declared_symbol():
    lib = L.lib()          # resolves each entry of SYMBOLS, raises AttributeError on a missing export
    assert lib.bz_abi_version() == L.ABI_VERSION
    for name in _declared():
        assert hasattr(lib, name), name


def test_header_cites_reference_interfaces():
    src = open(os.path.join(ROOT, "include", "blazr_hip.h")).read()
    for anchor in ("executor_generate.rs:357,372", "executor_generate.rs:259-262", "sampling.rs:445-460", "awq.rs:190-225",
                   "gptq.rs:198-259", "cuda_graphs.rs:97-189", "swarm_forward.rs:205,239-263"):
        assert anchor in src, anchor


def test_product_does_not_touch_oracle():
    """the product path must not import, link or dlopen anything under oracle/ (only tests/, smoke(), bench cpu leg may)"""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "blazr_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                for bad in ("liborc", "orc_py", "from oracle", "import oracle", "orc.h\"", "dlopen"):
                    assert bad not in text, (os.path.join(dirpath, f), bad)


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="GPU present")
def test_no_device_fails_loudly():
    import ctypes as C
    h = C.c_void_p()
    rc = L.lib().bz_device_open(0, C.byref(h))
    assert rc == L.E_NODEVICE
    assert b"no CPU fallback" in L.lib().bz_last_error() or b"gfx950" in L.lib().bz_last_error()
    from blazr_amd import runtime
    with pytest.raises(L.BlazrHipError):
        runtime.Device(0)
