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


def test_product_does_not_touch_oracle_outside_the_package():
    """the same for the other product sources: the C header, the C++ driver, and bench.py outside its cpu_baseline / parity leg"""
    for rel in ("include/blazr_hip.h", "tools/bz_run.cpp"):
        text = open(os.path.join(ROOT, rel), errors="replace").read()
        for bad in ("liborc", "orc_py", "from oracle", "import oracle", "orc.h\"", "dlopen"):
            assert bad not in text, (rel, bad)
    bench = open(os.path.join(ROOT, "bench.py")).read()
    head, marker, tail = bench.partition("        # CPU baseline: the oracle")
    assert marker, "bench.py lost its cpu_baseline marker"
    for bad in ("liborc", "orc_py", "from oracle", "import oracle"):
        assert bad not in head, "bench.py touches oracle/ (%s) before its cpu_baseline leg" % bad


def test_no_cpp_exception_can_cross_the_c_abi():
    """include/blazr_hip.h promises that no C++ exception crosses the boundary: every status-returning extern "C" function with a body of more
    than one line sits between BZ_API_BEGIN / BZ_API_END (bz_internal.h: bad_alloc -> BZ_E_OOM, anything else -> BZ_E_INVALID)"""
    n = 0
    for f in sorted(os.listdir(os.path.join(ROOT, "blazr_amd", "csrc"))):
        if not f.endswith(".hip"):
            continue
        lines = open(os.path.join(ROOT, "blazr_amd", "csrc", f)).read().split("\n")
        for i, ln in enumerate(lines):
            if ln.startswith('extern "C" int ') and not ln.rstrip().endswith("}"):
                j = i
                while not lines[j].rstrip().endswith("{"):
                    j += 1
                assert lines[j + 1].strip() == "BZ_API_BEGIN", (f, i + 1, ln)
                k = j + 1
                while lines[k] != "}":
                    k += 1
                assert lines[k - 1].strip() == "BZ_API_END", (f, i + 1, ln)
                n += 1
    assert n >= 70


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
