import faulthandler
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "watchdog(seconds): per-test limit of the hang watchdog (default 150 s; the workload-size tests wait for the CPU oracle)")


@pytest.fixture(scope="session")
def device():
    """One HIP device handle for the whole session (gpu tests only)."""
    from blazr_amd import runtime
    d = runtime.Device(0)
    yield d
    d.close()


@pytest.fixture(autouse=True)
def _watchdog(request):
    """A GPU call that never returns would otherwise sit in C until the box's limit: dump the Python stack and exit after
    150 s in one test (faulthandler's watchdog thread works while the main thread is blocked inside ctypes)."""
    if request.node.get_closest_marker("gpu") is None:
        yield
        return
    wd = request.node.get_closest_marker("watchdog")
    limit = int(wd.args[0]) if wd is not None and wd.args else int(os.environ.get("BZ_TEST_WATCHDOG_S", "150"))
    faulthandler.dump_traceback_later(limit, exit=True)
    try:
        yield
    finally:
        faulthandler.cancel_dump_traceback_later()
