import os
import sys

import pytest

# The oracle is OpenMP code with many short parallel regions.  libgomp's default is to spin at barriers, which turns into
# minutes of wasted CPU as soon as anything else in the test process (a HIP runtime helper thread, a child process of a
# launch test) competes for the cores of a small CI box; a bounded spin (then sleep) keeps the short regions fast and the pathological case away.
os.environ.setdefault("GOMP_SPINCOUNT", "30000")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
if os.path.join(ROOT, "tests") not in sys.path:
    sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: CPU test of several minutes and tens of GB (deselected unless -m slow is given)")


def pytest_collection_modifyitems(config, items):
    # `-m "not gpu"` (the driver's CPU run) must stay a few minutes: slow tests run only when the marker expression names them
    if "slow" in (config.getoption("-m") or ""):
        return
    skip = pytest.mark.skip(reason="slow: run with -m slow")
    for it in items:
        if "slow" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as o
    o.build()
    return o


def pytest_sessionfinish(session, exitstatus):
    # what the history-parity tests measured (tests/parity_log.py) -> gpurun_out/r05_parity_devs.json
    try:
        import parity_log
        parity_log.flush()
    except Exception:
        pass
