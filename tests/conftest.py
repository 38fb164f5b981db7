import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
TESTCASES = os.path.join(GOLDEN, "testcases")
ALL_CASES = ["b20", "b30", "b40", "b50", "b60", "b70", "b80", "b90", "b100", "b200", "b512", "b1024"]


_CONFIG = None


def pytest_runtest_logreport(report):
    """Push the progress dots out after every test: on the GPU box the runner's output goes through a pipe, and a run
    that writes nothing for minutes is taken to be hung."""
    tr = _CONFIG.pluginmanager.get_plugin("terminalreporter") if _CONFIG is not None else None
    if tr is not None:
        try:
            tr._tw.flush()
        except Exception:  # noqa: BLE001
            pass
    sys.stdout.flush()


def pytest_configure(config):
    global _CONFIG
    _CONFIG = config
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long CPU-only oracle runs, opt in with NB_SLOW=1")
    config.addinivalue_line("markers", "heavy: GPU tests of about a minute each (full-size 8-rank shapes); part of -m gpu, "
                                       "deselect with -m 'gpu and not heavy' when the suite must be short")


@pytest.fixture(scope="session")
def oracle():
    """The CPU checker (oracle/). Test infrastructure: never imported by the product package."""
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def nb():
    """The product package, with its HIP library built."""
    import nbody_amd
    from nbody_amd import capi
    if not (os.path.exists(capi.library_path()) and os.path.exists(os.path.join(ROOT, "bin", "hw5"))):
        import __graft_entry__
        __graft_entry__.build()
    return nbody_amd


def case_path(case, ext):
    return os.path.join(TESTCASES, f"{case}.{ext}")


def read_golden(case):
    with open(case_path(case, "out")) as f:
        lines = f.read().split("\n")
    dev, cost = lines[2].split()
    return float(lines[0]), int(lines[1]), int(dev), float(cost), "\n".join(lines)
