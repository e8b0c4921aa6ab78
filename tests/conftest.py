"""pytest configuration: path setup, the `gpu` marker, golden-fixture loader."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "blurry-edges_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


_CONFIG = None


def pytest_configure(config):
    global _CONFIG
    _CONFIG = config
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.hookimpl(trylast=True)
def pytest_runtest_logreport(report):
    """Flush the progress output after every test: with stdout on a pipe or a file the dots sit in an 8 KB buffer, and a runner that
    watches for output (the GPU box kills a command that has been silent for 7 minutes) sees a suite with one 3-minute test as hung."""
    tr = _CONFIG.pluginmanager.get_plugin("terminalreporter") if _CONFIG is not None else None
    if tr is not None:
        try:
            tr._tw.flush()
        except Exception:
            pass
    sys.stdout.flush()


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


@pytest.fixture(scope="session")
def golden():
    return load_golden


def relmax(a, b):
    """L-inf error normalised by the L-inf norm of the expected tensor (SURVEY §8c 'L∞/L∞')."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-30))
