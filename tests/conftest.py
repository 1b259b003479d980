import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as graft  # noqa: E402

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")
    config.addinivalue_line("markers", "slow: minutes of CPU; skipped unless --runslow")


def pytest_addoption(parser):
    parser.addoption("--runslow", action="store_true", default=False, help="run minutes-long CPU pins")


def pytest_collection_modifyitems(config, items):
    if config.getoption("--runslow"):
        return
    skip = pytest.mark.skip(reason="needs --runslow")
    for item in items:
        if "slow" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def O():
    """The CPU oracle (test infrastructure)."""
    mod = graft.load_oracle()
    mod.build()
    return mod


@pytest.fixture(scope="session")
def pkg():
    return graft.load_package()


@pytest.fixture(scope="session")
def renderer(pkg):
    """One dmt_ctx on device 0.  Fails loudly (no fallback) when the HIP library or GPU is missing."""
    r = pkg.Renderer(0)
    yield r
    r.close()


def golden(name):
    return np.load(GOLDEN / name)


def film_rmse(a, b):
    """scripts/rmse.py:15-19 of the reference: per-pixel sqrt(mean_c (a-b)^2), averaged over pixels."""
    d = a[..., :3].astype(np.float64) - b[..., :3].astype(np.float64)
    return float(np.sqrt((d ** 2).mean(axis=-1)).mean())
