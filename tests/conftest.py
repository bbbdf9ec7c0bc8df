import os
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    config.addinivalue_line("markers", "timeout: pytest-timeout's per-test limit")


def pytest_collection_modifyitems(config, items):
    # a kernel that never returns would hold the GPU box until the driver's own limit: every GPU test
    # gets a deadline (pytest-timeout, 'thread' method: the process exits even from inside a blocking
    # HIP call); the slowest of them takes well under a minute
    if not config.pluginmanager.hasplugin("timeout"):
        return
    for item in items:
        if "gpu" in item.keywords and item.get_closest_marker("timeout") is None:
            item.add_marker(pytest.mark.timeout(420, method="thread"))


@pytest.fixture(scope="session")
def oisst():
    d = np.load(os.path.join(GOLDEN, "oisst_2003_2004.npz"))
    out = {k: d[k] for k in d.files}
    # "days since 2003-01-01 12:00:00", proleptic_gregorian
    out["time64"] = (np.datetime64("2003-01-01") + out["time"].astype("timedelta64[D]"))
    return out


@pytest.fixture(scope="session")
def clim_golden():
    d = np.load(os.path.join(GOLDEN, "clim_oisst.npz"))
    return {k: d[k] for k in d.files}


@pytest.fixture(scope="session")
def literals():
    d = np.load(os.path.join(GOLDEN, "literals.npz"))
    return {k: d[k] for k in d.files}
