import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_addoption(parser):
    parser.addoption("--runslow", action="store_true", default=False,
                     help="also run the tests marked `slow` (long trajectories; PTAMD_RUN_SLOW=1 does the same)")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: minutes-long variants (200-step trajectories) of tests whose short forms run by "
                                       "default; skipped unless --runslow / PTAMD_RUN_SLOW=1 (builder-run, records under profiles/)")


def pytest_collection_modifyitems(config, items):
    if config.getoption("--runslow") or os.environ.get("PTAMD_RUN_SLOW") == "1":
        return
    skip = pytest.mark.skip(reason="slow variant: run with --runslow or PTAMD_RUN_SLOW=1")
    for item in items:
        if "slow" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))
        return cache[name]

    return load
