import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the library's test hooks (CATCHHIP_* switches that force one of several exact code paths) are only honoured
# with this set -- csrc/internal.h: chip_test_env, catch_amd/_lib.py: test_env
os.environ["CATCHHIP_TEST_HOOKS"] = "1"
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    """CPU oracle (test infrastructure): builds oracle/liboracle.so with gcc."""
    from oracle import oracle as o
    o.build()
    return o


@pytest.fixture(scope="session")
def ctx():
    from catch_amd import engine
    return engine.default_context()
