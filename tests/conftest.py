import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def pytest_collection_modifyitems(config, items):
    # a GPU test that does not come back must fail, not hang the run (pytest-timeout, if present)
    for item in items:
        if "gpu" in item.keywords and not any(m.name == "timeout" for m in item.iter_markers()):
            item.add_marker(pytest.mark.timeout(300))


@pytest.fixture(scope="session")
def oracle():
    from tests import oracle_lib

    return oracle_lib.load()
