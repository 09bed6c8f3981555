import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def pytest_sessionstart(session):
    # The first `import torch` on a fresh GPU box can take minutes while the image pages in (seen: > 300 s).
    # Do it here, outside every per-test timeout, when the GPU tests are selected.
    expr = session.config.getoption("markexpr", "") or ""
    if "gpu" in expr and "not gpu" not in expr:
        try:
            import torch  # noqa: F401
        except Exception:  # the tests that need it report the problem themselves
            pass


def pytest_collection_modifyitems(config, items):
    # a GPU test that does not come back must fail, not hang the run (pytest-timeout, if present)
    for item in items:
        if "gpu" in item.keywords and not any(m.name == "timeout" for m in item.iter_markers()):
            item.add_marker(pytest.mark.timeout(600))


@pytest.fixture(scope="session")
def oracle():
    from tests import oracle_lib

    return oracle_lib.load()
