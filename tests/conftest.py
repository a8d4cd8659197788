import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import slalibs
    return slalibs.oracle()


@pytest.fixture(scope="session")
def ref():
    import slalibs
    r = slalibs.ref()
    if r is None:
        pytest.skip("oracle/_ref/libsla_ref.so not built (reference sources absent)")
    return r
