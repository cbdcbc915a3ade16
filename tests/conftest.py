import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import bindings
    return bindings.Oracle()


@pytest.fixture(scope="session")
def reference():
    from oracle import bindings
    if not bindings.have_reference():
        pytest.skip("oracle/_ref/libwhisper_ref.so not built (needs /root/reference)")
    return bindings.Reference()
