import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'domain-rag-system_amd')
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


import pytest


@pytest.fixture(autouse=True)
def _device_checks_held(request):
    """After every GPU test: none of the library's device-side consistency checks fired
    (tests that provoke them on purpose clear the flags themselves)."""
    yield
    if request.node.get_closest_marker('gpu') is not None:
        import rag_hip
        rag_hip.check_faults()
