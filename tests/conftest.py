import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "ref: needs oracle/_ref/libfasp_ref.so (the reference build)")


@pytest.fixture(scope="session")
def fa():
    import faspsolver_amd as m
    m.lib()
    return m


@pytest.fixture(scope="session")
def gpu(fa):
    if not fa.available():
        pytest.fail("no HIP device: -m gpu tests need the MI355X (libfasp_hip has no CPU fallback)")
    return fa
