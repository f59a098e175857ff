import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ground-fusion_amd"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_py
    oracle_py.build()
    oracle_py.lib()
    return oracle_py


@pytest.fixture(scope="session")
def gf():
    import gfamd
    gfamd.lib()
    if gfamd.device_count() < 1:
        pytest.fail("no HIP device: the HIP path must run on the GPU box (no CPU fallback)")
    return gfamd
