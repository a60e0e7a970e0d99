import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def gpu():
    """Bind the HIP library to GPU 0; GPU tests fail (not skip) when that is impossible."""
    import raytracing_amd
    raytracing_amd.init(0)
    return raytracing_amd
