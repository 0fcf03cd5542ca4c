import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def host():
    from raytracing_rust_amd import Host

    return Host()


@pytest.fixture(scope="session")
def orc64():
    from oracle.oracle import Oracle

    return Oracle("f64")


@pytest.fixture(scope="session")
def orc32():
    from oracle.oracle import Oracle

    return Oracle("f32")


@pytest.fixture(autouse=True)
def _release_host_objects(request):
    """Every test builds its own worlds through the session's Host; what it built (scenes resident on the GPU with their
    per-sample buffers — up to 25 GB each) is released when it ends, so that the suite's footprint does not depend on how
    many tests ran before (the 8-ranks-on-one-GPU test needs 200 GB free)."""
    yield
    if "host" in request.fixturenames:
        request.getfixturevalue("host").free_all()
