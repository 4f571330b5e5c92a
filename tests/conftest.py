import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _native_built():
    """Make sure the in-tree native artefacts exist (no-op when they are up to date)."""
    from amplisolve_amd import build as b

    if not (os.path.exists(b.HIP_LIB) and os.path.exists(b.HOST_LIB)):
        b.build_hip()
        b.build_host()
    from oracle import pyoracle

    if not os.path.exists(pyoracle.LIB_PATH):
        pyoracle.build()
    yield


@pytest.fixture(scope="session")
def ctx():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import time

    from amplisolve_amd import AmpliError, Context

    c = None
    for attempt in range(3):  # seen once on a GPU box: the first in-process initialisation right after a burst of short-lived GPU
        try:                  # processes found no device, while a process started a moment later did
            c = Context(0)
            break
        except AmpliError:
            if attempt == 2:
                raise
            time.sleep(2.0)
    yield c
    c.close()


@pytest.fixture(autouse=True)
def _clear_kernel_flags(request):
    """The session's context keeps AMPLI_FLAG_* bits until somebody reads them: start every test with none raised."""
    if "ctx" in request.fixturenames:
        request.getfixturevalue("ctx").flags()
    yield
