import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gpu_ctx():
    """One rcn_ctx on cuda:0 for the whole GPU session (one process, one context)."""
    import torch  # noqa: F401  (first, so librcn.so shares torch's HIP runtime)
    from reconstructor_amd import _lib
    ctx = _lib.Context(0)
    yield ctx
    ctx.close()
