"""Test configuration: registers the ``gpu`` marker (tests that need a real MI355X) and puts the repository root on
``sys.path``. ``-m "not gpu"`` tests cover the oracle against the golden vectors, the host logic and the C-ABI's
symbol table; ``-m gpu`` tests are the parity tests proper and call through the C ABI."""

import sys
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parent.parent
if str(REPO) not in sys.path:
    sys.path.insert(0, str(REPO))

GOLDEN = REPO / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built_library():
    """Builds libgtsfm_amd.so if needed (hipcc cross-compiles without a GPU)."""
    from gtsfm_amd.csrc import build

    return build.build(verbose=False)


@pytest.fixture(scope="session")
def gpu_device(built_library):
    import torch

    if not torch.cuda.is_available():
        pytest.fail("a -m gpu test was selected but no GPU is visible")
    return torch.device("cuda:0")
