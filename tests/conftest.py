"""pytest configuration: the `gpu` marker and shared fixtures."""
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import binding
    return binding.load_oracle()


@pytest.fixture(scope="session")
def ref_available():
    from oracle import binding
    return binding.load_ref() is not None
