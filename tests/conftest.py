"""pytest configuration: registers the ``gpu`` marker and makes the repo root and the
drop-in root (``intro-tc-vae_amd/``) importable, the latter the way a user of the reference
would have it: its modules (``ops``, ``models``, ``solvers``) at the top level."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "intro-tc-vae_amd")
for p in (PKG, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
