"""pytest configuration: markers, import paths, shared fixtures."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "quadruped-reactive-walking_amd")
for p in (ROOT, PKG, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle  # oracle/oracle.py (test infrastructure)

    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def synth_mod():
    import synth

    return synth
