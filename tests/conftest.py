import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ob():
    """the CPU oracle binding (test infrastructure)"""
    from oracle import binding
    binding.build()
    return binding


@pytest.fixture(scope="session")
def pie_mod():
    """the product's host mirror (nested_hashing_psi_amd.pie over libpiehip.so)"""
    from nested_hashing_psi_amd import pie
    return pie
