import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from __graft_entry__ import load_package  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long CPU test")


@pytest.fixture(scope="session")
def pkg():
    return load_package()


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture
def engine_option():
    """Set kernel-selection options of an engine library for one test (sdm_set_option - the library reads no environment variable);
    everything is back at its default afterwards.  usage: engine_option(engine_or_bindings, "attn_nw", 8)"""
    touched = []

    def set_option(target, name, value):
        lib = getattr(target, "lib", target)
        lib.set_option(name, value)
        if lib not in touched:
            touched.append(lib)
    yield set_option
    for lib in touched:
        lib.reset_options()
