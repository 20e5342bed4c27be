"""Shared test plumbing: the ``gpu`` marker, import paths, fixture loading."""
import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def pkg():
    """The product package (its directory name has a hyphen, so import it by string)."""
    return importlib.import_module("gan-discovery-pso_amd")


@pytest.fixture(scope="session")
def synth():
    return importlib.import_module("gan-discovery-pso_amd.synth")


@pytest.fixture(scope="session")
def oracle():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    return importlib.import_module("pso_oracle")
