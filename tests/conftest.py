import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_addoption(parser):
    parser.addoption("--waldo-lib", default=None,
                     help="run the suite against another build of the library (a tools_dev/build_variant.py variant)")


def pytest_configure(config):
    if config.getoption("--waldo-lib", default=None):
        from waldo_amd import _lib
        _lib.use_library(config.getoption("--waldo-lib"))
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "live_ref: differential test against /root/reference (build container only)")


def load_golden(name):
    """Golden vectors generated from the reference itself by oracle/make_golden.py."""
    data = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: torch.from_numpy(np.asarray(data[k])) for k in data.files}


@pytest.fixture
def golden():
    return load_golden


@pytest.fixture(scope="session")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
