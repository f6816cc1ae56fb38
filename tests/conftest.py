import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """A plain `pytest tests` on a machine without a GPU SKIPS the gpu-marked tests instead of failing 190 of them.  When the
    gpu tests are asked for by name (`-m gpu`, the driver's round-end run) nothing is skipped: without a GPU they fail, loudly."""
    import torch
    if torch.cuda.is_available() or "gpu" in (config.getoption("-m") or "").replace("not gpu", ""):
        return
    skip = pytest.mark.skip(reason="needs an MI355X (torch.cuda.is_available() is False); run with -m gpu on the GPU box")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def pkg():
    from ggq_pkg import load_package
    return load_package()


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
