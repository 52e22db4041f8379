import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the slowest tests go into every log (the driver's -m gpu run has a wall-clock limit: what eats it must be visible in the log it pulls)
    if getattr(config.option, "durations", None) is None:
        config.option.durations = 15
        config.option.durations_min = 1.0


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no ROCm device visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def dev():
    return torch.device("cuda:0")
