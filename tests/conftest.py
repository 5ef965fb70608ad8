import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    # the oracle (torch CPU ops on small tensors) gets catastrophically slow with one thread per hardware thread on
    # the 128/256-CPU GPU hosts; 16 is near the optimum everywhere we measured
    import torch
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
