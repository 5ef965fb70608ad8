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


@pytest.fixture(autouse=True)
def _collect_on_the_main_thread(request):
    """GPU tests build HIP graphs, private memory pools and autograd graphs with reference cycles.  Left to the cyclic
    collector they are destroyed whenever it happens to run -- possibly on autograd's worker thread in the middle of a later
    test, where tearing down a captured graph aborted the process once in ~8 full runs.  Collect deterministically, on the
    main thread, with the device idle, after every GPU test."""
    yield
    if "gpu" in request.keywords:
        import gc
        import torch
        if torch.cuda.is_available():
            torch.cuda.synchronize()
            gc.collect()
            torch.cuda.synchronize()
