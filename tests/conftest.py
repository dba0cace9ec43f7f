import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: takes more than ~30 s on the 8-core CPU container")


def pytest_collection_modifyitems(config, items):
    import torch
    # The CPU oracle (checker) runs on torch's intra-op pool, which sizes itself by the HOST's core count; a GPU box hands a
    # job 16 of its cores, and a pool of 100+ threads on 16 cores made the oracle-bound parity tests several times slower
    # (the fp64 pass of the 512 px networks: 175 s).  bench.py's cpu_baseline leg does the same.
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(16, avail)))
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
