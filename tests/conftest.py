import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)
    return load


@pytest.fixture(autouse=True)
def _fp32_point_rows():
    """MotionNet.forward records its compute dtype in pcaccumulation_amd.ops (the element type of the per-point MLP rows); tests that
    call sub-modules directly must not inherit the bf16 / fp32x3 mode of whichever model ran before them."""
    import torch
    from pcaccumulation_amd import ops
    ops.set_point_dtype(torch.float32)
    ops.set_split(False)
    ops.set_mixed(False)
    ops.set_poison(False)
    yield
    ops.set_point_dtype(torch.float32)
    ops.set_split(False)
    ops.set_mixed(False)
    ops.set_poison(False)
    from pcaccumulation_amd import native
    if native._lib is not None:                   # a test may have changed a launcher switch through the environment (monkeypatch is undone by now)
        native.reload_switches()


def pytest_collection_modifyitems(config, items):
    """Tests that start process trees (bench.py under the launcher, subprocess benches) run LAST: `pytest -x` then reaches every parity test before
    the first process tree is started, and a launch that misbehaves cannot take the parity evidence with it (round 4: one hanging launch collected
    first zeroed the GPU suite).  Stable for everything else."""
    last = ('test_bench_multirank.py',)
    items.sort(key=lambda it: 1 if os.path.basename(str(it.fspath)) in last else 0)
