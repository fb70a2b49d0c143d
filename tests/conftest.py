"""Test configuration: marker registration, import paths, fixture loading."""

import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')
for p in (ROOT, os.path.join(ROOT, 'ml-quant_amd'), GOLDEN):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU in this container')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


class Golden:
    """Lazy access to tests/golden/*.npz as torch tensors."""

    def __init__(self, name):
        self._z = np.load(os.path.join(GOLDEN, name + '.npz'))

    def __contains__(self, key):
        return key in self._z.files

    def keys(self):
        return list(self._z.files)

    def np(self, key):
        return self._z[key]

    def __getitem__(self, key):
        return torch.from_numpy(np.array(self._z[key]))

    def text(self, key):
        return bytes(self._z[key]).decode()

    def json(self, key):
        return json.loads(self.text(key))


@pytest.fixture(scope='session')
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = Golden(name)
        return cache[name]
    return load


def pytest_sessionfinish(session, exitstatus):
    """LSQ_RECORD_PARITY=1: write what the free-running parity assertions observed (tests/test_gpu_parity.py)."""
    if not os.environ.get('LSQ_RECORD_PARITY'):
        return
    mod = sys.modules.get('test_gpu_parity')
    if mod is None or not getattr(mod, 'OBSERVED', None):
        return
    out = os.path.join(ROOT, 'gpurun_out')
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, 'free_running_observed.json'), 'w') as f:
        json.dump({'observed_max': mod.OBSERVED, 'limits': mod.FREE_LIMIT}, f, indent=1)
