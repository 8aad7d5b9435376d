import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
    """`gpu` is not only a label: on a box without a HIP device every test carrying it is skipped (also the ones that only
    launch subprocesses), so a plain `pytest tests` stays green there.  device_count() does not initialise the GPU."""
    import torch
    if torch.cuda.device_count() > 0:
        return
    skip = pytest.mark.skip(reason='no HIP device on this box (run with -m gpu on an MI355X)')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope='session')
def golden():
    import numpy as np

    def load(name):
        return dict(np.load(os.path.join(GOLDEN, name)))
    return load
