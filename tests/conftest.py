import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'oracle')):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu via gpurun)')
    config.addinivalue_line('markers', 'reference: executes the unmodified reference (/root/reference or oracle/_ref/omnisafe_ref.zip)')


def pytest_collection_modifyitems(config, items):
    import torch

    have_gpu = torch.cuda.is_available()
    import ref_harness

    have_ref = ref_harness.reference_available()  # /root/reference, or the archive staged by build()
    for item in items:
        if 'gpu' in item.keywords and not have_gpu:
            item.add_marker(pytest.mark.skip(reason='no GPU in this container'))
        if 'reference' in item.keywords and not have_ref:
            item.add_marker(pytest.mark.skip(reason='reference not present'))


@pytest.fixture(scope='session')
def golden():
    import numpy as np

    def load(name):
        return dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))

    return load


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN
