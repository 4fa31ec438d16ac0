import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')

# The placement probe (recstudio_amd/placement.py) warms the GPU for 0.6 s before it times a candidate allocation -- what a
# benchmark wants, and 3.5 minutes over the hundreds of distinct arena sizes of this suite.  The tests check VALUES, which no
# placement decision can change (test_placement_changes_addresses_only); they keep the default path (placement on) with a
# token warm-up.
os.environ.setdefault('RSA_PLACEMENT_WARM_S', '0.02')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + '.npz'))
    return load
