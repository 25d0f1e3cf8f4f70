import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'pc-accumulation-lib_amd')
GOLDEN = os.path.join(ROOT, 'tests', 'golden')
# the drop-in root goes FIRST so that its `datasets` package wins over the installed HuggingFace one
for p in (ROOT, PKG):
    if p in sys.path:
        sys.path.remove(p)
sys.path.insert(0, ROOT)
sys.path.insert(0, PKG)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)

    return load
