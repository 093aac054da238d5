import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden():
    return dict(np.load(os.path.join(GOLDEN_DIR, 'reference_vectors.npz')))


def canonical_order(polys, scores):
    """Order-independent view of a decode result (the reference's own order is implementation-defined)."""
    flat = np.concatenate([polys.reshape(len(polys), -1), scores.reshape(-1, 1)], axis=1)
    order = np.lexsort(flat.T[::-1])
    return flat[order]
