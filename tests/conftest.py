import os
import sys
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'))


@pytest.fixture(scope='session')
def golden():
    return load_golden


def numpy_uses_svml_exp():
    """True where the running NumPy evaluates float64 np.exp with the SVML routine it bundles (x86-64 with AVX512_SKX):
    the hosts the goldens were generated on, and the ones on which 'the reference's bits' of an exp are defined the
    way beta_cores_amd/csrc/bc_np_exp.h restates them.  (The product's own probe: beta_cores_amd/util/numpy_bits.py.)"""
    from beta_cores_amd.util.numpy_bits import numpy_uses_svml_exp as probe
    return probe()
