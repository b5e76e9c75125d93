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
    way beta_cores_amd/csrc/bc_np_exp.h restates them."""
    import math
    import numpy as np
    try:
        from numpy._core._multiarray_umath import __cpu_features__ as feats
    except Exception:
        try:
            from numpy.core._multiarray_umath import __cpu_features__ as feats
        except Exception:
            return False
    if not feats.get('AVX512_SKX', False):
        return False
    x = -np.random.RandomState(0).uniform(0, 50, 20000)
    return bool((np.exp(x) != np.array([math.exp(v) for v in x])).any())
