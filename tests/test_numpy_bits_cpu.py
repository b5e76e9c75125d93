"""util/numpy_bits.py: the probe of NumPy's np.exp routine, the warning a device projector gives on a host whose NumPy is
not the one csrc/bc_np_exp.h restates, and the host-side evaluation of constant rows (LinearRegression.host_constants:
model_neurlinr.py:102-110 at x = 0) -- everything that runs without a GPU."""
import warnings

import numpy as np
import pytest

from oracle import models_ref as M


def test_probe_is_cached_and_boolean():
    from beta_cores_amd.util import numpy_bits
    a = numpy_bits.numpy_uses_svml_exp()
    assert isinstance(a, bool) and numpy_bits.numpy_uses_svml_exp() is a


def test_warning_when_numpy_is_not_the_restated_one(monkeypatch):
    from beta_cores_amd.util import numpy_bits
    from beta_cores_amd.likelihoods import LinearRegression, GaussianLocation, LogisticRegression
    monkeypatch.setattr(numpy_bits, '_cached', False)
    monkeypatch.setattr(numpy_bits, '_warned', set())
    with pytest.warns(UserWarning, match='evaluated on the host'):
        assert numpy_bits.warn_if_constant_bits_differ(LinearRegression(1.0)) is True
    with pytest.warns(UserWarning, match='keeps the AVX-512 NumPy bits'):
        assert numpy_bits.warn_if_constant_bits_differ(GaussianLocation(np.eye(3), 0.)) is True
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        assert numpy_bits.warn_if_constant_bits_differ(LinearRegression(2.0)) is True       # once per model class
        assert numpy_bits.warn_if_constant_bits_differ(LogisticRegression()) is False       # its constant comes from NumPy already
    monkeypatch.setattr(numpy_bits, '_cached', True)
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        assert numpy_bits.warn_if_constant_bits_differ(LinearRegression(1.0)) is False


@pytest.mark.parametrize('sigsq,beta', [(1.0, 0.1), (2.5, 0.5), (0.3, 0.2)])
def test_host_constants_are_the_reference_expression_at_zero_features(sigsq, beta):
    """The S equal values a row [0, ..., 0, y] projects to under the beta-likelihood: LinearRegression.host_constants against
    the oracle's restatement of model_neurlinr.py:102-110 on such rows, bit for bit."""
    from beta_cores_amd.likelihoods import LinearRegression
    rng = np.random.RandomState(3)
    y = np.sort(rng.randn(40) * 4.)
    Z = np.zeros((40, 7))
    Z[:, -1] = y
    th = rng.randn(5, 6)
    ref = M.linreg_beta_lik(Z, th, beta, sigsq)
    assert np.all(ref == ref[:, :1])                      # constant rows indeed
    got = LinearRegression(sigsq).host_constants(y, beta)
    assert np.array_equal(got, ref[:, 0])
