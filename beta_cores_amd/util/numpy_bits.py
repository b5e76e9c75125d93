"""Which routine does THIS host's NumPy evaluate float64 `np.exp` with?

Constant projection rows (a data row with all-zero features) are centred by `lls -= lls.mean(axis=1)` (projector.py:26 / :55), and
whether that leaves exactly 0 -- a row the Hilbert coreset drops, a NaN candidate of the greedy-VI classes -- hangs on the last
bit of the constant.  For the beta-likelihoods of the linear-regression and Gaussian models the constant contains `np.exp`
(model_neurlinr.py:107, gaussian.py:42), and on x86-64 hosts with AVX512_SKX NumPy >= 1.22 evaluates that with the SVML
routine it bundles, whose last bit differs from libm's for ~5 % of arguments.  The library restates the SVML routine
(csrc/bc_np_exp.h) -- the bits of the hosts the goldens were generated on.  On any other host "the reference's bits" are that
host's own NumPy's; the projector then evaluates the constants of constant rows on the host (LinearRegression) and says so.
"""
import math
import warnings

import numpy as np

_cached = None


def numpy_uses_svml_exp():
    """True where np.exp (float64) is NumPy's bundled SVML routine: AVX512_SKX dispatch present AND its results differ from
    libm's on a sample of arguments (the same probe the tests use)."""
    global _cached
    if _cached is None:
        _cached = _probe()
    return _cached


def _probe():
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


_warned = set()


def warn_if_constant_bits_differ(model):
    """Called when a device projector is built around `model`.  Returns True when the host route for constant rows is needed
    (the model's constant holds an np.exp and this NumPy is not the one the library restates); warns once per model class."""
    if not getattr(model, 'constant_has_numpy_exp', False) or numpy_uses_svml_exp():
        return False
    name = type(model).__name__
    if name not in _warned:
        _warned.add(name)
        how = ('their constants are evaluated on the host with this NumPy and handed to the kernel'
               if getattr(model, 'host_constants', None) is not None else
               'the kernel keeps the AVX-512 NumPy bits for them (they only arise for degenerate sample matrices with this model)')
        warnings.warn('%s: this NumPy does not evaluate np.exp with the SVML routine the GPU library restates (no AVX512_SKX '
                      'dispatch): the last bit of constant projection rows (data rows with all-zero features, beta-likelihood) '
                      'follows the host here -- %s' % (name, how), UserWarning, stacklevel=3)
    return True
