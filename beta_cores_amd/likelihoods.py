"""Likelihood models evaluated by the K1 projection kernel.

Each class names one formula of the reference's examples/common/*.py, holds its
hyper-parameters and knows how to lay them out for bc_project (include/beta_cores.h).
`theta_for_device(samples)` returns the S x D matrix the contraction runs against
(for the Gaussian-location model that is Theta.Siginv, so the kernel's GEMM gives
x^T Siginv theta directly).
"""
import numpy as np

LINREG_LL, LINREG_BETA, LOGISTIC_LL, LOGISTIC_BETA, GAUSS_LL, GAUSS_BETA, GAUSS_BETA_GRAD = range(7)


def _checked_beta(beta):
    """The device bodies evaluate exp(-b q) with a clamp for large NEGATIVE arguments only (q >= 0): b < 0 is refused."""
    beta = float(beta)
    if beta < 0.:
        raise ValueError('beta must be >= 0 for the device beta-likelihoods (got %r)' % beta)
    return beta


class _Model:
    model_id = None
    beta_model_id = None
    beta_grad_model_id = None
    has_grad_x = False        # d/dx of the log-likelihood (bc_project_grad_x)

    def theta_for_device(self, samples):
        return np.ascontiguousarray(np.atleast_2d(samples), dtype=np.float64)

    def data_width(self, theta_dim):
        raise NotImplementedError


class LinearRegression(_Model):
    """Rows z = [x (D), y].  log-lik: model_linreg.py:4-10 == model_neurlinr.py:90-97;
    beta-likelihood: model_neurlinr.py:102-110; x-gradient: model_linreg.py:12-17."""
    has_grad_x = True
    model_id = LINREG_LL
    beta_model_id = LINREG_BETA

    constant_has_numpy_exp = True      # the beta-likelihood of a row with x = 0 is c(y) and holds an np.exp (util/numpy_bits.py)

    def __init__(self, sigsq=1.0):
        self.sigsq = float(sigsq)

    def params(self, beta=None, grad=False):
        return np.array([self.sigsq] if beta is None else [self.sigsq, _checked_beta(beta)])

    def host_constants(self, y, beta):
        """model_neurlinr.py:102-110 for rows with all-zero features (XST = x.th^T = 0 for every sample), evaluated with this
        host's NumPy in the reference's expression order: the S equal values such a row projects to, one per y.  (Array
        arithmetic on an (n, 1) column, the shape the reference's N x S expression has per sample.)"""
        sigsq, beta = self.sigsq, float(beta)
        y = np.asarray(y, dtype=np.float64)
        XST = np.zeros((y.shape[0], 1))
        with np.errstate(all='ignore'):
            vals = 1. / (2 * np.pi * sigsq) ** (beta / 2.) * (-(beta + 1.) / beta * np.exp(-beta / (2. * sigsq) * (y[:, np.newaxis] ** 2 - 2 * XST * y[:, np.newaxis] + XST ** 2))
                                                              + 1. / np.sqrt(1. + beta))
        return vals[:, 0]

    def data_width(self, theta_dim):
        return theta_dim + 1


class LogisticRegression(_Model):
    """Rows z = y*x (D).  log-lik: model_lr.py:72-79; beta-likelihood: model_lr.py:81-86; z-gradient: :107-114."""
    has_grad_x = True
    model_id = LOGISTIC_LL
    beta_model_id = LOGISTIC_BETA

    MAX_BETA = 32.     # csrc/bc_k1_math.h: BC_K1_POWTAB_MAX_BETA (the power series of the device body is truncated for beta up to here)

    def params(self, beta=None, grad=False):
        if beta is None:
            return np.array([])
        beta = _checked_beta(beta)
        if not 0. < beta <= self.MAX_BETA:
            raise ValueError('the device logistic beta-likelihood takes 0 < beta <= %g (got %r)' % (self.MAX_BETA, beta))
        return np.array([beta, self.beta_value_at_zero(beta)])

    @staticmethod
    def beta_value_at_zero(beta):
        """model_lr.py:85 at m = 0, i.e. the S equal values a data row z = 0 projects to, evaluated by NumPy's own ARRAY
        arithmetic exactly as the reference does (its `**` is np.power, whose last bit differs from libm's pow for ~5 % of
        exponents on AVX-512 hosts; a 1-element array takes the same routine as any element of an N x S one).  Whether such
        a row centres to exactly 0 (projector.py:55) -- and then is a NaN candidate of every argmax (bcores.py:78-81) --
        hangs on that bit, so K1 takes the constant from here instead of computing its own (goldens F19 / F20)."""
        m = np.zeros((1, 1))
        with np.errstate(all='ignore'):
            c = -(((beta + 1.) / beta) * (1 + np.exp(m)) ** (-beta) - ((1 + np.exp(m)) ** (-beta - 1.) + (1 + np.exp(-m)) ** (-beta - 1.)))
        return float(c[0, 0])

    def data_width(self, theta_dim):
        return theta_dim


class GaussianLocation(_Model):
    """Rows x (d), known covariance.  gaussian.py:7-15 (log-lik), :17-20 (x-gradient), :34-44 (beta-likelihood),
    :46-62 (d/dbeta)."""
    has_grad_x = True
    model_id = GAUSS_LL
    beta_model_id = GAUSS_BETA
    beta_grad_model_id = GAUSS_BETA_GRAD

    constant_has_numpy_exp = True      # (gaussian.py:42; constant rows need all samples equidistant from x: no host route)
    host_constants = None

    def __init__(self, Siginv, logdetSig):
        self.Siginv = np.ascontiguousarray(Siginv, dtype=np.float64)
        self.logdetSig = float(logdetSig)

    def params(self, beta=None, grad=False):
        head = [self.logdetSig] if beta is None else [_checked_beta(beta), self.logdetSig]
        return np.concatenate((np.array(head), self.Siginv.ravel()))

    def theta_for_device(self, samples):
        return np.ascontiguousarray(np.atleast_2d(samples), dtype=np.float64)

    def data_width(self, theta_dim):
        return theta_dim
