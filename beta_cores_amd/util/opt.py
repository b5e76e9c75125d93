"""Projected ADAM-moment optimisers for the (tiny) coreset weight vector.

Host NumPy on purpose: x has M (+1) entries; the expensive part is the `grd`
callback, which runs the device projection.  Arithmetic follows
bayesiancoresets/util/opt.py:36-77 operation for operation (pure element-wise
float64, so results are bit-identical to the reference)."""
import numpy as np


def _adam_loop(x0, grd, opt_itrs, step_sched, b1, b2, eps, project):
    x = x0.copy()
    mom1 = np.zeros(x.shape[0])
    mom2 = np.zeros(x.shape[0])
    for i in range(opt_itrs):
        g = grd(x)
        mom1 = b1 * mom1 + (1. - b1) * g
        mom2 = b2 * mom2 + (1. - b2) * g ** 2
        upd = step_sched(i) * mom1 / (1. - b1 ** (i + 1)) / (eps + np.sqrt(mom2 / (1. - b2 ** (i + 1))))
        x -= upd
        x = project(x)
    return x


def nn_opt(x0, grd, opt_itrs=1000, step_sched=lambda i: 1. / (i + 1), b1=0.9, b2=0.999, eps=1e-8, verbose=False):
    """x <- max(x - upd, 0) on every coordinate (opt.py:36-54)."""
    return _adam_loop(x0, grd, opt_itrs, step_sched, b1, b2, eps, lambda x: np.maximum(x, 0.))


def partial_nn_opt(x0, grd, nn_idcs, opt_itrs=1000, step_sched=lambda i: 1. / (i + 1), b1=0.9, b2=0.999, eps=1e-8,
                   verbose=False):
    """Non-negativity only on `nn_idcs` (opt.py:56-77)."""
    def project(x):
        x[nn_idcs] = np.maximum(x[nn_idcs], 0.)
        return x
    return _adam_loop(x0, grd, opt_itrs, step_sched, b1, b2, eps, project)
