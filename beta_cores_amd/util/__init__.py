"""bayesiancoresets.util surface (util/__init__.py:1-7): nn_opt, set_verbosity, TOL, set_tolerance."""
from .opt import nn_opt, partial_nn_opt
from .log import set_verbosity, install_default_handler
from .errors import NumericalPrecisionError

TOL = 1e-12


def set_tolerance(tol):
    global TOL
    TOL = tol


install_default_handler()
