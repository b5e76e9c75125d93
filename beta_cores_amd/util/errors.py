class NumericalPrecisionError(Exception):
    """Raised when a greedy step hits the numeric-precision limit.

    Same name and role as bayesiancoresets/util/errors.py:1-2: it is always caught
    inside build()/optimize() and turned into `reached_numeric_limit = True`."""
