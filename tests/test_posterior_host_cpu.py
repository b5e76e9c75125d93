"""Host-side pieces of the posterior / sampler code that need no GPU: the re-entrant single-thread BLAS scope and the
samplers' look-ahead normals (same stream, same order as the reference's closures)."""
import numpy as np

from beta_cores_amd import posterior as P
from beta_cores_amd.samplers import _PosteriorSampler


def test_small_lapack_scope_is_reentrant_and_restores():
    assert P._limit_depth == 0
    with P.small_lapack_scope(64):
        outer = P._limit_depth
        with P.small_lapack_scope(64):                 # the per-call scope inside an open loop-level scope: a no-op
            assert P._limit_depth == outer
        assert P._limit_depth == outer
    assert P._limit_depth == 0
    with P.small_lapack_scope(4096):                   # large systems keep the BLAS threads
        assert P._limit_depth == 0
    try:
        with P.small_lapack_scope(8):
            raise ValueError('boom')
    except ValueError:
        pass
    assert P._limit_depth == 0                         # the count is restored when the body raises


def test_prefetched_normals_are_the_next_draw():
    class S(_PosteriorSampler):
        def __init__(self, rng):
            super().__init__(rng)
            self._shape = None

    a, b = S(np.random.RandomState(5)), S(np.random.RandomState(5))
    x1 = a._normals(7, 3)
    a.prefetch()                                        # draws call 2's matrix now ...
    x2 = a._normals(7, 3)                               # ... and hands it out here
    y1, y2 = b._normals(7, 3), b._normals(7, 3)
    assert np.array_equal(x1, y1) and np.array_equal(x2, y2)
    a.prefetch()
    import pytest
    with pytest.raises(RuntimeError, match='prefetched'):   # shape changed: the block is already out of the stream, dropping it
        a._normals(5, 3)                                    # would leave the stream one block ahead of the reference's -- loud
    assert a._ahead is None
