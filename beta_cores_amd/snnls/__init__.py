"""Sparse non-negative least-squares solvers with the matrix resident on the GPU.

The public names are the ones `bayesiancoresets.snnls` exposes (its __init__ lists FrankWolfe,
ImportanceSampling, UniformSampling, GIGA and OrthoPursuit), plus the shared base class.
"""
from .snnls import SparseNNLS
from .giga import GIGA
from .frankwolfe import FrankWolfe
from .orthopursuit import OrthoPursuit
from .sampling import ImportanceSampling, UniformSampling

__all__ = ['SparseNNLS', 'GIGA', 'FrankWolfe', 'OrthoPursuit', 'ImportanceSampling', 'UniformSampling']
