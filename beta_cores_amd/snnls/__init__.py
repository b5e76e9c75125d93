"""Same exports as bayesiancoresets/snnls/__init__.py:1-4."""
from .frankwolfe import FrankWolfe
from .sampling import ImportanceSampling, UniformSampling
from .giga import GIGA
from .orthopursuit import OrthoPursuit
from .snnls import SparseNNLS
